#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel stats + the two PMC passes of the bench command, into gpurun_out/prof_final.
# usage: bash tools/collect_profiles.sh   (then tools/summarise_profiles.py locally copies the summaries into profiles/)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
echo "write done"
ls -R $OUT | head -30
