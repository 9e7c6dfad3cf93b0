#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel stats + separate PMC passes of the bench command and of the attack kernels, into
# gpurun_out/prof_final.  Then tools/summarise_profiles.py <tag> (locally) copies the summaries into profiles/.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains other than
# the kernel trace).  The program itself follows `--` (no env / bash -c hop).
# usage: bash tools/collect_profiles.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
echo "write done"
# MFMA utilisation as the counters report it: busy cycles of the matrix pipe against the cycles the dispatch kept the chip busy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_mfma.log 2>&1 || echo "mfma pass failed (see pmc_mfma.log)"
echo "mfma done"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma2 -o m2 -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_mfma2.log 2>&1 || echo "mfma2 pass failed (see pmc_mfma2.log)"
echo "mfma2 done"
# the attack kernels at B=16, 256x256
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/att_stats -o a -- python3 $ROOT/tools/attack_bench.py 20 > $OUT/att_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/att_fetch -o af -- python3 $ROOT/tools/attack_bench.py 5 > $OUT/att_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/att_write -o aw -- python3 $ROOT/tools/attack_bench.py 5 > $OUT/att_write.log 2>&1
python3 $ROOT/tools/attack_bench.py 50 > $OUT/attack_bench.json 2> $OUT/attack_bench.err
echo "attacks done"
# the widened configurations through the model surface, the literal IRNrhi step, the one-pass backward kernel's phases, the co-issue micro
cd $ROOT
(python3 tools/bench_c5.py train_hidden_c3.yml bf16 72 && python3 tools/bench_c5.py train_hidden_c3.yml f16 72 && python3 tools/bench_c5.py train_hidden_c5.yml bf16 72 &&
 python3 tools/bench_c5.py train_hidden_c5_fp16.yml f16 72) 2> $OUT/c3_c5.err | grep '^{' > $OUT/c3_c5_steps.jsonl
(python3 tools/bench_literal.py 4 bf16 12 && python3 tools/bench_literal.py 4 f16 12) 2> $OUT/literal.err | grep '^{' > $OUT/literal_steps.jsonl
(python3 tools/bench_inn.py 8 bf16 6 && python3 tools/bench_inn.py 8 bf16 6 graph && python3 tools/bench_inn.py 8 f16 6 graph) 2> $OUT/inn.err | grep '^{' > $OUT/inn_steps.jsonl
(python3 tools/phase_bwd.py 0 && python3 tools/phase_bwd.py 256 && python3 tools/phase_bwd.py 8 && python3 tools/phase_bwd.py 1048832) 2>&1 | grep -v amdgpu.ids > $OUT/bwd_phase_cycles.txt
if [ -x tools/micro/mfma_rate ]; then tools/micro/mfma_rate > $OUT/mfma_coissue_micro.txt 2>&1; fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lit_stats -o l -- python3 $ROOT/tools/bench_literal.py 4 bf16 6 > $OUT/lit_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o c -- python3 $ROOT/tools/bench_c5.py train_hidden_c5_fp16.yml f16 44 > $OUT/c5_stats.log 2>&1
# north_star's second size: 512 x 512 (C4's per-GPU shard: 8 frames), bench line + kernel stats
python3 $ROOT/bench.py --no-cpu-baseline --size 512 --batch 8 2> $OUT/b512.err | tail -1 > $OUT/bench_512_b8.json
python3 $ROOT/bench.py --no-cpu-baseline --keep-dead-grads 2> /dev/null | tail -1 > $OUT/bench_c2_keep_dead_grads.json
python3 $ROOT/bench.py --no-cpu-baseline 2> /dev/null | tail -1 > $OUT/bench_c2.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s512_stats -o q -- python3 $ROOT/bench.py --no-cpu-baseline --size 512 --batch 8 --steps 10 --warmup 3 > $OUT/s512_stats.log 2>&1
bash $ROOT/tools/pmc_bwd.sh > $OUT/bwd_sq_counters.txt 2>&1
cd $ROOT
(python3 tools/phase_bwd8.py && python3 tools/phase_bwd8.py gvec) 2>&1 | grep -v amdgpu.ids > $OUT/bwd8_phase_cycles.txt
echo "widened done"
ls -R $OUT | head -40
