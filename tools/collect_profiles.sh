#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel stats + separate PMC passes of the bench command and of the attack kernels, into
# gpurun_out/prof_final.  Then tools/summarise_profiles.py <tag> (locally) copies the summaries into profiles/.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains other than
# the kernel trace).  The program itself follows `--` (no env / bash -c hop).
# usage: bash tools/collect_profiles.sh a|b     (two gpurun calls: a = the benchmarked step + the attack kernels, b = the widened configurations
# and the kernel diagnostics; each fits gpurun's 20-minute limit, both write into gpurun_out/prof_final, which gpurun merges back)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_final
PART=${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
set -e
ONE="--no-cpu-baseline --no-extra --one-stream --no-graph"
if [ "$PART" = "a" ]; then
# Round 4: bench.py's default runs the step as two chains on two streams, replayed from a hipGraph: beside a launch of the other chain a
# kernel's traced duration includes the time it shares the chip.  The per-kernel numbers (durations, counters) are therefore taken from
# the SAME step enqueued on one stream (--one-stream --no-graph: the very launches, one after the other -- what bench.py's own event
# brackets time); the default mode's trace is kept beside it ("stats2").  --no-extra: the headline workload only (no 512 x 512 / keep-dead region).
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py $ONE --steps 10 --warmup 3 > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o s2 -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/stats2.log 2>&1
echo "stats (two chains + graph) done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py $ONE --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py $ONE --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
echo "write done"
# MFMA utilisation as the counters report it: busy cycles of the matrix pipe against the cycles the dispatch kept the chip busy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 $ROOT/bench.py $ONE --steps 3 --warmup 1 > $OUT/pmc_mfma.log 2>&1 || echo "mfma pass failed (see pmc_mfma.log)"
echo "mfma done"
# the attack kernels at B=16, 256x256
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/att_stats -o a -- python3 $ROOT/tools/attack_bench.py 20 > $OUT/att_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/att_fetch -o af -- python3 $ROOT/tools/attack_bench.py 5 > $OUT/att_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/att_write -o aw -- python3 $ROOT/tools/attack_bench.py 5 > $OUT/att_write.log 2>&1
python3 $ROOT/tools/attack_bench.py 50 > $OUT/attack_bench.json 2> $OUT/attack_bench.err
echo "attacks done"
python3 $ROOT/bench.py --no-cpu-baseline --no-extra --size 512 --batch 8 2> $OUT/b512.err | tail -1 > $OUT/bench_512_b8.json
python3 $ROOT/bench.py --no-cpu-baseline --no-extra --keep-dead-grads 2> /dev/null | tail -1 > $OUT/bench_c2_keep_dead_grads.json
python3 $ROOT/bench.py --no-cpu-baseline 2> /dev/null | tail -1 > $OUT/bench_c2.json
# the step's four modes, interleaved, two rounds (one GPU): one stream / two chains x enqueued / replayed
(for r in 1 2; do for m in "--one-stream --no-graph" "--two-streams --no-graph" "--two-streams --graph" "--one-stream --graph"; do
  python3 $ROOT/bench.py --no-cpu-baseline --no-extra $m 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m: ms_per_step', round(d['ms_per_step'],4), 'median of the step events', round(d['ms_per_step_median_events'],4), 'min', round(d['ms_per_step_min_events'],4), 'host enqueue ms', round(d['host_enqueue_ms_median'],3), 'bwd_ws8 us', round(d['roofline']['avg_launch_ms']*1000,1), 'fwd 64->64 us', round(d['roofline_mfma']['avg_launch_ms']*1000,1))"
done; done) > $OUT/step_modes.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s512_stats -o q -- python3 $ROOT/bench.py $ONE --size 512 --batch 8 --steps 10 --warmup 3 > $OUT/s512_stats.log 2>&1
echo "part a done"
ls -R $OUT | head -40
exit 0
fi
# ---- part b: the widened configurations through the model surface, the literal IRNrhi step, the one-pass backward kernel's phases, the co-issue micro
cd $ROOT
(python3 tools/bench_c5.py train_hidden_c3.yml bf16 130 && python3 tools/bench_c5.py train_hidden_c3.yml bf16 130 deferred && python3 tools/bench_c5.py train_hidden_c3.yml f16 130 && python3 tools/bench_c5.py train_hidden_c5.yml bf16 130 &&
 python3 tools/bench_c5.py train_hidden_c5_fp16.yml f16 130) 2> $OUT/c3_c5.err | grep '^{' > $OUT/c3_c5_steps.jsonl
(python3 tools/bench_literal.py 4 bf16 12 && python3 tools/bench_literal.py 4 f16 12) 2> $OUT/literal.err | grep '^{' > $OUT/literal_steps.jsonl
(python3 tools/bench_inn.py 8 bf16 6 && python3 tools/bench_inn.py 8 bf16 6 graph && python3 tools/bench_inn.py 8 f16 6 graph && INN_PAR=0 python3 tools/bench_inn.py 8 bf16 6 graph) 2> $OUT/inn.err | grep '^{' > $OUT/inn_steps.jsonl
(python3 tools/phase_bwd.py 0 && python3 tools/phase_bwd.py 256 && python3 tools/phase_bwd.py 8 && python3 tools/phase_bwd.py 1048832) 2>&1 | grep -v amdgpu.ids > $OUT/bwd_phase_cycles.txt
if [ -x tools/micro/mfma_rate ]; then tools/micro/mfma_rate > $OUT/mfma_coissue_micro.txt 2>&1; fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lit_stats -o l -- python3 $ROOT/tools/bench_literal.py 4 bf16 6 > $OUT/lit_stats.log 2>&1
INN_PAR=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inn_stats -o i -- python3 $ROOT/tools/bench_inn.py 8 bf16 4 > $OUT/inn_stats.log 2>&1   # (one stream: every kernel alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o c -- python3 $ROOT/tools/bench_c5.py train_hidden_c5_fp16.yml f16 44 > $OUT/c5_stats.log 2>&1
bash $ROOT/tools/pmc_bwd.sh > $OUT/bwd_sq_counters.txt 2>&1
bash $ROOT/tools/pmc_fwd.sh > $OUT/fwd_sq_counters.txt 2>&1
bash $ROOT/tools/trace_step.sh > $OUT/trace_step.log 2>&1; cp $ROOT/gpurun_out/trace_step/step.txt $OUT/step_trace_one_stream.txt
cd $ROOT
python3 tools/phase_ws.py 2>&1 | grep -v amdgpu.ids > $OUT/fwd_phase_cycles.txt
if [ -f tools/micro/ab/libwm_hip_half.so ]; then python3 tools/bench_two_chains.py half 5 2>&1 | grep -v amdgpu.ids > $OUT/two_chains.txt; fi
(python3 tools/phase_bwd8.py && python3 tools/phase_bwd8.py gvec) 2>&1 | grep -v amdgpu.ids > $OUT/bwd8_phase_cycles.txt
echo "widened done"
ls -R $OUT | head -40
