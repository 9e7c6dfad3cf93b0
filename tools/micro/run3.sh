mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_graph.py -x -q > gpurun_out/r4/t4.log 2>&1; tail -4 gpurun_out/r4/t4.log
for r in 1 2; do
for m in "--one-stream --no-graph" "--two-streams --no-graph" "--two-streams --graph" "--one-stream --graph"; do
  python bench.py --no-cpu-baseline --no-extra $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],4), round(d['ms_per_step_median_events'],4), 'host', round(d['host_enqueue_ms_median'],3), 'bwd', round(d['roofline']['avg_launch_ms']*1000,1), 'fwd', round(d['roofline_mfma']['avg_launch_ms']*1000,1))"
done; done 2>&1 | tee gpurun_out/r4/modes.log
AB_ROUNDS=2 bash tools/ab_libs.sh old new 2>&1 | tee gpurun_out/r4/ab_prologue.log
