mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_graph.py tests/test_gpu_hidden.py -x -q > gpurun_out/r4/t5.log 2>&1; tail -4 gpurun_out/r4/t5.log
for r in 1 2; do
for m in "--one-stream --no-graph" "--two-streams --no-graph" "--two-streams --graph" "--one-stream --no-graph --keep-dead-grads"; do
  python bench.py --no-cpu-baseline --no-extra $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],4), round(d['ms_per_step_median_events'],4), 'host', round(d['host_enqueue_ms_median'],3), 'bwd', round(d['roofline']['avg_launch_ms']*1000,1), 'fwd', round(d['roofline_mfma']['avg_launch_ms']*1000,1), round(d['step_flops_frac_of_peak'],4))"
done; done 2>&1 | tee gpurun_out/r4/modes2.log
python bench.py --no-cpu-baseline --two-streams --graph > gpurun_out/r4/b3.json 2> gpurun_out/r4/b3.err; tail -c 1800 gpurun_out/r4/b3.json
