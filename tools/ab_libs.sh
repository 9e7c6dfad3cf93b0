#!/bin/bash
# same-box A/B of several BUILDS of the library (a change with no debug knob): put them at tools/micro/ab/libwm_hip_<name>.so (git-ignored, they
# travel with gpurun) and run this on the GPU box: bench.py with each in turn, three rounds -> ms per step (wall, event median), dominant
# kernel us.  Resolves +-0.3 us on the dominant kernel.  usage: bash tools/ab_libs.sh A B [C ...]
# The variant is chosen through WM_LIB_VARIANT (_lib.py): the release library file is never touched.
ROUNDS=${AB_ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    WM_LIB_VARIANT=$v python bench.py --no-cpu-baseline --no-extra ${AB_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['ms_per_step_median_events'],4), round(d['roofline']['avg_launch_ms']*1000,1), round(d['roofline_mfma']['avg_launch_ms']*1000,1))"
  done
done
