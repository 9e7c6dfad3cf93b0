#!/bin/bash
# same-box A/B of several BUILDS of the library (a change with no debug knob): put them at tools/micro/ab/libwm_hip_<name>.so (git-ignored, they
# travel with gpurun) and run this on the GPU box: bench.py with each in turn, three rounds -> ms per step (wall, event median), dominant
# kernel us.  Resolves +-0.3 us on the dominant kernel.  usage: bash tools/ab_libs.sh A B [C ...]   (restores the library it found)
L=video_watermarking_forgery_detection_amd/lib/libwm_hip.so
cp $L /tmp/libwm_hip_saved.so
for r in 1 2 3; do
  for v in "$@"; do
    cp tools/micro/ab/libwm_hip_$v.so $L
    python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['ms_per_step_median_events'],4), round(d['roofline']['avg_launch_ms']*1000,1))"
  done
done
cp /tmp/libwm_hip_saved.so $L
