#!/bin/bash
# a library VARIANT for tools/ab_libs.sh: one source recompiled with extra -D flags, linked with the release objects of the others
# usage: bash tools/build_variant.sh <name> <source.hip> "<-D flags>"   ->  tools/micro/ab/libwm_hip_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/video_watermarking_forgery_detection_amd/csrc
name=$1; src=$2; defs=$3
python -m video_watermarking_forgery_detection_amd.build > /dev/null
mkdir -p $ROOT/tools/micro/ab /tmp/wmvar_$name
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
objs=""
for o in $C/_build/*.o; do
  case $o in *.dbg.o) continue;; esac
  b=$(basename $o)
  if [ "$b" = "$src.o" ]; then
    /opt/rocm/bin/hipcc $FL $defs -x hip -c $C/$src -o /tmp/wmvar_$name/$b 2> /tmp/wmvar_$name/$b.log; objs="$objs /tmp/wmvar_$name/$b"
  elif [ "$b" = "$src.f16.o" ]; then
    /opt/rocm/bin/hipcc $FL $defs -DWM_H16_F16 -x hip -c $C/$src -o /tmp/wmvar_$name/$b 2> /tmp/wmvar_$name/$b.log; objs="$objs /tmp/wmvar_$name/$b"
  else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/micro/ab/libwm_hip_$name.so $objs
grep -h -A12 "Function Name: .*bwd_ws8" /tmp/wmvar_$name/$src.o.log | grep -E "Function Name|VGPRs:|Spill|ScratchSize" | head -12
echo "built tools/micro/ab/libwm_hip_$name.so"
