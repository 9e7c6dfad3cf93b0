#!/bin/bash
# a library VARIANT for tools/ab_libs.sh: one or more sources recompiled with extra -D flags, linked with the release objects of the others
# usage: bash tools/build_variant.sh <name> <source.hip[,source2.hip,...]> "<-D flags>"   ->  tools/micro/ab/libwm_hip_<name>.so
# VARIANT_SRC_DIR=<dir>: take the named sources from <dir> instead of csrc/ (e.g. `git show HEAD~1:.../bwd_ws8.hip > dir/bwd_ws8.hip`: the
# previous commit's kernel as the A of an A/B); headers still come from csrc/
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/video_watermarking_forgery_detection_amd/csrc
name=$1; srcs=",$2,"; defs=$3
python -m video_watermarking_forgery_detection_amd.build > /dev/null
mkdir -p $ROOT/tools/micro/ab /tmp/wmvar_$name
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
objs=""
for o in $C/_build/*.o; do
  case $o in *.dbg.o) continue;; esac
  b=$(basename $o)
  src=${b%.o}; f16=""
  case $src in *.f16) src=${src%.f16}; f16="-DWM_H16_F16";; esac
  case $srcs in
    *,$src,*) /opt/rocm/bin/hipcc $FL $defs $f16 -I$C -x hip -c ${VARIANT_SRC_DIR:-$C}/$src -o /tmp/wmvar_$name/$b 2> /tmp/wmvar_$name/$b.log; objs="$objs /tmp/wmvar_$name/$b";;
    *) objs="$objs $o";;
  esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/micro/ab/libwm_hip_$name.so $objs
grep -h -A12 "Function Name: .*\(bwd_ws8\|conv3x3_ws_kernelILi64ELi64ELb1ELb1ELb1\)" /tmp/wmvar_$name/*.o.log 2>/dev/null | grep -E "Function Name|VGPRs:|Spill|ScratchSize" | head -12
echo "built tools/micro/ab/libwm_hip_$name.so"
