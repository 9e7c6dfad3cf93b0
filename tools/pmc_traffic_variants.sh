#!/bin/bash
# ON THE GPU BOX: L2-miss traffic (FETCH_SIZE, WRITE_SIZE; separate passes) of the one-pass backward kernel alone, for several BUILDS of the
# library (tools/micro/ab/libwm_hip_<name>.so, tools/build_variant.sh).  usage: bash tools/pmc_traffic_variants.sh A B ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_var
rm -rf $OUT; mkdir -p $OUT
# (the variant is chosen through WM_LIB_VARIANT, _lib.py: the release library file is never touched)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export WM_LIB_VARIANT=$v
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$v/f -o f -- python3 $ROOT/tools/run_bwd_fused.py 6 > $OUT/$v.f.log 2>&1 || echo "$v fetch failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$v/w -o w -- python3 $ROOT/tools/run_bwd_fused.py 6 > $OUT/$v.w.log 2>&1 || echo "$v write failed"
done
unset WM_LIB_VARIANT
python3 - "$@" <<PY
import csv, glob, sys
for v in sys.argv[1:]:
    tot = {}
    for sub, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
        fs = glob.glob("$OUT/%s/%s/**/*counter_collection.csv" % (v, sub), recursive=True)
        vals = [float(r["Counter_Value"]) for f in fs for r in csv.DictReader(open(f)) if "bwd_ws8" in r["Kernel_Name"] and r["Counter_Name"] == name]
        tot[name] = sum(vals) / max(1, len(vals))
    b = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
    print("%s: FETCH_SIZE %.0f KB (x2 = %.1f MB)  WRITE_SIZE %.0f KB (%.1f MB)  -> %.1f MB per launch = %.3f x 536.9 MB" % (v, tot["FETCH_SIZE"], 2 * tot["FETCH_SIZE"] * 1024 / 1e6, tot["WRITE_SIZE"], tot["WRITE_SIZE"] * 1024 / 1e6, b / 1e6, b / 536.9e6))
PY
