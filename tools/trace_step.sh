#!/bin/bash
# ON THE GPU BOX: the ORDERED kernel list of one benchmarked step (rocprofv3 kernel trace of tools/find_copies.py's loop): which launch
# precedes / follows every runtime blit (__amd_rocclr_copyBuffer) and every torch-side kernel, with start offsets and durations.
# usage: bash tools/trace_step.sh   -> gpurun_out/trace_step/step.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o k -- python3 $ROOT/tools/run_steps.py 4 > $OUT/run.log 2>&1 || echo "trace failed"
python3 - <<PY
import csv, glob
fs = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last pack_w3x3_batch triple back... simpler: split at hidden_metrics_kernel
ends = [i for i, r in enumerate(rows) if "hidden_metrics" in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
with open("$OUT/step.txt", "w") as f:
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%9.1f us  +gap %6.1f  dur %7.1f  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:110]))
        prev_end = e
    f.write("step: %d launches, %.1f us from first start to last end, %.1f us of kernels\n" % (b - a, (prev_end - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3))
print(open("$OUT/step.txt").read()[-400:])
PY
