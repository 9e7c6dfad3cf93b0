#!/bin/bash
# ON THE GPU BOX: SQ counter passes of the forward 64 -> 64 conv alone (tools/run_fwd.py; 4 consumer + 4 producer waves per workgroup: the counters are sums over both roles), where its cycles go:
# wave-cycles split into parked (s_waitcnt / barrier), issue-stalled, issuing; the LDS pipe; instruction counts.  Counters only (no trace domains).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_fwd
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o a -- python3 $ROOT/tools/run_fwd.py 6 > $OUT/p1.log 2>&1 || echo "p1 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -o b -- python3 $ROOT/tools/run_fwd.py 6 > $OUT/p2.log 2>&1 || echo "p2 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_WAVE32_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p3 -o c -- python3 $ROOT/tools/run_fwd.py 6 > $OUT/p3.log 2>&1 || echo "p3 failed"
python3 - <<PY
import csv, glob, collections
for sub in ("p1","p2","p3"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print(sub, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "conv3x3_ws_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(sub, k)
        for c, vals in sorted(v.items()):
            print("    %-32s %16.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
