"""gpurun_out/prof_final (tools/collect_profiles.sh) -> profiles/r01_* (committed summaries).
FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section); both counters
are in KB per dispatch."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_final")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_c2_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
steps = 13.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"{tag}_bench_c2_kernel_stats_summary.txt"), "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3\n")
    f.write(f"total kernel ms: {tot/1e6:.2f}  (/{steps:g} steps = {tot/1e6/steps:.2f} ms/step)\n")
    for r in rows[:40]:
        f.write(f"{r['Name'][:110]:110s} n={int(r['Calls']):5d} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%\n")
out = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, f"{tag}_bench_c2_pmc_{name}.csv"), "w") as g:
        g.write("kernel,dispatches,mean_KB_per_dispatch\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            g.write(f"\"{k[:120]}\",{len(v)},{sum(v)/len(v):.2f}\n")
    out[name] = {k: sum(v) / len(v) for k, v in agg.items()}
def traffic(pattern):
    k = [k for k in out["FETCH_SIZE"] if pattern in k][0]
    return k, out["FETCH_SIZE"][k], out["WRITE_SIZE"][k], (2.0 * out["FETCH_SIZE"][k] + out["WRITE_SIZE"][k]) * 1024.0
# dominant kernel: the fused 64->64 input-gradient conv (BNBWD = 2, BWDST); runner-up: the forward 64->64 conv (XFORM, STATS)
k, f, w, b = traffic("conv3x3_ws_kernel<64, 64, false, false, true, false, 2, true")
k2, f2, w2, b2 = traffic("conv3x3_ws_kernel<64, 64, true, true")
res = {"kernel": k, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": b,
       "fwd_kernel": k2, "fwd_FETCH_SIZE_KB": f2, "fwd_WRITE_SIZE_KB": w2, "fwd_hbm_bytes_per_launch": b2,
       "note": "FETCH_SIZE doubled (gfx950 counts half of wide coalesced reads); KB = 1024 B"}
json.dump(res, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
