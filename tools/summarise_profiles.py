"""gpurun_out/prof_final (tools/collect_profiles.sh) -> profiles/<tag>_* (committed summaries).

FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section); both traffic counters are in
KB per dispatch.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): busy cycles of the
matrix pipes over the SIMD-cycles the dispatch held the chip (the gfx94x MfmaUtil formula; GRBM_GUI_ACTIVE is summed over the XCDs).
usage: python tools/summarise_profiles.py <tag>"""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_final")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
os.makedirs(dst, exist_ok=True)
NXCD, NCU, NSIMD = 8, 256, 4


def find(sub, pat):
    # (gpurun MERGES the box's gpurun_out into the local one: files of earlier collections with other -o prefixes survive; take the newest)
    return max(glob.glob(os.path.join(src, sub, "**", pat), recursive=True), key=os.path.getmtime)


def stats_summary(sub, out_base, steps, cmd):
    stats = find(sub, "*kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, out_base + ".csv"))
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(dst, out_base + "_summary.txt"), "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- {cmd}\n")
        f.write(f"total kernel ms: {tot/1e6:.2f}" + (f"  (/{steps:g} steps = {tot/1e6/steps:.2f} ms/step)\n" if steps else "\n"))
        for r in rows[:48]:
            per = f"ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} " if steps else ""
            f.write(f"{r['Name'][:110]:110s} n={int(r['Calls']):5d} {per}avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%\n")
    return {r["Name"]: float(r["AverageNs"]) for r in rows}


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def traffic_tables(fetch_sub, write_sub, base):
    out = {}
    for name, sub in (("FETCH_SIZE", fetch_sub), ("WRITE_SIZE", write_sub)):
        agg = counters(sub)
        with open(os.path.join(dst, f"{base}_pmc_{name}.csv"), "w") as g:
            g.write("kernel,dispatches,mean_KB_per_dispatch\n")
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][name])):
                g.write(f"\"{k[:120]}\",{len(v[name])},{sum(v[name])/len(v[name]):.2f}\n")
        out[name] = {k: sum(v[name]) / len(v[name]) for k, v in agg.items()}
    return out


avg_ns = stats_summary("stats", f"{tag}_bench_c2_kernel_stats", 13.0, "python3 bench.py --no-cpu-baseline --no-extra --one-stream --no-graph --steps 10 --warmup 3   (the step's launches one after the other: every kernel alone on the chip)")
try:
    stats_summary("stats2", f"{tag}_bench_c2_two_chains_graph_kernel_stats", 13.0, "python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3   (bench.py's default: two chains on two streams, replayed from a hipGraph -- a kernel's duration here includes the time it shares the chip with a launch of the other chain)")
    os.remove(os.path.join(dst, f"{tag}_bench_c2_two_chains_graph_kernel_stats.csv"))
except (ValueError, IndexError, OSError) as e:
    print("missing: stats2", e)
tr = traffic_tables("pmc_fetch", "pmc_write", f"{tag}_bench_c2")

# ---- MFMA utilisation per kernel and for the whole step
m = counters("pmc_mfma")
rows, tot_busy, tot_act = [], 0.0, 0.0
for k, v in m.items():
    if "GRBM_GUI_ACTIVE" not in v:
        continue
    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
    act = sum(v["GRBM_GUI_ACTIVE"])
    tot_busy += busy
    tot_act += act
    n = len(v["GRBM_GUI_ACTIVE"])
    rows.append((busy, k, n, busy / n, act / n, busy / (act / NXCD * NCU * NSIMD) if act else 0.0))
rows.sort(reverse=True)
step_util = tot_busy / (tot_act / NXCD * NCU * NSIMD)
with open(os.path.join(dst, f"{tag}_bench_c2_pmc_mfma.csv"), "w") as g:
    g.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --no-cpu-baseline --no-extra --one-stream --no-graph --steps 3 --warmup 1\n")
    g.write(f"# whole run (every kernel of 4 training steps): MFMA utilisation {step_util:.4f}\n")
    g.write("kernel,dispatches,mean_SQ_VALU_MFMA_BUSY_CYCLES,mean_GRBM_GUI_ACTIVE,mfma_util\n")
    for busy, k, n, mb, ma, u in rows:
        g.write(f"\"{k[:120]}\",{n},{mb:.0f},{ma:.0f},{u:.4f}\n")


def pick(d, pattern):
    ks = [k for k in d if pattern in k]
    return ks[0] if ks else None


def kern(pattern):
    k = pick(tr["FETCH_SIZE"], pattern)
    if k is None:
        return None
    f, w = tr["FETCH_SIZE"][k], tr["WRITE_SIZE"].get(k, 0.0)
    mu = next((u for _, kk, _, _, _, u in rows if kk == k), None)
    return {"kernel": k, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "mfma_util": mu,
            "avg_us_rocprof": avg_ns.get(k, 0.0) / 1e3}


res = {"note": "FETCH_SIZE doubled (gfx950 counts half of wide coalesced reads); KB = 1024 B; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 * 4)",
       "step_mfma_util": step_util,
       "bwd_fused": kern("bwd_ws8_kernel<true, false>") or kern("bwd_ws_kernel<0, true, false"),
       "bwd_fused_unmasked_g": kern("bwd_ws_kernel<0, false, false"),
       "bwd_fused_gvec": kern("bwd_ws8_kernel<false, true>") or kern("bwd_ws_kernel<0, false, true"),
       "bwd_fused16": kern("bwd_ws16_kernel<true>"),
       "dgrad_fused": kern("conv3x3_ws_kernel<64, 64, false, false, true, false, 2, true"),
       "fwd": kern("conv3x3_ws_kernel<64, 64, true, true"),
       "wgrad": kern("wgrad_ws16_kernel<64, true, 0>")}
# round-1 keys bench.py reads
if res["bwd_fused"]:
    res["hbm_bytes_per_launch"] = res["bwd_fused"]["hbm_bytes_per_launch"]
elif res["dgrad_fused"]:
    res["hbm_bytes_per_launch"] = res["dgrad_fused"]["hbm_bytes_per_launch"]
if res["fwd"]:
    res["fwd_hbm_bytes_per_launch"] = res["fwd"]["hbm_bytes_per_launch"]
# what the measurement is valid for: bench.py drops these numbers when the kernel sources hash differently
sys.path.insert(0, ROOT)
import bench  # noqa: E402
res["kernel_sources_sha"] = bench.kernel_sources_sha()
res["collected_at_commit"] = os.popen(f"git -C {ROOT} rev-parse --short HEAD").read().strip()
json.dump(res, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)

# ---- attack kernels
try:
    aavg = stats_summary("att_stats", f"{tag}_attack_kernel_stats", 0, "python3 tools/attack_bench.py 20   (B=16, 3x256x256 f32)")
    atr = traffic_tables("att_fetch", "att_write", f"{tag}_attack")
    px = 16 * 256 * 256
    with open(os.path.join(dst, f"{tag}_attack_roofline.csv"), "w") as g:
        g.write("# per launch at B=16, 3x256x256 f32: rocprofv3 average duration, HBM bytes from the PMC passes (2*FETCH_SIZE + WRITE_SIZE), GB/s of those bytes, fraction of 8 TB/s\n")
        g.write("kernel,avg_us,hbm_MB_pmc,GBps_pmc,frac_of_8TBps\n")
        for k, ns in sorted(aavg.items(), key=lambda kv: -kv[1]):
            if k not in atr["FETCH_SIZE"] or "at::native" in k or "rocclr" in k:
                continue
            b = (2.0 * atr["FETCH_SIZE"][k] + atr["WRITE_SIZE"].get(k, 0.0)) * 1024.0
            g.write(f"\"{k[:100]}\",{ns/1e3:.1f},{b/1e6:.1f},{b/ns:.0f},{b/ns/8000.0:.3f}\n")
    shutil.copy(os.path.join(src, "attack_bench.json"), os.path.join(dst, f"{tag}_attack_bench.json"))
except (IndexError, OSError) as e:
    print("attack profiles missing:", e)
# ---- widened configurations, the literal step, the one-pass backward kernel's phases, the co-issue microbenchmark
for name, out in (("c3_c5_steps.jsonl", f"{tag}_c3_c5_steps.jsonl"), ("literal_steps.jsonl", f"{tag}_literal_steps.jsonl"), ("inn_steps.jsonl", f"{tag}_inn_steps.jsonl"),
                  ("bwd_phase_cycles.txt", f"{tag}_bwd_phase_cycles.txt"), ("mfma_coissue_micro.txt", f"{tag}_mfma_coissue_micro.txt"),
                  ("bench_512_b8.json", f"{tag}_bench_512_b8.json"), ("bench_c2.json", f"{tag}_bench_c2.json"),
                  ("bench_c2_keep_dead_grads.json", f"{tag}_bench_c2_keep_dead_grads.json"), ("bwd_sq_counters.txt", f"{tag}_bwd_sq_counters.txt"),
                  ("bwd8_phase_cycles.txt", f"{tag}_bwd8_phase_cycles.txt"), ("fwd_sq_counters.txt", f"{tag}_fwd_sq_counters.txt"),
                  ("fwd_phase_cycles.txt", f"{tag}_fwd_phase_cycles.txt"), ("two_chains.txt", f"{tag}_two_chains.txt"),
                  ("step_modes.txt", f"{tag}_step_modes.txt"), ("step_trace_one_stream.txt", f"{tag}_step_trace_one_stream.txt")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, out))
    else:
        print("missing:", name)
# the counter passes of the dominant kernel: the role-split form's (collected now) followed by the single-wave form's (kept from mid-round:
# the evidence csrc/bwd_ws8.hip's header quotes)
_sq, _sq1 = os.path.join(dst, f"{tag}_bwd_sq_counters.txt"), os.path.join(dst, f"{tag}_bwd_sq_counters_single_wave.txt")
if os.path.exists(_sq) and os.path.exists(_sq1):
    with open(_sq, "a") as g:
        g.write("\n" + open(_sq1).read())
for sub, base, steps, cmd in (("lit_stats", f"{tag}_literal_step_kernel_stats", 8.0, "python3 tools/bench_literal.py 4 bf16 6   (8 steps: the reference's literal IRNrhi step, 24 frames 256x256, bf16)"),
                              ("s512_stats", f"{tag}_bench_512_b8_kernel_stats", 13.0, "python3 bench.py --no-cpu-baseline --no-extra --one-stream --no-graph --size 512 --batch 8 --steps 10 --warmup 3   (13 steps of 8 frames 512x512, one stream)"),
                              ("inn_stats", f"{tag}_inn_step_kernel_stats", 7.0, "INN_PAR=0 python3 tools/bench_inn.py 8 bf16 4   (7 steps enqueued on one stream -- every kernel alone: the invertible embedder, 8 frames 4x256x256, embed + extract + backward + AdamW, bf16)"),
                              ("c5_stats", f"{tag}_c5_fp16_kernel_stats", 42.0, "python3 tools/bench_c5.py train_hidden_c5_fp16.yml f16 44   (42 steps with work, 16 frames 256x256 each, UNet head, f16 + device GradScaler)")):
    try:
        stats_summary(sub, base, steps, cmd)
        os.remove(os.path.join(dst, base + ".csv"))   # (the summary is what is cited; the full CSVs of the benchmarked step are kept above)
    except (ValueError, IndexError, OSError) as e:
        print("missing:", sub, e)
print(json.dumps(res, indent=1)[:1500])


# ---- the measurement table of DESIGN.md section 5 / README, generated from the files above so that the prose cannot drift from them
def _load(name):
    try:
        return json.load(open(os.path.join(dst, name)))
    except (OSError, ValueError):
        return None


def emit_measurements_md():
    b = _load(f"{tag}_bench_c2.json")
    if not b:
        print("no bench_c2.json: measurements table not written")
        return
    pm = res
    L = []
    A = L.append
    A(f"<!-- generated by tools/summarise_profiles.py {tag} from profiles/{tag}_*: do not edit by hand -->")
    A("| quantity | value (files under `profiles/`) |")
    A("|---|---|")
    A(f"| step time / throughput, C2 as `bench.py` runs it by default (two chains on two streams, replayed from a hipGraph; every {b.get('kernel_events_every')}th step enqueued on one stream with its kernels bracketed by events) | **{b['ms_per_step']:.3f} ms → {b['value']:.0f} frames/s** (`{tag}_bench_c2.json`; step events: median {b['ms_per_step_median_events']:.3f} ms, fastest {b['ms_per_step_min_events']:.3f} ms); host enqueue per replayed step {b['host_enqueue_ms_median']:.2f} ms" + " |")
    try:
        modes = open(os.path.join(dst, f"{tag}_step_modes.txt")).read().strip().splitlines()
        A("| the step's four modes, same box, interleaved, two rounds (`" + f"{tag}_step_modes.txt" + "`) | " + "<br>".join(m.replace("|", "/") for m in modes) + " |")
    except OSError:
        pass
    A(f"| FLOPs as executed / time against 2.5 PFLOP/s dense bf16 | {b['step_gflop_per_frame']:.1f} GFLOP per frame → **{100 * b['step_flops_frac_of_peak']:.1f} %** of peak |")
    rs, c4 = b.get("reference_state"), b.get("c4_shard_512")
    if rs:
        A(f"| `reference_state` (same invocation: every launch of the reference's autograd, 249.0 GFLOP per frame) | {rs['ms_per_step']:.3f} ms → {rs['value']:.0f} frames/s = {100 * rs['step_flops_frac_of_peak']:.1f} % of peak |")
    if c4:
        r4 = c4.get("roofline") or {}
        A(f"| `c4_shard_512` (same invocation: 512×512, 8 frames per GPU) | {c4['ms_per_step']:.3f} ms → {c4['value']:.0f} frames/s = {100 * c4['step_flops_frac_of_peak']:.1f} % of peak; dominant kernel {1e3 * r4.get('avg_launch_ms', 0):.1f} µs = {r4.get('frac', 0):.3f} of the HBM roof by algorithmic bytes |")
    A(f"| MFMA utilisation as the counters report it, whole step (`{tag}_bench_c2_pmc_mfma.csv`, one-stream order) | **{100 * pm['step_mfma_util']:.1f} %** |")
    bf, fw = pm.get("bwd_fused"), pm.get("fwd")
    if bf:
        alg = 4 * 16 * 256 * 256 * 64 * 2
        A(f"| dominant kernel `bwd_ws8_kernel<true,false>` (11 + 2 launches per step) | rocprofv3 {bf['avg_us_rocprof']:.1f} µs per launch (bench's own events: {1e3 * b['roofline']['avg_launch_ms']:.1f} µs on its box); algorithmic {alg / 1e6:.1f} MB → {alg / bf['avg_us_rocprof'] / 1e6:.2f} TB/s = **{alg / bf['avg_us_rocprof'] / 1e6 / 8:.3f} of 8 TB/s**; 154.6 GFLOP → {154.6188 / bf['avg_us_rocprof']:.3f} PFLOP/s = {154.6188 / bf['avg_us_rocprof'] / 2.5:.3f} of the MFMA peak; PMC traffic {bf['hbm_bytes_per_launch'] / 1e6:.1f} MB = **{bf['hbm_bytes_per_launch'] / alg:.3f} × algorithmic** (FETCH {bf['FETCH_SIZE_KB'] * 1024 / 1e6:.1f} MB × 2 + WRITE {bf['WRITE_SIZE_KB'] * 1024 / 1e6:.1f} MB); matrix pipe busy {100 * bf['mfma_util']:.1f} % |")
    if fw:
        A(f"| forward 64→64 conv (15 + 1 launches per step) | rocprofv3 {fw['avg_us_rocprof']:.1f} µs per launch; 77.3 GFLOP → {77.3094 / fw['avg_us_rocprof']:.3f} PFLOP/s = {77.3094 / fw['avg_us_rocprof'] / 2.5:.3f} of peak; PMC traffic {fw['hbm_bytes_per_launch'] / 1e6:.1f} MB = {fw['hbm_bytes_per_launch'] / 268435456.0:.3f} × algorithmic; matrix pipe busy {100 * fw['mfma_util']:.1f} % |")
    # time by kernel family from the one-stream kernel stats
    fam = collections.OrderedDict((k, 0.0) for k in ("one-pass backward 64→64 (bwd_ws8)", "forward conv 64→64", "other convolutions / input gradients", "image-fed first layers (forward, weight gradient, one-pass backward)",
                                                     "weight gradients outside the one-pass kernel", "slab reductions", "BatchNorm finalisations", "pools, heads (1×1, pooled head)", "concat side kernels", "rest (layout, attack, losses, Adam, packs, copies)"))
    tot = 0.0
    for r in csv.DictReader(open(os.path.join(dst, f"{tag}_bench_c2_kernel_stats.csv"))):
        n, ms = r["Name"], float(r["TotalDurationNs"]) / 1e6 / 13.0
        tot += ms
        if "bwd_ws8" in n: k = "one-pass backward 64→64 (bwd_ws8)"
        elif "conv3x3_ws_kernel<64, 64, true, true" in n: k = "forward conv 64→64"
        elif "conv3x3_ws_kernel<16" in n or "wgrad_ws16_kernel<16" in n or "bwd_ws16" in n: k = "image-fed first layers (forward, weight gradient, one-pass backward)"
        elif "conv3x3_ws_kernel" in n or "conv3x3_kernel" in n: k = "other convolutions / input gradients"
        elif "wgrad_ws16_kernel" in n or "wgrad_kernel" in n: k = "weight gradients outside the one-pass kernel"
        elif "wgrad_reduce" in n: k = "slab reductions"
        elif "finalize" in n or "tree_reduce" in n or "colsum" in n: k = "BatchNorm finalisations"
        elif "avgpool" in n or "head" in n: k = "pools, heads (1×1, pooled head)"
        elif "concat_side" in n or "msg_" in n or "dy_" in n or "side_pack" in n: k = "concat side kernels"
        else: k = "rest (layout, attack, losses, Adam, packs, copies)"
        fam[k] += ms
    A(f"| time by kernel family (ms per step under rocprofv3, one-stream order, {tot:.2f} ms of kernels: `{tag}_bench_c2_kernel_stats_summary.txt`) | " + " · ".join(f"{k} {v:.2f}" for k, v in fam.items()) + " |")
    cb = b.get("cpu_baseline")
    if cb:
        A(f"| CPU baseline (`cpu_baseline`) | {cb['value']:.2f} frames/s on {cb['cores']} threads ({cb['sample']}) |")
    # the widened configurations (tools/collect_profiles.sh b)
    def _jsonl(name):
        try:
            return [json.loads(l) for l in open(os.path.join(dst, name)) if l.startswith("{")]
        except OSError:
            return []
    cc = _jsonl(f"{tag}_c3_c5_steps.jsonl")
    if cc:
        A(f"| C3 / C5 through the model surface (`feed_data` of pinned host batches + `optimize_parameters`), attack cycling with the step (`{tag}_c3_c5_steps.jsonl`; wall clock of the whole loop per step, and the median of the steps' own event times) | " + "<br>".join(
            f"{r['yml']} {r['dtype']}{', deferred logs' if r.get('deferred_logs') else ''}: wall {r.get('wall_ms_per_step', float('nan')):.2f} ms → {r.get('wall_frames_per_s', float('nan')):.0f} frames/s; step events {r['ms_per_step_median']:.2f} ms (slowest attack {max(r['ms_per_step_by_attack'].items(), key=lambda kv: kv[1])[0]} {max(r['ms_per_step_by_attack'].values()):.2f} ms)" for r in cc) + " |")
    lt = _jsonl(f"{tag}_literal_steps.jsonl")
    if lt:
        A(f"| the literal IRNrhi step (generator + localizer + discriminator, `{tag}_literal_steps.jsonl`) | " + "<br>".join(
            f"{r['dtype']}, {r['frames_per_step']} frames: {r['ms_per_step_median']:.1f} ms = {r['tflops']:.0f} TFLOP/s ({100 * r['flops_frac_of_mfma_peak']:.1f} % of peak)" for r in lt) + " |")
    inn = _jsonl(f"{tag}_inn_steps.jsonl")
    if inn:
        A(f"| the invertible embedder's step (`{tag}_inn_steps.jsonl`) | " + "<br>".join(
            f"{r['dtype']}, {r['frames_per_step']} frames, {'replayed' if r.get('graph') else 'enqueued'}{'' if r.get('parallel_subnets', True) else ', the subnets on one stream'}: {r['ms_per_step_median']:.1f} ms = {r['tflops']:.0f} TFLOP/s ({100 * r['flops_frac_of_mfma_peak']:.1f} % of peak)" for r in inn) + " |")
    try:
        rows = [r for r in csv.DictReader(l for l in open(os.path.join(dst, f"{tag}_attack_roofline.csv")) if not l.startswith("#"))]
        def short(n):
            m = re.search(r"::(\w+<[^>]*>|\w+)\(", n)
            return m.group(1) if m else n[:40]
        A(f"| attack kernels at B=16, 3×256×256 f32, per launch (`{tag}_attack_roofline.csv`: rocprofv3 average, PMC bytes, fraction of 8 TB/s) | " + "<br>".join(
            f"`{short(r['kernel'])}` {float(r['avg_us']):.1f} µs, {float(r['hbm_MB_pmc']):.1f} MB → {float(r['frac_of_8TBps']):.3f}" for r in rows[:12]) + " |")
    except (OSError, KeyError, ValueError):
        pass
    text = "\n".join(L) + "\n"
    open(os.path.join(dst, f"{tag}_measurements.md"), "w").write(text)
    print("wrote", f"profiles/{tag}_measurements.md")
    # DESIGN.md carries the table between two markers; refresh it in place so that the prose cannot drift from the committed files
    dpath = os.path.join(os.path.dirname(dst), "DESIGN.md")
    d = open(dpath).read()
    b0, b1 = f"<!-- BEGIN {tag}_measurements -->", f"<!-- END {tag}_measurements -->"
    if b0 in d and b1 in d:
        d = d[:d.index(b0) + len(b0)] + "\n" + text + d[d.index(b1):]
        open(dpath, "w").write(d)
        print("refreshed the table in DESIGN.md")


emit_measurements_md()
