#!/usr/bin/env python3
"""bench.py -- frames/sec of the watermark embed -> JPEG -> decode GAN training step (BASELINE.json
metric) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1] -- 256x256, batch 16 PER GPU (weak scaling),
HiDDeN order (hidden_models/hidden.py:54-118): D(cover) fwd/bwd, encoder -> Jpeg(50) -> decoder,
D(enc) fwd/bwd, Adam(D), D(enc) + MSE + MSE fwd/bwd, Adam(enc+dec); L=30, encoder 4x64,
decoder 7x64, discriminator 3x64; bf16 activations, f32 accumulate/params; synthetic data,
random-init weights; inputs resident in HBM before the timed region.  One step = one batch.

Launches the reference's autograd runs whose results nothing can observe are not launched by default, and their FLOPs are not counted
(config.workload says which; `step_gflop_per_frame`): (1) round 3 -- g_loss.backward() (hidden.py:101) also accumulates the generator loss's
gradients into the DISCRIMINATOR's parameters, which no optimiser step uses and hidden.py:67 zeroes unread
(Hidden(keep_dead_discriminator_grads=False): 2 x 4.832 + 0.226 GFLOP per 256x256 frame); (2) round 4 -- Jpeg(Q) quantises with torch.round,
whose gradient is identically zero (noise_layers/jpeg.py:226-240), so the decoder's gradient wrt its input is multiplied by zero
(Hidden.skip_zero_attack_gradient: 0.226 GFLOP).  Every loss, output and parameter update is identical (tests/test_gpu_configs.py,
tests/test_gpu_graph.py).  The SAME invocation times the step with every launch of the reference's autograd (`reference_state`, 249.0 GFLOP
per frame) and the 512 x 512 / 8-frame shard of BASELINE's C4 (`c4_shard_512`); --keep-dead-grads makes the former the headline.

Round 4, one GPU: the step runs as its two independent chains on two streams (Hidden.two_streams) and is replayed from a hipGraph
(Hidden.enable_graph) -- bit-identical to the eager one-stream step; --one-stream / --no-graph switch either off.  The steps whose kernels
are bracketed with events (the last of every --kernel-events-every) run on one stream, eagerly, so that an event pair times the kernel
alone.  With a gradient all-reduce (N > 1) the step runs eagerly on one stream as in round 3.

The JSON line carries, besides the driver contract:
  roofline     -- the dominant kernel (most time per step, 13 launches): bwd_ws8_kernel (csrc/bwd_ws8.hip; csrc/bwd_ws.hip for shapes that
                  are not whole 8x16 tiles) = the whole backward of a 64->64 body layer in one pass: input gradient with the BatchNorm-backward apply fused, the feeding layer's
                  BatchNorm sums, and the weight gradient, from one staged dy / activation tile.  Its 154.6 GFLOP ride on 4 tensor
                  passes (reads g, y, y of the layer below; writes dx = 537 MB algorithmic per launch at B=16 256x256:
                  288 FLOP/B, just under the 312 FLOP/B ridge), so the bound is HBM: achieved = algorithmic bytes / launch
                  duration measured live with events on the launch stream inside the timed region (every launch of every
                  `kernel_events_every`-th timed step: an event pair costs ~11 us of idle GPU around its launch); peak = 8 TB/s; `traffic` =
                  HBM bytes / launch from the committed rocprofv3 PMC passes of this round (profiles/r04_pmc_traffic.json;
                  FETCH_SIZE doubled as the gfx950 guide says); `mfma_util_pmc` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x
                  1024 SIMDs) from the committed counter pass (profiles/r04_bench_c2_pmc_mfma.csv); `step_mfma_util_pmc` = the same
                  ratio over every kernel of the step.  Those three come from a FILE, not from this run: the file carries the sha256
                  of the kernel sources it was measured on (csrc/*, build.py), and when the sources here hash differently the three
                  are null and `pmc_stale` is true.  (Until this kernel the work was two launches -- the fused input gradient,
                  671 MB, and the weight gradient, 268 MB: 939 MB for the same result.)
  roofline_mfma -- the runner-up, conv3x3_ws_kernel<64,64,XFORM,STATS,M16> (forward 64->64 conv, fused BN+ReLU input and
                  BatchNorm statistics, 15 launches / step): 77.3 GFLOP per launch against 2.5 PFLOP/s dense bf16
  cpu_baseline -- the oracle (oracle/hidden_ref.py, torch CPU fp32, all host cores) on a bounded
                  sample of the same step (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # warm-up 10, time 50
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--noise", default="Jpeg50")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-events-every", type=int, default=20,
                    help="the roofline kernels are bracketed with HIP events on every n-th timed step only: an event pair idles the GPU for ~11 us "
                         "around the launch it brackets (28 bracketed launches = 0.31 ms of a 5.5 ms step when every step is instrumented)")
    ap.add_argument("--keep-dead-grads", action="store_true", help="also compute the discriminator weight gradients of the generator pass (the reference's state; nothing reads them)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step from a hipGraph (Hidden.enable_graph; one GPU only: with a gradient all-reduce the step runs eagerly); "
                         "the steps whose kernels are bracketed with events (every --kernel-events-every-th) are enqueued eagerly")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--two-streams", dest="two_streams", action="store_true", default=None,
                    help="the step's two independent chains (discriminator passes | encoder -> attack -> decoder) on two streams (Hidden.two_streams)")
    ap.add_argument("--one-stream", dest="two_streams", action="store_false")
    ap.add_argument("--no-extra", action="store_true", help="skip the two secondary timed regions (reference_state, c4_shard_512)")
    ap.add_argument("--extra-steps", type=int, default=25, help="timed steps of each secondary region (after 5 warm-up steps)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames per step of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (median reported) after one warm-up step")
    return ap.parse_args()


def _physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; None when the file does not say"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(size, frames, steps=3):
    """oracle step (torch CPU fp32) on `frames` frames: the reported CPU baseline ("port"), SURVEY §8d: one warm-up step,
    then the median of `steps` timed steps of the same HiDDeN-order step on the host cores of the GPU box."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import hidden_ref, jpeg_ref
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    # torch's CPU convolution stops scaling past a few dozen threads (256 threads measured 20x slower than 32 on the
    # GPU box's host), so the baseline runs on min(32, usable) threads and says so
    cores = min(32, usable)
    torch.set_num_threads(cores)
    torch.manual_seed(10)
    cfg = hidden_ref.HiDDenConfiguration(H=size, W=size)
    ref = hidden_ref.HiddenRef(cfg, lambda x: jpeg_ref.jpeg_layer(x, 50, "round"))
    images = torch.rand(frames, 3, size, size)
    messages = torch.randint(0, 2, (frames, 30)).float()
    wf = max(1, frames // 4)
    ref.train_on_batch(images[:wf], messages[:wf])     # warm-up (allocator, thread pool, oneDNN primitives) on a quarter batch
    ts = []
    for _ in range(max(1, steps)):
        t0 = time.time()
        ref.train_on_batch(images, messages)
        ts.append(time.time() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "logical_cores": logical, "usable_cores": usable, "physical_cores": _physical_cores(), "cpu_model": _cpu_model(),
            "step_seconds": [round(t, 3) for t in ts],
            "sample": f"median of {len(ts)} steps (after 1 warm-up step on {wf} frames) of the same HiDDeN-order step on {frames} frames "
                      f"{size}x{size} (oracle/hidden_ref.py, torch {torch.__version__} CPU fp32, {cores} threads, median {dt:.1f} s)"}


def attack_roofline(timer, B, S, name):
    """the attack kernels (HBM-bound, SURVEY §8d: 24 B/px forward = 12 read + 12 written at f32 NCHW I/O; block-JPEG backward:
    Jpeg (torch.round: the gradient is identically zero, the kernel only writes it) 12 B/px, JpegMask 24 B/px (gradient in,
    gradient out), JpegSS 36 B/px (it also re-reads x for round_ss')) timed live like the conv kernels: events on the launch
    stream around every launch of the timed region"""
    out = {}
    px = float(B * S * S)
    for key, bpp in (("jpeg_fwd", 24.0), ("jpeg_bwd", 36.0 if name.startswith("JpegSS") else 24.0 if name.startswith("JpegMask") else 12.0)):
        ms = timer.elapsed_ms(key)
        if not ms:
            continue
        avg = sum(ms) / len(ms)
        gbs = bpp * px / (avg * 1e-3) / 1e9
        out[key] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "avg_launch_ms": avg,
                    "launches_timed": len(ms), "algorithmic_bytes_per_launch": bpp * px}
    return out or None


def kernel_sources_sha():
    """sha256 over the kernel sources and the build recipe: what a committed PMC measurement is valid for"""
    import hashlib
    pkg = os.path.join(ROOT, "video_watermarking_forgery_detection_amd")
    h = hashlib.sha256()
    csrc = os.path.join(pkg, "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(pkg, "build.py"), "rb").read())
    return h.hexdigest()[:16]


PMC_FILE = "r04_pmc_traffic.json"
def graph_failures(h):
    """reasons of failed step captures (Hidden falls back to enqueueing those steps eagerly and warns), [] if none"""
    return [g.failed for g in (h._graphs or {}).values() if g.failed is not None]


GRAPH_DEFAULT = True          # --graph / --no-graph: the step replayed from a hipGraph (one GPU; N > 1 runs eagerly around the all-reduces)
TWO_STREAMS_DEFAULT = True    # --two-streams / --one-stream: the step's two independent chains on two streams
_pmc = {}


def pmc_traffic(args, S, B, key="hbm_bytes_per_launch"):
    """HBM bytes per launch of the dominant kernel (key: of the runner-up), from this round's committed PMC passes: same workload, same
    kernel sources only (else None)."""
    if args.dtype != "bf16" or S != 256 or B != 16 or args.keep_dead_grads or args.noise != "Jpeg50":
        return None
    if not _pmc:
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
                _pmc["d"] = json.load(f)
        except (OSError, ValueError):
            _pmc["d"] = None
        _pmc["stale"] = _pmc["d"] is None or _pmc["d"].get("kernel_sources_sha") != kernel_sources_sha()
    if _pmc["stale"]:
        return None
    d = _pmc["d"]
    try:
        for part in key.split("."):
            d = d[part]
        return d
    except (KeyError, TypeError):
        return None


def file_sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def timed_steps(h, batch, steps, warmup, barrier, world, dev, timer=None, every=5):
    """`warmup` untimed steps, then `steps` steps bracketed like the headline region (barrier + synchronize on both sides, wall clock, MAX
    over ranks) -> seconds.  timer: an ops.KernelTimer switched on for every `every`-th timed step"""
    import torch.distributed as dist
    from video_watermarking_forgery_detection_amd import ops
    ts = h.two_streams
    for w in range(warmup):
        cold = w == 0 and timer is not None     # (as the headline loop's warm-up: one step on the bracketed steps' path)
        if cold:
            ops.set_kernel_timer(ops.KernelTimer(lambda name, i: name == "conv3x3_bwd_fused"))
            h.two_streams = False
        h.train_on_batch(batch)
        if cold:
            ops.set_kernel_timer(None)
            h.two_streams = ts
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        bracket = timer is not None and ((i + 1) % every == 0 or (steps < every and i == steps - 1))
        ops.set_kernel_timer(timer if bracket else None)
        h.two_streams = ts and not bracket      # (a bracketed step runs its kernels one after the other: see the headline loop)
        h.train_on_batch(batch)
    barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    h.two_streams = ts
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def zero_attack(args):
    """does the configured attack pass back an identically zero gradient (Jpeg: torch.round)?"""
    return args.noise.startswith("Jpeg") and not args.noise.startswith(("JpegSS", "JpegMask"))


def step_gflop_per_frame(args, S, reference_state=False):
    """GFLOP per frame AS EXECUTED.  The reference autograd's step is 249.0 GFLOP per 256x256 frame; not executed, hence not
    counted, unless --keep-dead-grads / reference_state: (1) the generator pass's three discriminator weight-gradient GEMMs, whose results
    nothing reads (2 x 4.832 + 0.226); (2) under an attack with an identically zero gradient (Jpeg), the decoder's gradient wrt its
    input image (the 3 <- 64 input-gradient GEMM of its first layer, 0.226), whose result is multiplied by zero"""
    g = 249.0
    if not (reference_state or args.keep_dead_grads):
        g -= 2 * 4.8318 + 0.2265
        if zero_attack(args):
            g -= 0.2265
    return g * (S / 256.0) ** 2


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N fresh ranks with torch.distributed.run as a CHILD process --
    this parent has not touched the GPU (no HIP call, no torch.cuda.is_available()) and never replaces itself -- and pass the
    child's exit code on.  Rank 0 of the children prints the one JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU; the modulo only matters when the multi-process flow is rehearsed on a box with fewer GPUs
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm; WM_DIST_BACKEND=gloo only rehearses the multi-process flow on a one-GPU box
        dist.init_process_group(os.environ.get("WM_DIST_BACKEND", "nccl"))

    import video_watermarking_forgery_detection_amd as wm
    from video_watermarking_forgery_detection_amd import _lib as _wm_lib, ops
    from video_watermarking_forgery_detection_amd.distributed import GradSync, broadcast_parameters
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration

    dev = torch.device("cuda", local_rank)
    S, B = args.size, args.batch
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    kind = "".join(c for c in args.noise if not c.isdigit())
    Q = int("".join(c for c in args.noise if c.isdigit()) or 50)
    noise = {"Jpeg": NL.Jpeg, "JpegSS": NL.JpegSS, "JpegMask": NL.JpegMask}[kind](Q) if kind != "Identity" else NL.Identity()

    torch.manual_seed(10)  # identical initial weights on every rank (plus the DDP-style broadcast below)
    cfg = HiDDenConfiguration(H=S, W=S)
    sync = GradSync(profile=True) if world > 1 else None
    h = Hidden(cfg, dev, noise, None, compute_dtype=dtype, grad_sync=sync, keep_dead_discriminator_grads=args.keep_dead_grads)
    h.skip_zero_attack_gradient = not args.keep_dead_grads
    broadcast_parameters([h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator])
    two_streams = (bool(args.two_streams) if args.two_streams is not None else TWO_STREAMS_DEFAULT) and world == 1
    h.two_streams = two_streams
    use_graph = bool(args.graph) if args.graph is not None else GRAPH_DEFAULT
    use_graph = use_graph and world == 1
    if use_graph:
        h.enable_graph()
    torch.manual_seed(10 + rank)  # rank r draws its shard with seed 10+r
    images = torch.rand(B, 3, S, S, device=dev)
    messages = torch.randint(0, 2, (B, 30), device=dev).float()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        # the first warm-up step takes the path of the timed region's bracketed steps (one stream, eagerly, kernel events on): its buffers come
        # from the main stream's pool, which the replayed / two-chain steps never touch -- left cold, the first bracketed step of the timed
        # region pays the allocator's hipMallocs (~10 ms, once)
        bracket = w == 0 and (use_graph or two_streams)
        if bracket:
            ops.set_kernel_timer(ops.KernelTimer(lambda name, i: name in ("conv3x3_bwd_fused", "conv3x3_fwd", "jpeg_fwd", "jpeg_bwd")))
            h.two_streams = False
        h.train_on_batch([images, messages])
        if bracket:
            ops.set_kernel_timer(None)
            h.two_streams = two_streams
    if sync is not None:
        sync.report()   # (drop the warm-up's entries)
    # the two heaviest kernels: the one-pass backward of the 64->64 body layers (13 launches / step, 16-bit activations only) and the
    # forward 64->64 conv with fused BN+ReLU input transform (15 launches / step)
    timer = ops.KernelTimer(lambda name, i: (name == "conv3x3_bwd_fused" and not i["gvec"]) or
                            (name == "conv3x3_fwd" and i["Cin"] == 64 and i["CoutP"] == 64 and i["xform"]) or
                            name in ("jpeg_fwd", "jpeg_bwd"))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step events on the launch stream (§8d: median)
    every = max(1, args.kernel_events_every)
    barrier()
    t0 = time.perf_counter()
    host_ms, host_ms_eager = [], []
    for i in range(args.steps):
        # The kernel events are part of the timed region, and they are not free: rocprofv3's kernel trace shows 5.3-6.1 us of idle GPU on
        # either side of every bracketed launch (the event's marker packet) and none around the launches that are not bracketed -- so the
        # roofline kernels are bracketed on every `every`-th timed step, all their launches of that step.
        # (the LAST step of every `every`, not the first: the first steps behind the barrier + synchronize run on a chip whose clock is still
        # ramping up, and an eagerly enqueued step there also starts from an empty queue -- 6.2 ms instead of 5.5)
        bracket = (i + 1) % every == 0 or (args.steps < every and i == args.steps - 1)
        ops.set_kernel_timer(timer if bracket else None)
        # a bracketed step runs its launches ONE AFTER THE OTHER (one stream, eagerly): beside a launch of the other chain a kernel's event
        # pair would time the sharing of the chip, not the kernel (178 us instead of 149 for the dominant one)
        h.two_streams = two_streams and not bracket
        if sync is not None:
            sync.profile = bracket    # (the same for the events around the gradient all-reduces' waits)
        marks[i].record()
        th = time.perf_counter()
        losses, _ = h.train_on_batch([images, messages])
        (host_ms_eager if (use_graph and bracket) else host_ms).append(1e3 * (time.perf_counter() - th))
    marks[args.steps].record()
    host_ms.sort(); host_ms_eager.sort()
    if not host_ms:
        host_ms = host_ms_eager
    barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    h.two_streams = two_streams
    sync_rep = sync.report(max(1, args.steps // every)) if sync is not None else None
    if sync_rep is not None:   # every rank: its buckets and how long its compute stream stood waiting for them (the exposed part of the all-reduces)
        sys.stderr.write(f"[bench rank {rank}] grad sync: {json.dumps(sync_rep)}\n")
    step_ms_order = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    step_ms = sorted(step_ms_order)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kms = timer.elapsed_ms("conv3x3_fwd")
    dms = timer.elapsed_ms("conv3x3_bwd_fused")
    # ---- two more timed regions of the SAME invocation (secondary keys of the line; north_star asks for 256x256 AND 512x512, and the headline
    # workload drops three dead weight-gradient GEMMs the reference executes), bracketed exactly like the headline region; one GPU only
    extra = {}
    if sync is not None:
        sync.profile = False
    peak_tf = 2500.0 if dtype == torch.bfloat16 else 157.3
    run_extra = not args.no_extra and world == 1   # (N > 1: the scaling runs time the headline region only)
    if run_extra and not args.keep_dead_grads and S == 256:
        # (1) the reference's exact .grad state: g_loss.backward() (hidden.py:101) also leaves its gradients in the discriminator's parameters
        h.keep_dead_discriminator_grads = True
        h.skip_zero_attack_gradient = False
        dte = timed_steps(h, [images, messages], args.extra_steps, 5, barrier, world, dev)
        h.keep_dead_discriminator_grads = False
        h.skip_zero_attack_gradient = True
        gf = 249.0
        extra["reference_state"] = {
            "workload": f"the same step with EVERY launch of the reference's autograd: the discriminator's dead weight gradients of the generator pass (the reference's .grad state, hidden.py:67,101) and the decoder's input gradient under Jpeg's zero-gradient rounding, {S}x{S}, batch {B}/GPU",
            "steps": args.extra_steps, "warmup": 5, "ms_per_step": 1e3 * dte / args.extra_steps, "value": world * B * args.extra_steps / dte, "unit": "frames/s",
            "step_gflop_per_frame": gf, "step_flops_frac_of_peak": gf * 1e9 * B * args.extra_steps / dte / (peak_tf * 1e12)}
    if run_extra and S == 256 and dtype == torch.bfloat16:
        # (2) BASELINE.json configs[3]'s per-GPU shard: 512x512, 8 frames per GPU
        S2, B2 = 512, 8
        torch.manual_seed(10)
        h2 = Hidden(HiDDenConfiguration(H=S2, W=S2), dev, noise, None, compute_dtype=dtype, grad_sync=sync, keep_dead_discriminator_grads=args.keep_dead_grads)
        h2.skip_zero_attack_gradient = not args.keep_dead_grads
        broadcast_parameters([h2.encoder_decoder.encoder, h2.encoder_decoder.decoder, h2.discriminator])
        h2.two_streams = two_streams
        if use_graph:
            h2.enable_graph()
        torch.manual_seed(10 + rank)
        im2 = torch.rand(B2, 3, S2, S2, device=dev)
        ms2 = torch.randint(0, 2, (B2, 30), device=dev).float()
        timer2 = ops.KernelTimer(lambda name, i: name == "conv3x3_bwd_fused" and not i["gvec"])
        dt2 = timed_steps(h2, [im2, ms2], args.extra_steps, 5, barrier, world, dev, timer2, every)
        d2 = timer2.elapsed_ms("conv3x3_bwd_fused")
        gf2 = step_gflop_per_frame(args, S2)
        c4 = {"workload": f"C4's per-GPU shard: the headline step at {S2}x{S2}, batch {B2}/GPU", "steps": args.extra_steps, "warmup": 5,
              "ms_per_step": 1e3 * dt2 / args.extra_steps, "value": world * B2 * args.extra_steps / dt2, "unit": "frames/s",
              "step_gflop_per_frame": gf2, "step_flops_frac_of_peak": gf2 * 1e9 * B2 * args.extra_steps / dt2 / (peak_tf * 1e12)}
        if d2:
            a2 = sum(d2) / len(d2)
            by2 = 4.0 * B2 * S2 * S2 * 64 * 2
            c4["roofline"] = {"bound": "hbm", "kernel": "bwd_ws8_kernel", "achieved": by2 / (a2 * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": by2 / (a2 * 1e-3) / 1e9 / 8000.0, "avg_launch_ms": a2, "launches_timed": len(d2), "algorithmic_bytes_per_launch": by2}
        extra["c4_shard_512"] = c4
        del h2, im2, ms2
    if sync is not None:
        sync.report()
    if rank == 0:
        fps = world * B * args.steps / dt
        esz = 2 if dtype == torch.bfloat16 else 4
        tensor_bytes = float(B * S * S * 64 * esz)
        flops_per_launch = 2.0 * B * S * S * 64 * 9 * 64
        avg_ms = sum(kms) / max(1, len(kms))
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        peak = 2500.0 if dtype == torch.bfloat16 else 157.3
        mfma = {"bound": "mfma",
                "kernel": ("conv3x3_ws_kernel<64,64,XFORM,STATS,M16>" if dtype == torch.bfloat16 else "conv3x3_kernel<float,64,true>") + " (forward 64->64 implicit GEMM, fused BN+ReLU input, BN statistics)",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": pmc_traffic(args, S, B, "fwd_hbm_bytes_per_launch"), "mfma_util_pmc": pmc_traffic(args, S, B, "fwd.mfma_util"),
                "launches_timed": len(kms), "avg_launch_ms": avg_ms,
                "flops_per_launch": flops_per_launch, "hbm_algorithmic_bytes_per_launch": 2.0 * tensor_bytes}
        # FLOPs per frame as executed: the reference autograd's 249.0 GFLOP at 256x256 minus, unless --keep-dead-grads, the
        # two 64->64 and the 3->64 discriminator weight-gradient GEMMs of the generator pass (2 x 4.832 + 0.226 GFLOP)
        step_gflop = step_gflop_per_frame(args, S)
        if dms:   # bf16: the one-pass backward kernel is the dominant one; 288 FLOP/B sits just under the ridge: HBM-bound
            davg = sum(dms) / len(dms)
            dbytes = 4.0 * tensor_bytes   # reads g, y, y of the layer below; writes dx
            dflops = 2.0 * flops_per_launch   # the input gradient and the weight gradient
            roof = {"bound": "hbm",
                    "kernel": "bwd_ws8_kernel (64->64 body layer, one pass: input gradient + BatchNorm-backward apply + the feeding layer's BatchNorm sums + weight gradient; the two GEMMs on different waves)",
                    "achieved": dbytes / (davg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": dbytes / (davg * 1e-3) / 1e9 / 8000.0,
                    "traffic": pmc_traffic(args, S, B), "mfma_util_pmc": pmc_traffic(args, S, B, "bwd_fused.mfma_util"),
                    "launches_timed": len(dms), "avg_launch_ms": davg,
                    "algorithmic_bytes_per_launch": dbytes, "flops_per_launch": dflops,
                    "flops_frac_of_mfma_peak": dflops / (davg * 1e-3) / 1e12 / peak}
        else:
            roof = mfma
        out = {
            "metric": f"frames/sec training step (embed->JPEG->decode), {S}x{S}",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C2: HiDDeN GAN step enc(4x64)->{args.noise}->dec(7x64)+disc(3x64), L=30, {S}x{S}, batch {B}/GPU"
                                   + ("" if args.keep_dead_grads else "; not launched (results identical): the generator pass's dead discriminator weight gradients"
                                      + (", the decoder's input gradient that Jpeg's zero-gradient rounding multiplies by 0" if zero_attack(args) else "")
                                      + " -- reference_state times the step with them"),
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "ms_per_step_median_events": step_ms[len(step_ms) // 2], "ms_per_step_min_events": step_ms[0], "ms_per_step_max_events": step_ms[-1],
            "graph": use_graph and not graph_failures(h), "graph_capture_failed": graph_failures(h) or None, "two_streams": bool(two_streams),
            "kernel_events_note": "the steps whose kernels are bracketed with events run on one stream, eagerly (every kernel alone on the chip): avg_launch_ms is the kernel's own duration; the other steps run as two chains" + (" replayed from a hipGraph" if use_graph else "") if two_streams or use_graph else None,
            "host_enqueue_ms_median": host_ms[len(host_ms) // 2], "host_enqueue_ms_max": host_ms[-1],
            "host_enqueue_ms_median_eager_steps": host_ms_eager[len(host_ms_eager) // 2] if host_ms_eager else None,
            "kernel_events_every": every, "ms_per_step_events": [round(v, 3) for v in step_ms_order],
            "roofline": roof,
            "roofline_mfma": mfma,
            "roofline_attack": attack_roofline(timer, B, S, args.noise),
            "step_mfma_util_pmc": pmc_traffic(args, S, B, "step_mfma_util"),
            "step_gflop_per_frame": step_gflop,
            "step_flops_frac_of_peak": (step_gflop * 1e9 * world * B * args.steps / dt) / (peak * 1e12 * world),
            "pmc_stale": bool(_pmc.get("stale", True)),
            "kernel_sources_sha": kernel_sources_sha(), "library": os.path.relpath(_wm_lib.loaded_path(), ROOT), "library_sha16": file_sha16(_wm_lib.loaded_path()),
            "grad_sync": sync_rep,
            "last_losses": {k.strip(): v for k, v in losses.items()},
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, args.cpu_frames, args.cpu_steps)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
