"""ON THE GPU BOX: where do the step's device-to-device copies (`__amd_rocclr_copyBuffer`, 15 per step in profiles/r03_bench_c2_kernel_stats)
and its torch-side kernels come from?  One benchmarked step under torch.profiler with Python stacks; prints, per aten op that launches a
copy / fill / elementwise kernel, the innermost frames of this package.    python tools/find_copies.py [--keep-dead-grads]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_watermarking_forgery_detection_amd.hidden_models import Hidden            # noqa: E402
from video_watermarking_forgery_detection_amd import noise_layers as NL              # noqa: E402
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration     # noqa: E402


def main():
    dev = torch.device("cuda")
    S, B = 256, 16
    torch.manual_seed(10)
    h = Hidden(HiDDenConfiguration(H=S, W=S), dev, NL.Jpeg(50), None, compute_dtype=torch.bfloat16,
               keep_dead_discriminator_grads="--keep-dead-grads" in sys.argv)
    images = torch.rand(B, 3, S, S, device=dev)
    messages = torch.randint(0, 2, (B, 30), device=dev).float()
    for _ in range(3):
        h.train_on_batch([images, messages])
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        h.train_on_batch([images, messages])
        torch.cuda.synchronize()
    pkg = "video_watermarking_forgery_detection_amd"
    seen = {}
    for ev in prof.events():
        name = ev.name
        if not name.startswith("aten::"):
            continue
        if name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::select", "aten::slice", "aten::empty_strided",
                    "aten::reshape", "aten::detach", "aten::alias", "aten::_unsafe_view", "aten::permute", "aten::unsqueeze", "aten::squeeze",
                    "aten::result_type", "aten::lift_fresh", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::is_pinned", "aten::set_"):
            continue
        stack = [f for f in (ev.stack or []) if pkg in f or "bench.py" in f or "find_copies" in f]
        key = (name, tuple(stack[:3]))
        seen[key] = seen.get(key, 0) + 1
    for (name, stack), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d} x {name}")
        for f in stack:
            print("        ", f)
    print("---- device-side (kernel / memcpy) events of the step, by name")
    kn = {}
    for ev in prof.events():
        if ev.device_type is not None and "cuda" in str(ev.device_type).lower():
            kn[ev.name] = kn.get(ev.name, 0) + 1
    for k, n in sorted(kn.items(), key=lambda kv: -kv[1]):
        if "Memcpy" in k or "Memset" in k or "at::" in k or "rocclr" in k:
            print(f"{n:3d} x {k[:140]}")


if __name__ == "__main__":
    main()
