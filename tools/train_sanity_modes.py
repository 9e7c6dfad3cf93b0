"""ON THE GPU BOX: the same training run (fixed seeds, fresh random batches from a seeded device generator) in the step's four modes -- enqueued
on one stream / as two chains, each also replayed from a hipGraph.  Every mode must give the SAME losses and parameters, bit for bit, at
every checkpoint.   usage: python tools/train_sanity_modes.py [size=128] [steps=300] [Identity|Jpeg50|JpegSS50|JpegMask50|GaussianBlur|MiddleBlur3] [batch=16]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                                   # noqa: E402
from video_watermarking_forgery_detection_amd.hidden_models import Hidden                      # noqa: E402
from video_watermarking_forgery_detection_amd import noise_layers as NL                        # noqa: E402
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration               # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kind = sys.argv[3] if len(sys.argv) > 3 else "Identity"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda", 0)


def digest(h):
    m = hashlib.sha256()
    for net in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
        for t in net.state_dict().values():
            m.update(t.detach().float().cpu().numpy().tobytes())
    return m.hexdigest()[:12]


results = {}
for mode in ("one stream", "two chains", "one stream + graph", "two chains + graph"):
    torch.manual_seed(10)
    noise = {"Identity": NL.Identity, "Jpeg50": lambda: NL.Jpeg(50), "JpegSS50": lambda: NL.JpegSS(50), "JpegMask50": lambda: NL.JpegMask(50),
             "GaussianBlur": NL.GaussianBlur, "MiddleBlur3": lambda: NL.MiddleBlur(3)}[kind]()
    h = Hidden(HiDDenConfiguration(H=S, W=S), dev, noise, None, compute_dtype=torch.bfloat16)
    h.two_streams = "two" in mode
    if "graph" in mode:
        h.enable_graph()
    g = torch.Generator(device="cuda").manual_seed(1)
    trace = []
    for it in range(N):
        images = torch.rand(B, 3, S, S, device=dev, generator=g)
        messages = torch.randint(0, 2, (B, 30), device=dev, generator=g).float()
        losses, _ = h.train_on_batch([images, messages])
        if it in (0, 1, 2, 3, 5, 10, 20, 50, 100, 200) or it == N - 1:
            trace.append((it, round(dict((k.strip(), v) for k, v in losses.items())["loss"], 6), digest(h)))
    results[mode] = trace
    print(mode, trace[-1], flush=True)
# (under an attack whose gradient is identically zero -- Jpeg's rounding -- the two-chain order sums the decoder's first weight gradient in another
#  kernel than the one-stream order: those two are compared within tolerance by tests/test_gpu_graph.py; a replayed run must equal ITS enqueued run)
for mode, base in (("two chains", "one stream"), ("one stream + graph", "one stream"), ("two chains + graph", "two chains")):
    tr, ref = results[mode], results[base]
    first = next((a for a, b in zip(tr, ref) if a != b), None)
    print(f"{mode:20s} vs {base:12s}:", "identical at every checkpoint" if first is None else f"FIRST DIFFERENCE at step {first[0]}: {first} vs {[b for b in ref if b[0] == first[0]][0]}")
