"""per-(op, shape) time table of the literal IRNrhi step (or, with a leading `inn`, of tools/bench_inn.py's invertible-embedder step):
wraps the ops.* entry points glayers.py calls with HIP event pairs keyed by the tensor shapes and integer arguments.
usage: python tools/literal_shapes.py [inn] [bs=4] [dtype=bf16] [steps=3]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops
from video_watermarking_forgery_detection_amd.models.IRNrhi_literal import IRNrhiLiteralModel
INN = len(sys.argv) > 1 and sys.argv[1] == "inn"
if INN:
    del sys.argv[1]
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NAMES = ["haar", "chan_copy_", "coupling_fwd", "coupling_bwd", "gconv_fwd", "gconv_wgrad", "gcolsum", "conv3x3_fwd", "conv3x3_wgrad", "unary_fwd", "unary_bwd", "qfatt_fwd", "qfatt_bwd", "spectral_norm_fwd",
         "spectral_norm_bwd", "gconv_pack", "pack_w3x3", "add_scaled", "gpool_fwd", "gpool_bwd", "pad_nchw_to_nhwc", "pad_nchw_to_nhwc_bwd", "gunpack_nchw",
         "gunpack_nchw_bwd", "adam_step", "clip_grad_norm_", "conv3x3_fwd_elu", "conv3x3_dgrad_elufused", "conv3x3_wgrad_bias", "unary_bwd_colsum", "chan_place", "chan_copy"]
events = []
recording = [False]
def wrap(name, fn):
    def w(*a, **k):
        if not recording[0]:
            return fn(*a, **k)
        key = (name,) + tuple(tuple(x.shape) if torch.is_tensor(x) else x for x in a if torch.is_tensor(x) or isinstance(x, (int, bool, tuple))) + \
              tuple((kk, tuple(v.shape) if torch.is_tensor(v) else v) for kk, v in sorted(k.items()) if torch.is_tensor(v) or isinstance(v, (int, bool, tuple)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        events.append((key, e0, e1))
        return r
    return w
for n in NAMES:
    if hasattr(ops, n):
        setattr(ops, n, wrap(n, getattr(ops, n)))
torch.manual_seed(3)
if INN:
    from video_watermarking_forgery_detection_amd import glayers as G
    from video_watermarking_forgery_detection_amd.models.invertible_net import Inveritible_Decolorization_PAMI, ResBlock
    net = Inveritible_Decolorization_PAMI(dims_in=[[4, 256, 256]], subnet_constructor=ResBlock,
                                          dtype={"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dt]).cuda()
    opt = G.FlatAdamW(net, lr=1e-5)
    x = torch.rand(bs, 4, 256, 256, device="cuda")
    for i in range(N + 2):
        recording[0] = i >= 2
        y = net(x)
        back, mid = net(y, rev=True)
        loss = ((y - x) ** 2).mean() + ((back - x) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
else:
    model = IRNrhiLiteralModel({"gpu_ids": [0], "is_train": True, "dist": False,
                                "train": {"lr_D": 1e-4, "beta1": 0.9, "beta2": 0.999, "weight_decay_G": 0.0, "gradient_clipping": 1.0, "compute_dtype": dt}})
    with torch.no_grad():
        model.localizer.BayarConv2D.weight.uniform_(0.5, 1.5)
    base = torch.rand(bs, 3, 256, 256)
    imgs = [torch.clamp(base + 0.02 * q * torch.randn(bs, 3, 256, 256), 0, 1) for q in range(6)]
    for i in range(N + 2):
        recording[0] = i >= 2
        model.feed_data((imgs, None))
        model.optimize_parameters(i)
torch.cuda.synchronize()
tab = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in events:
    t = tab[key]; t[0] += 1; t[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in tab.values()) / N
print(f"ops total {tot:.2f} ms/step ({len(events) // N} calls/step)")
byname = collections.defaultdict(float)
for k, v in tab.items(): byname[k[0]] += v[1] / N
for n, v in sorted(byname.items(), key=lambda x: -x[1]): print(f"  {n:24s} {v:8.3f} ms/step")
print("top shapes:")
for k, v in sorted(tab.items(), key=lambda x: -x[1][1])[:45]:
    print(f"  {v[1] / N:7.3f} ms/step  {v[0] // N:4d} x {v[1] / v[0] * 1000:7.1f} us  {k}")
