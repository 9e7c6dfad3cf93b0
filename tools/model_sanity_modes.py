"""ON THE GPU BOX: configuration C3 through feed_data / optimize_parameters (pinned host batches, the 7-attack cycle) in three modes --
enqueued eagerly with the logs read every step | one captured step per attack layer, logs read every step | captured + train.deferred_logs
with the logs read one step late (the host runs ahead).  All three must end with the same parameters, bit for bit.
usage: python tools/model_sanity_modes.py [steps=80] [size=256]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
from video_watermarking_forgery_detection_amd.options.options import dict_to_nonedict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
attacks = ["Jpeg50", "JpegSS70", "JpegMask90", "GaussianBlur", "MiddleBlur3", "Resize", "Crop"]


def digest(m):
    h = hashlib.sha256()
    for net in (m.netG.encoder, m.netG.decoder, m.discriminator):
        for t in net.state_dict().values():
            h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def run(graph, deferred, two):
    torch.manual_seed(10)
    opt = dict_to_nonedict({"gpu_ids": [0], "dist": False, "is_train": True, "datasets": {"train": {"GT_size": S, "batch_size": 16}},
                            "train": {"compute_dtype": "bf16", "attacks": attacks, "lr_G": 1e-3, "manual_seed": 10, "save_interval": 10 ** 9, "localizer": False,
                                      "graph": graph, "deferred_logs": deferred, "two_streams": two},
                            "path": {"models": "/tmp/wm_models", "training_state": "/tmp/wm_state"}})
    m = IRNrhiModel(opt)
    g = torch.Generator().manual_seed(3)
    pending, losses = None, []
    for step in range(1, N + 1):
        x = torch.rand(16, 3, S, S, generator=g).pin_memory()
        m.feed_data(x)
        m.messages = torch.randint(0, 2, (16, 30), generator=g).float().cuda()
        logs, _ = m.optimize_parameters(step, None)
        if deferred:
            if pending is not None and len(pending):
                losses.append(dict(pending)["loss"])
            pending = logs
        elif logs:
            losses.append(dict(logs)["loss"])
    if deferred and pending is not None and len(pending):
        losses.append(dict(pending)["loss"])
    torch.cuda.synchronize()
    return digest(m), losses


ref = run(False, False, False)
print("eager, one stream, logs every step        ", ref[0], round(ref[1][-1], 6))
for name, args in (("eager, two chains", (False, False, True)), ("captured, two chains, logs every step", (True, False, True)),
                   ("captured, two chains, deferred logs", (True, True, True)), ("captured, one stream, deferred logs", (True, True, False))):
    got = run(*args)
    first = next((i for i, (a, b) in enumerate(zip(got[1], ref[1])) if a != b), None)
    print(f"{name:42s}", got[0], round(got[1][-1], 6), "IDENTICAL" if got[0] == ref[0] and first is None else f"DIFFERS (first differing logged loss: entry {first})")
