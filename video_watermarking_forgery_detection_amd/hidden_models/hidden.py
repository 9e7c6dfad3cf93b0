"""Hidden -- the GAN training step of the reference's hidden_models/hidden.py:12-118 on the MI355X
kernels: same constructor, `train_on_batch([images, messages])` / `validate_on_batch(...)` returning
(losses dict with the reference's keys, (encoded, noised, decoded)), same order of operations:

    D(cover) -> BCE(1) -> bwd | enc -> noise -> dec | D(enc.detach()) -> BCE(0) -> bwd | Adam(D)
    | D(enc) -> adv*BCE(1) + enc*MSE(enc, img) + dec*MSE(dec, msg) -> bwd | Adam(enc+dec)

The backward is written out (engine.py) instead of traced by autograd, every heavy op is a HIP kernel
behind the C ABI, parameters/gradients/moments of each optimiser live in flat f32 buffers (one fused
Adam launch, one RCCL bucket for data parallel), and the seven logged scalars are fetched with a
single asynchronous copy that the host waits for only when a value is read (StepLosses).
"""
import collections.abc

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, ops
from ..options import HiDDenConfiguration
from .discriminator import Discriminator
from .encoder_decoder import EncoderDecoder


LOSS_KEYS = ('loss           ', 'encoder_mse    ', 'dec_mse        ', 'bitwise-error  ', 'adversarial_bce', 'discr_cover_bce',
             'discr_encod_bce')   # hidden.py:105-113


class StepLosses(collections.abc.Mapping):
    """The reference's losses dict (hidden.py:105-113: name -> float), fetched from the device on first access.

    The seven scalars of a step are copied to pinned host memory asynchronously; reading any value (items(), [], ...)
    waits for that copy.  A training loop that logs every step behaves exactly like the reference's; one that only looks
    at the losses now and then (bench.py, print_each) no longer drains the GPU queue once per step."""

    def __init__(self, dev_vals, extra=None):
        self._host = torch.empty(len(LOSS_KEYS), dtype=torch.float32, pin_memory=True)
        self._host.copy_(dev_vals, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._vals = None
        self._extra = extra

    def _resolve(self):
        if self._vals is None:
            self._event.synchronize()
            self._vals = dict(zip(LOSS_KEYS, self._host.tolist()))
            if self._extra:
                self._vals['_extra'] = self._extra
            self._host = None
        return self._vals

    def __getitem__(self, k):
        return self._resolve()[k]

    def __iter__(self):
        return iter(self._resolve())

    def __len__(self):
        return len(self._resolve())

    def pop(self, k, *default):
        if k == '_extra' and self._vals is None:   # host-side data: no need to wait for the device
            extra, self._extra = self._extra, None
            if extra:
                return extra
            if default:
                return default[0]
            raise KeyError(k)
        return self._resolve().pop(k, *default)

    def __repr__(self):
        return repr(self._resolve())


class _FlatAdam:
    """torch.optim.Adam defaults (hidden.py:24-25) over flat parameter buffers, one kernel per buffer.
    Exposes `param_groups` / `state_dict` enough for BaseModel-style lr handling and checkpoints."""

    def __init__(self, modules, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        self.modules = list(modules)
        self.param_groups = [{"lr": lr, "initial_lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self.decoupled = decoupled
        self.step_count = 0
        self._m = None
        self._v = None
        self.capturing = False  # True while a step is being CAPTURED into a hipGraph (_StepGraph): step() leaves step_count to the graph's owner,
        self.hyper_dev = None   # and (without a scaler) launches the kernel that reads its step-count-dependent constants from this [2] f32
                                # device tensor, which the owner refreshes before each replay

    def _ensure(self):
        if self._m is None:
            self._m = [torch.zeros_like(m.flat_params) for m in self.modules]
            self._v = [torch.zeros_like(m.flat_params) for m in self.modules]

    def zero_grad(self):
        for m in self.modules:
            m.flat_grads.zero_()

    def attach_amp(self, amp):
        """run under a device-side GradScaler (ops.AmpState): the gradients arrive multiplied by its scale, the step divides it
        out, is skipped when they hold an inf / nan, and takes its bias corrections from the scaler's per-optimiser step count"""
        self.amp, self.amp_slot = amp, amp.slot()

    def step(self, grad_scale=1.0):
        self._ensure()
        if not self.capturing:
            self.step_count += 1
        g = self.param_groups[0]
        amp = getattr(self, "amp", None)
        if amp is not None:   # scaler.step(optimizer), IRNcrop_model.py:413-414
            amp.found_inf(self.amp_slot, [ops.sumsq(mod.flat_grads) for mod in self.modules])
            for mod, m, v in zip(self.modules, self._m, self._v):
                ops.adam_step_amp(mod.flat_params, mod.flat_grads, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                                  amp, self.amp_slot, decoupled=self.decoupled, grad_scale=grad_scale)
            return
        if self.hyper_dev is not None:   # being captured: the constants of step t come from device memory at replay time
            for mod, m, v in zip(self.modules, self._m, self._v):
                ops.adam_step_dev(mod.flat_params, mod.flat_grads, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                  g["weight_decay"], self.hyper_dev, decoupled=self.decoupled, grad_scale=grad_scale)
            return
        for mod, m, v in zip(self.modules, self._m, self._v):
            ops.adam_step(mod.flat_params, mod.flat_grads, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                          g["weight_decay"], self.step_count, decoupled=self.decoupled, grad_scale=grad_scale)

    def hyper(self, step):
        """the pair adam_step derives from the step count (host arithmetic): what a captured step's replay needs in hyper_dev"""
        g = self.param_groups[0]
        return ops.adam_hyper(g["lr"], g["betas"][0], g["betas"][1], step)

    def _params(self):
        return [p for mod in self.modules for p in mod.parameters()]

    def state_dict(self):
        """torch.optim.Adam's layout (what the reference's `{iter}.state` files hold, base_model.py:129-150): per-parameter
        {step, exp_avg, exp_avg_sq} sliced out of the flat moment buffers in parameters() order, one param group."""
        self._ensure()
        state, i = {}, 0
        steps = self.amp.step_count(self.amp_slot) if getattr(self, "amp", None) is not None else self.step_count
        for mod, m, v in zip(self.modules, self._m, self._v):
            off = 0
            for p in mod.parameters():
                n = p.numel()
                state[i] = {"step": torch.tensor(float(steps)), "exp_avg": m[off:off + n].view_as(p).detach().cpu().clone(),
                            "exp_avg_sq": v[off:off + n].view_as(p).detach().cpu().clone()}
                off += n
                i += 1
        g = dict(self.param_groups[0])
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": bool(self.decoupled), "initial_lr": g.get("initial_lr", g["lr"]), "params": list(range(i))}
        return {"state": state if steps else {}, "param_groups": [group]}

    def load_state_dict(self, sd):
        """accepts torch.optim.Adam / AdamW state_dicts (the reference's files) and this class's round-1 flat layout"""
        self._ensure()
        if "state" not in sd:   # round-1 layout: {step, param_groups, exp_avg[list], exp_avg_sq[list]}
            self._set_steps(int(sd["step"]))
            self._update_group(sd["param_groups"][0])
            for t, src in zip(self._m, sd["exp_avg"]):
                t.copy_(src)
            for t, src in zip(self._v, sd["exp_avg_sq"]):
                t.copy_(src)
            return
        groups = sd["param_groups"]
        if len(groups) != 1:
            raise ValueError("expected one param group (torch.optim.Adam over one parameter list)")
        params = self._params()
        ids = list(groups[0].get("params", range(len(params))))
        if len(ids) != len(params):
            raise ValueError(f"optimizer state has {len(ids)} parameters, the networks have {len(params)}")
        self._update_group(groups[0])
        state = sd["state"]
        steps = set()
        i = 0
        for mod, m, v in zip(self.modules, self._m, self._v):
            off = 0
            for p in mod.parameters():
                n = p.numel()
                st = state.get(ids[i], state.get(str(ids[i])))
                if st is None:   # torch keeps no entry for a parameter that never received a gradient
                    m[off:off + n].zero_(); v[off:off + n].zero_()
                else:
                    if tuple(st["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state of parameter {i} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
                    m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(float(st["step"])))
                off += n
                i += 1
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused flat-buffer step keeps one count")
        self._set_steps(steps.pop() if steps else 0)

    def _set_steps(self, n):
        """the loaded step count: on the host, and -- under a scaler -- in the device slot adam_amp_kernel takes its bias correction's t from
        (a resumed f16 run otherwise stepped with t = 1 on warm moments: bc1 = 0.1, bc2 = 0.001 instead of ~1)"""
        self.step_count = n
        if getattr(self, "amp", None) is not None:
            self.amp.set_step_count(self.amp_slot, n)

    def _update_group(self, g):
        for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]
        if "decoupled_weight_decay" in g:
            self.decoupled = bool(g["decoupled_weight_decay"])


def _noise_fwd(noiser, enc, cover):
    """explicit fwd/bwd if the noise layer offers it, autograd otherwise (any nn.Module works)."""
    if hasattr(noiser, "fwd") and hasattr(noiser, "bwd"):
        y, c = noiser.fwd(enc)
        return y, ("explicit", c)
    x = enc.detach().requires_grad_(True)
    with torch.enable_grad():
        out = noiser([x, cover])
        y = out[0] if isinstance(out, (list, tuple)) else out
    return y.detach(), ("autograd", (x, y))


def _noise_bwd(noiser, ctx, g):
    kind, c = ctx
    if kind == "explicit":
        return noiser.bwd(c, g)
    x, y = c
    if not y.requires_grad:
        return torch.zeros_like(g)
    (gx,) = torch.autograd.grad(y, x, g, allow_unused=True)
    return gx if gx is not None else torch.zeros_like(g)


def _noise_bwd_is_zero(noiser, ctx):
    """does the attack layer declare the gradient of THIS forward call identically zero (noise_layers.Jpeg: torch.round)?"""
    kind, c = ctx
    f = getattr(noiser, "bwd_is_zero", None)
    return bool(kind == "explicit" and f is not None and f(c))


def _accepts_id(fn):
    """does fn(image, id=...) exist?  Looked up from the signature (cached per function), never by catching TypeError around
    the call -- that would hide a TypeError raised inside the attack and run it twice."""
    key = getattr(fn, "__func__", fn)
    hit = _ACCEPTS_ID.get(key)
    if hit is None:
        import inspect
        try:
            ps = inspect.signature(fn).parameters
            hit = "id" in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values())
        except (TypeError, ValueError):
            hit = False
        _ACCEPTS_ID[key] = hit
    return hit


_ACCEPTS_ID = {}


class _StepGraph:
    """One training step of `Hidden` as a hipGraph (the reference runs one Python step per batch per rank, train.py:99-109; here the host's
    ~190 launches per step are enqueued once and replayed).  Exact by construction: the graph holds the very launches of the eager step --
    the step has no autograd and no host synchronisation (engine.py) --, its inputs are static tensors refreshed by copy_, and the only
    host-side number a step depends on, Adam's step count, reaches the kernels through device memory (wm_adam_step_dev) that is refreshed
    before each replay.  Call k of a shape: k < WARMUP eager (real steps; they build the weight-pack plans and settle the allocator),
    k == WARMUP capture + replay, later calls replay.  The tensors a replay returns (encoded, noised, decoded) are the graph's own and are
    overwritten by the next replay of the same graph, as torch.cuda.CUDAGraph documents; the losses are copied out per step."""
    WARMUP = 2

    def __init__(self, hidden):
        import weakref
        self._h = weakref.ref(hidden)   # (no Hidden <-> _StepGraph cycle: a dropped model must die by reference count, taking its hipGraph
                                        # with it THEN -- not whenever the cyclic collector next runs, which may be inside another capture)
        self.calls = 0
        self.graph = None
        self.failed = None   # why the capture failed, if it did: the step then runs eagerly (same launches, same results) and says so once
        self._seen = {}

    @property
    def h(self):
        return self._h()

    def _hyper_refresh(self):
        opts = (self.h.optimizer_discrim, self.h.optimizer_enc_dec)
        if self.h.amp is not None:
            return   # under the scaler the step counts live on the device already (wm_adam_step_amp)
        vals = [v for o in opts for v in o.hyper(o.step_count + 1)]
        # a FRESH pinned tensor per replay: the copy below is asynchronous, and a host that runs ahead of the GPU (a loop that does not read
        # the losses every step: bench.py) would overwrite a reused staging buffer with the NEXT step's constants before this step's copy
        # has run -- found in round 4 by tools/train_sanity_modes.py (the per-step tests synchronise and never saw it).  The caching host
        # allocator hands a pinned block out again only after the copies that read it have completed.
        host = torch.tensor(vals, dtype=torch.float32).pin_memory()
        self.hyper.copy_(host, non_blocking=True)

    def step(self, images, messages):
        h = self.h
        if self.calls < self.WARMUP or self.failed is not None:
            self.calls += 1
            return h._step_eager(images, messages)
        if self.graph is None:
            dev = images.device
            self.img, self.msg = torch.empty_like(images), torch.empty_like(messages)
            self.hyper = torch.zeros(4, device=dev, dtype=torch.float32)
            self.img.copy_(images); self.msg.copy_(messages)
            self._seen = {"img": (images, images._version), "msg": (messages, messages._version)}
            opts = (h.optimizer_discrim, h.optimizer_enc_dec)
            for i, o in enumerate(opts):
                o._ensure()
                o.capturing = True
                if h.amp is None:
                    o.hyper_dev = self.hyper[2 * i:2 * i + 2]
            graph = torch.cuda.CUDAGraph()
            # Python's cyclic collector must not run inside the capture: whatever dead cycle it finds may own device memory, events or
            # another hipGraph, and releasing those calls HIP functions that are illegal while a stream captures (the process aborts --
            # seen in round 4 with the previous test's model in a cycle).  torch 2.10's graph.__enter__ no longer collects first.
            import gc
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph):
                    self.result = h._step_launches(self.img, self.msg)
            except Exception as e:   # noqa: BLE001 -- whatever stopped the capture (nothing was executed: the model is as it was before)
                import warnings
                self.failed = f"{type(e).__name__}: {e}"
                self.result = None
                warnings.warn("hipGraph capture of the training step failed; this and the following steps are enqueued eagerly (same launches, "
                              "same results, more host time per step): " + self.failed)
            finally:
                if gc_was_on:
                    gc.enable()
                for o in opts:
                    o.hyper_dev, o.capturing = None, False
            if self.failed is not None:
                self.calls += 1
                return h._step_eager(images, messages)
            self.graph = graph
        # fresh inputs into the graph's static tensors -- unless the caller hands in the very tensor object of the previous call, unmodified
        # (torch's version counter: every in-place op bumps it, this library's own through ops._wrote): a loop over one resident batch
        for name, src, dst in (("img", images, self.img), ("msg", messages, self.msg)):
            seen = self._seen.get(name)
            if seen is None or seen[0] is not src or seen[1] != src._version:
                dst.copy_(src)
                self._seen[name] = (src, src._version)
        self._hyper_refresh()
        self.graph.replay()
        self.calls += 1
        h.optimizer_discrim.step_count += 1
        h.optimizer_enc_dec.step_count += 1
        vals, extra_logs, outs = self.result
        return h._wrap_losses(vals, extra_logs), outs


class Hidden:
    def __init__(self, configuration: HiDDenConfiguration, device: torch.device, noiser, tb_logger=None,
                 compute_dtype=torch.bfloat16, grad_sync=None, amp=None, keep_dead_discriminator_grads=True):
        """
        :param configuration: sizes / loss weights (options.HiDDenConfiguration)
        :param device: must be a cuda (ROCm) device -- the step has no CPU path
        :param noiser: attack layer(s): an object with forward([encoded, cover]) -> [noised, cover]
                       (noise_layers.Noiser) or any module of noise_layers
        :param tb_logger: accepted for signature compatibility; unused
        :param compute_dtype: torch.bfloat16 (production) or torch.float32 (parity path)
        :param grad_sync: None or a distributed.GradSync (its interface is required: start / finish / finish_all / scale / average_) --
                          the data-parallel bucket all-reduce started from inside the backward.  A plain callable(flat_grad_tensor) is
                          wrapped into that interface (run once per bucket, synchronously, where the bucket would be waited for)
        :param keep_dead_discriminator_grads: True (default) = the reference's state after a step: g_loss.backward() (hidden.py:101) also
                    accumulates the generator loss's gradients into the DISCRIMINATOR's parameters; nothing ever reads them (no optimiser
                    step uses them, hidden.py:67 zeroes them first thing in the next step).  False: that third pass through the
                    discriminator computes the gradient wrt `encoded` only (no conv weight gradients: 2 body-layer + 1 first-layer
                    weight-gradient GEMMs fewer per step, 9.9 of the step's 249.0 GFLOP per 256x256 frame).  Losses, outputs and every
                    parameter update are identical either way; only the discriminator's .grad left behind after the step differs.
        :param amp: optional ops.AmpState -- torch.cuda.amp.GradScaler semantics on the device (models/IRNcrop_model.py:143,
                    407-416): every loss gradient is multiplied by its scale, both optimisers step through it.  Required in
                    practice with compute_dtype=torch.float16 (the gradients of a 3M-element mean loss underflow f16 otherwise)
        """
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("Hidden runs on the MI355X HIP path only (device must be cuda)")
        self.encoder_decoder = EncoderDecoder(configuration, noiser).to(device)
        self.discriminator = Discriminator(configuration).to(device)
        engine.set_compute_dtype(self.encoder_decoder, compute_dtype)
        engine.set_compute_dtype(self.discriminator, compute_dtype)
        enc, dec = self.encoder_decoder.encoder, self.encoder_decoder.decoder
        for m in (enc, dec, self.discriminator):
            m.flatten_parameters_()
        self.optimizer_enc_dec = _FlatAdam([enc, dec])
        self.optimizer_discrim = _FlatAdam([self.discriminator])
        if configuration.use_vgg:
            raise NotImplementedError("use_vgg: the reference's vgg_loss module is absent from its tree")
        self.vgg_loss = None
        self.config = configuration
        self.device = device
        self.cover_label = 1
        self.encoded_label = 0
        self.tb_logger = tb_logger
        from ..distributed import as_grad_sync
        self.grad_sync = as_grad_sync(grad_sync)
        self.keep_dead_discriminator_grads = bool(keep_dead_discriminator_grads)
        self.amp = amp
        self.amp_owner = True   # this object calls amp.update() at the end of a step (a wrapping model may take that over)
        if amp is not None:
            self.optimizer_enc_dec.attach_amp(amp)
            self.optimizer_discrim.attach_amp(amp)
        self.noise_id = None  # optional deterministic choice for Combined/Noiser layers
        self.lazy_losses = True  # train_on_batch returns StepLosses (host sync on first read) instead of a plain dict
        self._graphs = None      # enable_graph(): (shapes, attack choice, ...) -> _StepGraph
        # An attack whose backward is identically zero (Jpeg: torch.round) makes the decoder's gradient wrt its input a value nothing depends
        # on: with this switch on (default) the step does not compute it (the decoder's first layer runs its weight gradient alone, the
        # attack's backward and the addition of its zeros are not launched).  Every loss and output is bit-identical either way, and so is
        # every parameter, .grad and optimiser state except ONE tensor's rounding: the decoder's first-layer weight gradient is then summed
        # by the weight-gradient-only kernel instead of the one-pass kernel -- the same sum in another order (tests/test_gpu_graph.py).
        # False = launch it all, as the reference's autograd does
        self.skip_zero_attack_gradient = True
        self.two_streams = False  # the step's two independent chains on two streams (_train_step_two_chains); same results bit for bit
        self._streams = None

    MAX_GRAPHS = 32   # captured variants kept per model (shapes x attack variants x ...); calls beyond that run eagerly

    def enable_graph(self, on=True):
        """replay the step from a hipGraph instead of enqueueing its ~190 launches every call (same results bit for bit: _StepGraph).
        Used for plain train_on_batch(batch) calls on one GPU with an attack layer that has an explicit fwd / bwd and declares itself
        `capturable` (the same launches with the same arguments every call: the Jpeg family, GaussianBlur, MiddleBlur, Identity -- not the
        layers that draw their arguments on the host, Resize / Crop / Combined) or names the deterministic variant this call runs through
        `capture_key()` (one graph per variant: the model surface's attack cycle, which calls Resize / Crop with fixed arguments); a call with
        extra_encoded_grad / clip / enc_gate, with a grad_sync, or while a kernel timer is installed runs eagerly as before."""
        self._graphs = {} if on else None
        return self

    # ------------------------------------------------------------------ helpers
    def _bce_logits(self, logits, target, gscale=1.0):
        """nn.BCEWithLogitsLoss (mean) value ([1] tensor) and gscale * gradient wrt logits, on a [B,1] tensor: one launch."""
        loss, grad = ops.bce_logits(logits, target, gscale, gscale_dev=self._gsd())
        return loss[0], grad.view_as(logits)

    def _gsd(self):
        """the AMP loss scale as a device scalar (None without a scaler): scaler.scale(loss).backward()"""
        return self.amp.scale if self.amp is not None else None

    def _run_noiser(self, enc, cover):
        n = self.encoder_decoder.noiser
        if hasattr(n, "fwd") and hasattr(n, "bwd"):
            if _accepts_id(n.fwd):   # Combined / Noiser / the attack cycle take a deterministic choice; single layers do not
                y, c = n.fwd(enc, id=self.noise_id)
            else:
                y, c = n.fwd(enc)
            return y, ("explicit", c)
        return _noise_fwd(n, enc, cover)

    # ------------------------------------------------------------------ the step
    def train_on_batch(self, batch: list, extra_encoded_grad=None, clip=None, enc_gate=None):
        """extra_encoded_grad: optional callable(encoded, images, g_enc) -> [(name, value)] that ADDS its gradient wrt the
        encoded image into g_enc (the tamper-localisation branch of models/IRNrhi_model.py), run after the decoder's backward;
        clip: optional callable(list of flat gradient buffers) applied before each optimiser step -- the buffers of one call
        are clipped TOGETHER by their joint norm (clip_grad_norm_ over netG.parameters(), IRNcrop_model.py:410-412);
        enc_gate: optional callable(encoded, images) -> [2] device tensor (psnr, weight): the PSNR-gated weight of the
        image-fidelity term (IRNcrop_model.py:379-388), multiplied into encoder_loss without a host sync."""
        images, messages = batch
        images = images.to(self.device, torch.float32).contiguous()
        messages = messages.to(self.device, torch.float32).contiguous()
        n = self.encoder_decoder.noiser
        if (self._graphs is not None and extra_encoded_grad is None and clip is None and enc_gate is None and self.grad_sync is None
                and not ops.kernel_timer_installed() and hasattr(n, "fwd") and hasattr(n, "bwd") and not torch.cuda.is_current_stream_capturing()):
            # a layer is capturable as a whole (`capturable`), or says per call which of its deterministic variants this call runs
            # (`capture_key()`: a hashable, or None for "not this time" -- the model surface's attack cycle, whose choice follows the step)
            ck = True if getattr(n, "capturable", False) else (n.capture_key() if hasattr(n, "capture_key") else None)
        else:
            ck = None
        if ck is not None:
            key = (tuple(images.shape), tuple(messages.shape), self.noise_id, self.keep_dead_discriminator_grads, self.lazy_losses, self.two_streams, self.skip_zero_attack_gradient,
                   self.encoder_decoder.encoder.compute_dtype, ck)
            if self.amp is not None:   # under the scaler the learning rate reaches wm_adam_step_amp as a launch argument: a scheduler's new value needs its own graph
                key += tuple(o.param_groups[0]["lr"] for o in (self.optimizer_discrim, self.optimizer_enc_dec))
            g = self._graphs.get(key)
            if g is None and len(self._graphs) < self.MAX_GRAPHS:
                g = self._graphs[key] = _StepGraph(self)
            if g is not None:
                return g.step(images, messages)
        return self._step_eager(images, messages, extra_encoded_grad, clip, enc_gate)

    def _step_eager(self, images, messages, extra_encoded_grad=None, clip=None, enc_gate=None):
        vals, extra_logs, outs = self._step_launches(images, messages, extra_encoded_grad, clip, enc_gate)
        return self._wrap_losses(vals, extra_logs), outs

    def _wrap_losses(self, vals, extra_logs):
        if self.lazy_losses:
            return StepLosses(vals, extra_logs)
        losses = dict(zip(LOSS_KEYS, vals.tolist()))
        if extra_logs:
            losses['_extra'] = extra_logs
        return losses

    def _step_launches(self, images, messages, extra_encoded_grad=None, clip=None, enc_gate=None):
        """every launch of one step, no host synchronisation -> ([7] device tensor of the logged scalars, extra logs, (encoded, noised, decoded))"""
        B = images.shape[0]
        cfg = self.config
        ed = self.encoder_decoder
        enc_net, dec_net, D = ed.encoder, ed.decoder, self.discriminator
        ed.train()
        D.train()
        gD = engine.grad_dict(D)
        gE = engine.grad_dict(enc_net)
        gDec = engine.grad_dict(dec_net)
        # packed conv weights: one launch per network here (+ one for D after its optimiser step) instead of one per
        # conv call; valid only inside this step (the parameters are ours to change until it returns)
        nets = (enc_net, dec_net, D)
        try:
            for n in nets:
                n.refresh_packs()
            with engine.defer_bn_counters(), engine.share_image_acts(dec_net.compute_dtype):
                return self._train_step(images, messages, B, cfg, ed, enc_net, dec_net, D, gD, gE, gDec, extra_encoded_grad, clip, enc_gate)
        finally:
            for n in nets:
                n.invalidate_packs()

    def _train_step(self, images, messages, B, cfg, ed, enc_net, dec_net, D, gD, gE, gDec, extra_encoded_grad, clip, enc_gate=None):
        if (self.two_streams and self.grad_sync is None and extra_encoded_grad is None and clip is None and enc_gate is None
                and hasattr(ed.noiser, "fwd") and hasattr(ed.noiser, "bwd")):
            return self._train_step_two_chains(images, messages, B, cfg, ed, enc_net, dec_net, D, gD, gE, gDec)
        # ---------------- train the discriminator (hidden.py:68-83)
        # (fwd_loss: the forward, the loss on its output and the head's share of the backward -- one launch behind the pool instead of four)
        d_on_cover, d_loss_on_cover, c = D.fwd_loss(images, self.cover_label, 1.0, gD, accumulate=False, gscale_dev=self._gsd())
        D.bwd(c, None, gD, accumulate=False, need_input_grad=False)       # zero_grad + backward

        encoded, cE = enc_net.fwd(images, messages)
        noised, cN = self._run_noiser(encoded, images)

        d_on_encoded, d_loss_on_encoded, c = D.fwd_loss(encoded, self.encoded_label, 1.0, gD, accumulate=True, gscale_dev=self._gsd())   # encoded.detach()
        D.bwd(c, None, gD, accumulate=True, need_input_grad=False)
        gs = self.grad_sync
        # data parallel: the discriminator's bucket travels under the decoder's forward (independent of D; the reference runs it
        # before D(encoded), the results are the same)
        pending_d = gs.start(D.flat_grads) if gs is not None else None
        decoded, msg_out, cDec = dec_net.fwd_loss(noised, messages, 2.0 * cfg.decoder_loss / (B * cfg.message_length), gDec, accumulate=False,
                                                  gscale_dev=self._gsd())     # mse, bit error, and the head's backward
        gscale = 1.0
        if gs is not None:
            gs.finish(pending_d)
            gscale = gs.scale
            if clip is not None:
                gs.average_(D.flat_grads)
                gscale = 1.0
        if clip is not None:
            clip([D.flat_grads])
        self.optimizer_discrim.step(grad_scale=gscale)
        D.refresh_packs()

        # ---------------- train the generator (hidden.py:85-103)
        d_on_encoded_for_enc, g_loss_adv, c = D.fwd_loss(encoded, self.cover_label, cfg.adversarial_loss, gD, accumulate=True, gscale_dev=self._gsd())
        g = None
        # the reference's g_loss.backward() also accumulates into the discriminator's .grad (zeroed at the start of the next step, never
        # read): kept by default so the .grad state matches; keep_dead_discriminator_grads=False computes the image gradient alone
        n_img = encoded.numel()
        gate = enc_gate(encoded, images) if enc_gate is not None else None
        if gate is None:
            # the adversarial term's gradient (NHWC, as the discriminator's first layer wrote it) + the image-fidelity term's, and that
            # term's loss, in one pass
            g_img = D.bwd(c, g, gD, accumulate=True, need_input_grad=True, weight_grads=self.keep_dead_discriminator_grads, raw_input_grad=True)
            g_enc, enc_part = ops.image_grad_mse(g_img, encoded, images, 2.0 * cfg.encoder_loss / n_img, gscale_dev=self._gsd())
        else:
            g_enc = D.bwd(c, g, gD, accumulate=True, need_input_grad=True, weight_grads=self.keep_dead_discriminator_grads)
            enc_part, g_mse = ops.mse_fwd_bwd_gated(encoded, images, 2.0 * cfg.encoder_loss / n_img, gate[1:2], gscale_dev=self._gsd())
            ops.axpy_(g_enc, g_mse)
        zero_attack = self.skip_zero_attack_gradient and _noise_bwd_is_zero(ed.noiser, cN)
        g_noised = dec_net.bwd(cDec, None, gDec, accumulate=False, need_input_grad=not zero_attack)
        # data parallel: the decoder's bucket goes out now and travels while the attack and the encoder run their backward
        pending = [gs.start(dec_net.flat_grads)] if gs is not None else []
        if not zero_attack:
            g_from_noise = _noise_bwd(ed.noiser, cN, g_noised)
            ops.axpy_(g_enc, g_from_noise.contiguous())
        extra_logs = []
        if extra_encoded_grad is not None:
            extra_logs = extra_encoded_grad(encoded, images, g_enc)
        if gate is not None:
            extra_logs = [('PF', gate[0:1])] + list(extra_logs)
        if gs is not None:
            # the encoder in two reverse-order buckets: [after_concat, final] leaves under the body layers' backward
            cut = enc_net.body_param_count()
            enc_net.bwd(cE, g_enc, gE, accumulate=False, after_head=lambda: pending.append(gs.start(enc_net.flat_grads[cut:])))
            pending.append(gs.start(enc_net.flat_grads[:cut]))
            gs.finish_all(pending)
        else:
            enc_net.bwd(cE, g_enc, gE, accumulate=False)
        gscale = 1.0
        if gs is not None:
            gscale = gs.scale
            if clip is not None:
                gs.average_(enc_net.flat_grads)
                gs.average_(dec_net.flat_grads)
                gscale = 1.0
        if clip is not None:
            clip([enc_net.flat_grads, dec_net.flat_grads])   # one norm over encoder_decoder.parameters()
        self.optimizer_enc_dec.step(grad_scale=gscale)
        if self.amp is not None and self.amp_owner:
            self.amp.update()   # scaler.update(), IRNcrop_model.py:416

        # ---------------- metrics: one host sync for all seven scalars (hidden.py:105-117)
        vals = ops.hidden_metrics(enc_part, n_img, msg_out, g_loss_adv, d_loss_on_cover, d_loss_on_encoded, cfg.adversarial_loss,
                                  cfg.encoder_loss, cfg.decoder_loss)
        return vals, extra_logs, (encoded, noised, decoded)

    def _train_step_two_chains(self, images, messages, B, cfg, ed, enc_net, dec_net, D, gD, gE, gDec):
        """The same step, same launches, same arithmetic -- enqueued as the TWO independent chains hidden.py:54-118 consists of until the encoder's
        backward: the discriminator's passes (stream A) beside encoder -> attack -> decoder forward and the decoder's backward (stream B).

            A: D(cover) fwd/bwd . . . . . . | wait encoded | D(enc.detach()) fwd/bwd, Adam(D), D(enc) fwd, dgrad to the image, + MSE gradient
            B: encoder fwd | attack fwd, decoder fwd, message loss, decoder bwd, attack bwd
            join: g_enc += attack gradient; encoder bwd; Adam(enc + dec); the seven scalars
        (an attack that passes back zeros -- Jpeg's torch.round, skip_zero_attack_gradient -- leaves chain B without a successor in the
        encoder: the encoder's backward then runs on chain A right behind the discriminator's, beside the decoder's backward)

        Why: every persistent kernel spends a fixed part of its launch outside its tile loop (filter -> LDS, first HBM round trip, for the
        one-pass backward the weight-gradient slabs) and the chip idles through each launch's ramp and tail; a second, independent launch fills
        those.  Measured on the step itself: 5.25 ms against 5.35 ms enqueued, 5.21 against 5.34 replayed from the graph (-1.7 ... -2.4 %,
        profiles/r04_step_modes.txt).  The kernels alone promised more (24 launches of the forward 64 -> 64 kernel 1.75 ms from two streams
        against 2.02 from one, of the one-pass backward 3.93 against 4.17: tools/bench_two_chains.py, profiles/r04_two_chains.txt), but that
        comparison's one-stream arm carries host gaps the step does not have.  Results are bit-identical to the one-stream order: no kernel's
        inputs or reduction order change, only when it runs."""
        main = torch.cuda.current_stream()
        sA, sB = self._chain_streams()
        dts = {D.compute_dtype, enc_net.compute_dtype}
        for dt in dts:
            engine.image_to_act(images, dt)              # converted once, before the fork: both chains read it
        sA.wait_stream(main); sB.wait_stream(main)
        with torch.cuda.stream(sA):
            d_on_cover, d_loss_on_cover, c = D.fwd_loss(images, self.cover_label, 1.0, gD, accumulate=False, gscale_dev=self._gsd())
            D.bwd(c, None, gD, accumulate=False, need_input_grad=False)
        with torch.cuda.stream(sB):
            encoded, cE = enc_net.fwd(images, messages)
            for dt in {D.compute_dtype, dec_net.compute_dtype}:
                engine.image_to_act(encoded, dt)         # (on the producing stream: chain A and -- under an Identity attack -- the decoder read it)
            ev_enc = torch.cuda.Event()
            ev_enc.record(sB)
            noised, cN = self._run_noiser(encoded, images)
            decoded, msg_out, cDec = dec_net.fwd_loss(noised, messages, 2.0 * cfg.decoder_loss / (B * cfg.message_length), gDec, accumulate=False,
                                                      gscale_dev=self._gsd())
            zero_attack = self.skip_zero_attack_gradient and _noise_bwd_is_zero(ed.noiser, cN)
            g_noised = dec_net.bwd(cDec, None, gDec, accumulate=False, need_input_grad=not zero_attack)
            g_from_noise = None if zero_attack else _noise_bwd(ed.noiser, cN, g_noised).contiguous()
        with torch.cuda.stream(sA):
            sA.wait_event(ev_enc)
            d_on_encoded, d_loss_on_encoded, c = D.fwd_loss(encoded, self.encoded_label, 1.0, gD, accumulate=True, gscale_dev=self._gsd())
            D.bwd(c, None, gD, accumulate=True, need_input_grad=False)
            self.optimizer_discrim.step(grad_scale=1.0)
            D.refresh_packs()
            d_on_encoded_for_enc, g_loss_adv, c = D.fwd_loss(encoded, self.cover_label, cfg.adversarial_loss, gD, accumulate=True, gscale_dev=self._gsd())
            g = None
            g_img = D.bwd(c, g, gD, accumulate=True, need_input_grad=True, weight_grads=self.keep_dead_discriminator_grads, raw_input_grad=True)
            n_img = encoded.numel()
            g_enc, enc_part = ops.image_grad_mse(g_img, encoded, images, 2.0 * cfg.encoder_loss / n_img, gscale_dev=self._gsd())
            if zero_attack:
                # nothing of chain B reaches the encoder's gradient (the attack passes back zeros): the encoder's backward continues chain A,
                # beside the decoder's backward on chain B, and the chains meet only at the optimiser step
                enc_net.bwd(cE, g_enc, gE, accumulate=False)
        main.wait_stream(sA); main.wait_stream(sB)
        if not zero_attack:
            ops.axpy_(g_enc, g_from_noise)
            enc_net.bwd(cE, g_enc, gE, accumulate=False)
        self.optimizer_enc_dec.step(grad_scale=1.0)
        if self.amp is not None and self.amp_owner:
            self.amp.update()
        vals = ops.hidden_metrics(enc_part, n_img, msg_out, g_loss_adv, d_loss_on_cover, d_loss_on_encoded, cfg.adversarial_loss,
                                  cfg.encoder_loss, cfg.decoder_loss)
        # Memory: the caching allocator hands a freed block back to the pool of the stream that allocated it.  Every tensor one stream
        # allocates and ANOTHER reads (encoded and its NHWC form, g_enc, g_from_noise, the encoder's activations in cE, the loss scalars) is
        # referenced until this function returns, and a side stream is given work only between a fork (wait_stream(main)) and the join
        # above -- so a block that returns to a side stream's pool is not handed out again before main's readers of it are ordered ahead
        return vals, [], (encoded, noised, decoded)

    def _chain_streams(self):
        if self._streams is None:
            self._streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        return self._streams

    def validate_on_batch(self, batch: list):
        """hidden.py:120-182: eval-mode BatchNorm (running statistics), no parameter update."""
        images, messages = batch
        images = images.to(self.device, torch.float32).contiguous()
        messages = messages.to(self.device, torch.float32).contiguous()
        B = images.shape[0]
        cfg = self.config
        ed = self.encoder_decoder
        enc_net, dec_net, D = ed.encoder, ed.decoder, self.discriminator
        ed.eval()
        D.eval()
        with torch.no_grad():
            d_on_cover, _ = D.fwd(images, training=False)
            d_loss_on_cover, _ = self._bce_logits(d_on_cover, self.cover_label)
            encoded, _ = enc_net.fwd(images, messages, training=False)
            noised, _ = self._run_noiser(encoded, images)
            decoded, _ = dec_net.fwd(noised, training=False)
            d_on_encoded, _ = D.fwd(encoded, training=False)
            d_loss_on_encoded, _ = self._bce_logits(d_on_encoded, self.encoded_label)
            g_loss_adv, _ = self._bce_logits(d_on_encoded, self.cover_label)
            g_loss_enc = F.mse_loss(encoded, images)
            g_loss_dec = F.mse_loss(decoded, messages)
            g_loss = cfg.adversarial_loss * g_loss_adv + cfg.encoder_loss * g_loss_enc + cfg.decoder_loss * g_loss_dec
            rounded = decoded.round().clamp(0, 1)
            bit_err = (rounded - messages).abs().sum() / (B * messages.shape[1])
            vals = torch.stack([g_loss, g_loss_enc, g_loss_dec, bit_err, g_loss_adv, d_loss_on_cover,
                                d_loss_on_encoded]).tolist()
        losses = {
            'loss           ': vals[0], 'encoder_mse    ': vals[1], 'dec_mse        ': vals[2],
            'bitwise-error  ': vals[3], 'adversarial_bce': vals[4], 'discr_cover_bce': vals[5],
            'discr_encod_bce': vals[6],
        }
        return losses, (encoded, noised, decoded)

    def to_stirng(self):
        return '{}\n{}'.format(str(self.encoder_decoder), str(self.discriminator))
