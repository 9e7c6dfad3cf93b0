"""IRNrhiModel -- the model class whose surface train.py drives (reference
models/IRNrhi_model.py:75,358-395,425-431,690-698,1138-1145; composition of the localisation branch
from models/IRNcrop_model.py:337-416):

    model = IRNrhiModel(opt)
    model.feed_data(batch)
    logs, debug_logs = model.optimize_parameters(step, latest_values)
    progbar.add(len(model.real_H), values=logs)

The step behind it is the HiDDeN-order watermark embed -> attack -> extract GAN step of
hidden_models/hidden.py:54-118 on the HIP kernels (hidden_models.Hidden), optionally followed by the
tamper-localisation branch: STE clamp -> Quantization -> splice with the previous batch by the mask ->
attack -> Quantization -> UNet -> BCEWithLogits on the (sigmoid) mask, whose gradient reaches both the
UNet and, through the attack, the encoder.

Bookkeeping kept from the reference: `global_step`, input clamp to [0,1] (:430), no work until two
previous batches exist (:446), checkpoint every `save_interval` steps at `step % save_interval == 10`
on rank <= 0 (:691-693), rotation of `previous_images` buffers (:694-697), `logs` as a list of
(name, float).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import engine, ops
from ..hidden_models import Hidden
from ..hidden_models.hidden import _FlatAdam
from ..network.UNet import UNet
from ..noise_layers import Combined, Crop, GaussianBlur, Identity, Jpeg, JpegMask, JpegSS, MiddleBlur, Resize
from ..options import HiDDenConfiguration
from .base_model import BaseModel
from .modules.Quantization import Quantization


def _get(d, *keys, default=None):
    for k in keys:
        if d is None:
            return default
        try:
            d = d[k]
        except (KeyError, TypeError, IndexError):
            return default
    return default if d is None else d


class DeferredLogs:
    """`logs` of a step whose scalars have not been waited for yet (train.deferred_logs; the reference's list of (name, value) pairs,
    IRNrhi_model.py optimize_parameters -> train.py:109).  Behaves as that list; the first look at it (iteration, len, truth value, indexing,
    comparison) waits for the device.  A loop that looks one step late -- after enqueueing the next step -- never drains the GPU queue."""

    def __init__(self, read):
        self._read, self._logs = read, None

    def resolve(self):
        if self._logs is None:
            self._logs, self._read = self._read(), None
        return self._logs

    def __iter__(self):
        return iter(self.resolve())

    def __len__(self):
        return len(self.resolve())

    def __getitem__(self, i):
        return self.resolve()[i]

    def __eq__(self, other):
        return self.resolve() == (other.resolve() if isinstance(other, DeferredLogs) else other)

    def __repr__(self):
        return repr(self.resolve())


class _AttackCycle:
    """the attack cycles deterministically with the step."""

    def __init__(self, layers):
        self.layers = layers
        self.k = 0
        self.name = "NotChosenYet"

    def _choose(self, id=None, exclude=()):
        i = self.k % len(self.layers) if id is None else id
        layer = self.layers[i]
        for _ in range(len(self.layers)):   # a layer type the caller cannot use (the localiser and Crop): take the next one in the cycle
            if not isinstance(layer, exclude):
                break
            i = (i + 1) % len(self.layers)
            layer = self.layers[i]
        else:
            i, layer = -1, Identity()
        self.name = getattr(layer, "name", type(layer).__name__)
        return i, layer

    def capture_key(self):
        """Hidden.enable_graph: every layer of the cycle runs with fixed arguments here (Resize at 0.7, Crop at a fixed rectangle), so a step
        through layer i launches the same kernels every time -- one captured graph per layer.  (Also names the layer, as fwd does: a
        replayed step does not call fwd.)"""
        i, layer = self._choose()
        return ("cycle", i, type(layer).__name__)

    def fwd(self, image, id=None, exclude=()):
        i, layer = self._choose(id, exclude)
        if isinstance(layer, Resize):
            y, c = layer.fwd(image, resize_ratio=0.7)
        elif isinstance(layer, Crop):
            H, W = image.shape[2], image.shape[3]
            y, c = layer.fwd(image, apex=(H // 8, H // 8 + int(0.75 * H), W // 8, W // 8 + int(0.75 * W)))
        else:
            y, c = layer.fwd(image)
        self.name = getattr(layer, "name", type(layer).__name__)   # (a layer may name itself in its forward: the reference's G_Blur -> GaussianBlur)
        return y, (layer, c)

    def bwd(self, ctx, g):
        layer, c = ctx
        return layer.bwd(c, g)


class IRNrhiModel(BaseModel):
    def __init__(self, opt):
        super(IRNrhiModel, self).__init__(opt)
        if self.device.type != "cuda":
            raise RuntimeError("IRNrhiModel runs on the MI355X HIP path only (opt['gpu_ids'] must not be None)")
        train_opt = opt['train'] or {}
        self.train_opt = train_opt
        self.rank = torch.distributed.get_rank() if opt['dist'] else -1
        if torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        size = _get(opt, 'datasets', 'train', 'GT_size', default=256)
        self.width_height = size
        self.global_step = 0
        self.real_H = self.mask = None
        self.previous_images = self.previous_previous_images = None
        self.save_interval = _get(train_opt, 'save_interval', default=3000)
        self.gradient_clipping = _get(train_opt, 'gradient_clipping', default=None)
        # train.deferred_logs (default false = the reference's behaviour: every step's scalars are read before optimize_parameters returns):
        # the returned logs wait for the device only when looked at (DeferredLogs)
        self.deferred_logs = bool(_get(train_opt, 'deferred_logs', default=False))
        dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16, "f32": torch.float32,
                 None: torch.bfloat16}[_get(train_opt, 'compute_dtype')]
        # torch.cuda.amp.autocast() + GradScaler() of the reference (IRNcrop_model.py:143,340,407-416): with f16 activations the
        # loss gradients are scaled by a device-side GradScaler (train.amp: true by default for f16, optional for the other dtypes)
        use_amp = _get(train_opt, 'amp', default=(dtype == torch.float16))
        self.amp = ops.AmpState(self.device, init_scale=_get(train_opt, 'amp_init_scale', default=65536.0),
                                growth_interval=_get(train_opt, 'amp_growth_interval', default=2000)) if use_amp else None

        # ---- attacks (IRNrhi_model.py:103-150 / IRNcrop_model.py:85-104) cycled per step
        attacks = _get(train_opt, 'attacks', default=None)
        table = {
            "Jpeg": lambda q: Jpeg(q), "JpegSS": lambda q: JpegSS(q), "JpegMask": lambda q: JpegMask(q),
            "GaussianBlur": lambda q: GaussianBlur(), "MiddleBlur": lambda q: MiddleBlur(q or 3),
            "Resize": lambda q: Resize(), "Crop": lambda q: Crop(), "Identity": lambda q: Identity(),
        }
        if attacks is None:
            attacks = ["Jpeg50"]
        layers = []
        for a in attacks:
            kind = "".join(ch for ch in a if not ch.isdigit())
            q = "".join(ch for ch in a if ch.isdigit())
            layers.append(table[kind](int(q) if q else None))
        self.attack = _AttackCycle(layers)
        self.Quantization = Quantization()

        # ---- networks: HiDDeN embedder / extractor / discriminator (+ UNet localiser)
        cfg = HiDDenConfiguration(H=size, W=size, message_length=_get(train_opt, 'message_length', default=30))
        grad_sync = None
        if opt['dist'] and torch.distributed.get_world_size() > 1:
            from ..distributed import GradSync
            grad_sync = GradSync()
        self.hidden = Hidden(cfg, self.device, self.attack, None, compute_dtype=dtype, grad_sync=grad_sync, amp=self.amp)
        # train.two_streams (default true): the step's two independent chains on two streams wherever a step has them to itself (one GPU, no
        # localisation branch / clipping / PSNR gate: Hidden._train_step_two_chains); same results bit for bit
        self.hidden.two_streams = bool(_get(train_opt, 'two_streams', default=True))
        # train.graph (default true): a step WITHOUT the localiser branch, gradient clipping and PSNR gate (configuration C3) is replayed from a
        # hipGraph, one graph per layer of the attack cycle (Hidden.enable_graph; same results bit for bit, tests/test_gpu_graph.py); one GPU only
        if grad_sync is None and bool(_get(train_opt, 'graph', default=True)):
            self.hidden.enable_graph()
        self.netG = self.hidden.encoder_decoder
        self.discriminator = self.hidden.discriminator
        lr = _get(train_opt, 'lr_G', default=1e-3)
        betas = (_get(train_opt, 'beta1', default=0.9), _get(train_opt, 'beta2', default=0.999))
        wd = _get(train_opt, 'weight_decay_G', default=0.0) or 0.0
        for o in (self.hidden.optimizer_enc_dec, self.hidden.optimizer_discrim):
            o.param_groups[0].update(lr=lr, initial_lr=lr, betas=betas, weight_decay=wd)
            o.decoupled = bool(_get(train_opt, 'adamw', default=False))
        self.optimizers = [self.hidden.optimizer_enc_dec, self.hidden.optimizer_discrim]
        self.use_localizer = bool(_get(train_opt, 'localizer', default=False))
        self.localizer = None
        if self.use_localizer:
            self.localizer = UNet(3, 1, 32).to(self.device)
            engine.set_compute_dtype(self.localizer, dtype)
            self.localizer.flatten_parameters_()
            self.optimizer_localizer = _FlatAdam([self.localizer], lr=lr, betas=betas, weight_decay=wd)
            if self.amp is not None:
                self.optimizer_localizer.attach_amp(self.amp)
            self.optimizers.append(self.optimizer_localizer)
            self.localizer_weight = _get(train_opt, 'localizer_weight', default=1.0)
        self.psnr_gate = bool(_get(train_opt, 'psnr_gate', default=True))   # IRNcrop_model.py:379-388
        # log side (IRNcrop_model.py:78,399-400: SummaryWriter scalars; :421-437: an image sheet every 500 steps at step % 500 == 10)
        tb_dir = _get(train_opt, 'tensorboard_dir', default=None)
        self.writer = None
        if tb_dir and self.rank <= 0:
            from ..utils import SummaryWriter
            self.writer = SummaryWriter(tb_dir)
        self.image_dump_interval = _get(train_opt, 'image_dump_interval', default=None)
        self.image_dump_dir = _get(self.opt, 'path', 'images', default=None)
        self._loc = None
        self._copy_stream = None
        self.messages = None
        self.keep_outputs = False    # tests / image dumps: keep the step's tensors in self.last_outputs
        self.last_outputs = {}
        if opt['dist']:
            from ..distributed import broadcast_parameters
            nets = [self.netG.encoder, self.netG.decoder, self.discriminator] + ([self.localizer] if self.localizer else [])
            broadcast_parameters(nets)
        self.grad_sync = grad_sync
        self.load()

    # ------------------------------------------------------------------ data
    def feed_data(self, batch):
        """Accepts what the reference's loaders produce for this path: a tensor [B,3,H,W] or clip
        [B,3,T,H,W] in [0,1], optionally with a tamper mask [B,1,(T,)H,W] -- as (imgs, mask), a dict
        {'GT': imgs, 'mask': mask} or the bare tensor.  Clips are folded into the batch (frames are
        independent units, IRNcrop_model.py:357-366)."""
        mask = None
        self.messages = None
        if isinstance(batch, dict):
            imgs, mask = batch.get('GT', batch.get('imgs')), batch.get('mask')
            if batch.get('messages') is not None:   # the watermark bits to embed ([B*T, L]); drawn at random per step when absent
                self.messages = batch['messages'].to(self.device, torch.float32).contiguous()
        elif isinstance(batch, (tuple, list)):
            imgs = batch[0]
            mask = batch[1] if len(batch) > 1 and torch.is_tensor(batch[1]) and batch[1].dim() >= 4 else None
        else:
            imgs = batch
        imgs, mask = self._to_device(imgs), (self._to_device(mask) if mask is not None else None)
        if imgs.dim() == 5:
            B, C, T, H, W = imgs.shape
            imgs = imgs.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
            if mask is not None:
                mask = mask.permute(0, 2, 1, 3, 4).reshape(B * T, 1, H, W)
        self.real_H = imgs.contiguous()
        self.mask = mask.contiguous() if mask is not None else None

    def _to_device(self, t):
        """host -> device.  A PINNED source (the training loader's default here, data/__init__.py) is copied on a side stream, so the copy
        runs beside whatever the step's stream still holds -- with train.deferred_logs that is the previous step -- and the step's stream
        only waits for the copy's event; a pageable source is copied in stream order (the host blocks: torch's semantics)."""
        if t.is_cuda or not t.is_pinned():
            return t.to(self.device, torch.float32)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._copy_stream):
            d = t.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        cur.wait_event(ev)
        d.record_stream(cur)
        return d if d.dtype == torch.float32 else d.float()

    # ------------------------------------------------------------------ localisation branch
    def _gate(self, encoded, images):
        """IRNcrop_model.py:344-348,379-388: one pass over the encoded batch forms Q(clamp(encoded)), the spliced (tampered)
        batch for _localise and the PSNR partial sums; the 1.0 / 0.8 forward weight stays on the device."""
        mask = self.mask
        if mask is None:
            mask = torch.zeros(encoded.shape[0], 1, encoded.shape[2], encoded.shape[3], device=encoded.device)
        _, tampered, part = ops.splice_fwd(encoded, real=images, prev=self.previous_images, mask=mask)
        self._loc = (tampered, mask)
        return ops.psnr_gate(part, encoded.numel(), 33.0, 1.0, 0.8)

    def _localise(self, encoded, images, g_enc):
        """IRNcrop_model.py:344-393 on the HIP kernels: STE clamp -> Quantization -> splice with the previous batch by the
        mask -> attack -> STE clamp -> Quantization -> UNet -> BCEWithLogits on the (sigmoid) mask.  Adds the gradient wrt
        `encoded` into g_enc (the splice passes g*(1-mask); clamp_with_grad and Quantization are identities backwards),
        steps the localiser, returns the logs."""
        net = self.localizer
        if self._loc is None:
            self._gate(encoded, images)
        (tampered, mask), self._loc = self._loc, None
        kind = self.attack.name                                     # the step's attack (set by the embed -> attack -> extract pass)
        attacked, cA = self.attack.fwd(tampered, exclude=(Crop,))   # the reference's localiser sees no geometric attack (:362-366)
        attacked_q = ops.clamp_quant(attacked)                      # clamp_with_grad + Quantization (:372-373)
        if self.keep_outputs:
            self.last_outputs.update(tampered=tampered, attacked=attacked_q)
        net.refresh_packs()   # all conv weights of the localiser packed in one launch, valid until its optimiser step
        try:
            pred, cU = net.fwd(attacked_q)
            if self.keep_outputs:
                self.last_outputs["pred"] = pred
            # the reference applies BCEWithLogits to the sigmoid output (:378,391-393); the kernel chains sigmoid'
            loss, g_logit = ops.bce_logits_target(pred, mask, self.localizer_weight, chain_sigmoid=True,
                                                  gscale_dev=self.amp.scale if self.amp is not None else None)
            grads = engine.grad_dict(net)
            # data parallel: the localiser's 31 MB of gradients leave in four reverse-order buckets from inside its backward
            gs, pending = self.grad_sync, []
            ready = (lambda lo, hi: pending.append(gs.start(net.flat_grads[lo:hi]))) if gs is not None else None
            g_att = net.bwd(cU, g_logit.view_as(pred), grads, accumulate=False, need_input_grad=True, g_is_logit=True, bucket_ready=ready)
        finally:
            net.invalidate_packs()
        gscale = 1.0
        if gs is not None:
            gs.finish_all(pending)
            gscale = gs.scale
            if self.gradient_clipping:
                gs.average_(net.flat_grads)
                gscale = 1.0
        self._clip([net.flat_grads])
        self.optimizer_localizer.step(grad_scale=gscale)
        g_tamp = self.attack.bwd(cA, g_att)
        ops.masked_axpy_(g_enc, g_tamp.contiguous(), mask)
        return [('lB', loss), ('CE', loss), ('Kind', kind), ('LocKind', self.attack.name)]

    def _clip(self, flats):
        if self.gradient_clipping:
            ops.clip_grad_norm_(flats, self.gradient_clipping)   # joint norm, clip coefficient stays on the device

    @torch.no_grad()
    def localise_mask(self, images, threshold=0.5):
        """integer tamper mask of a batch: the localiser in eval mode (running BatchNorm statistics), sigmoid output > threshold,
        uint8 [B,1,H,W]"""
        if self.localizer is None:
            raise RuntimeError("this model was built without the localiser (train.localizer)")
        x = images.to(self.device, torch.float32).contiguous()
        was = self.localizer.training
        self.localizer.eval()
        try:
            pred, _ = self.localizer.fwd(x, training=False)
        finally:
            self.localizer.train(was)
        return ops.mask_threshold(pred, threshold)

    # ------------------------------------------------------------------ the step
    def optimize_parameters(self, step, latest_values=None, train=True, eval_dir=None):
        self.global_step = self.global_step + 1
        logs, debug_logs = [], []
        self.real_H = torch.clamp(self.real_H, 0, 1)
        ready = self.previous_images is not None and self.previous_previous_images is not None
        if ready and train:
            B = self.real_H.shape[0]
            L = self.hidden.config.message_length
            messages = self.messages if self.messages is not None else torch.randint(0, 2, (B, L), device=self.device).float()
            self.attack.k = step
            extra = self._localise if self.use_localizer else None
            gate = self._gate if (self.use_localizer and self.psnr_gate) else None
            self._loc = None
            dump = bool(self.image_dump_interval and self.image_dump_dir and step % self.image_dump_interval == 10 % self.image_dump_interval)
            keep, self.keep_outputs = self.keep_outputs, self.keep_outputs or dump
            if self.keep_outputs:
                self.last_outputs = {}
            losses, outs = self.hidden.train_on_batch([self.real_H, messages], extra_encoded_grad=extra,
                                                      clip=self._clip if self.gradient_clipping else None, enc_gate=gate)
            if self.keep_outputs:
                self.last_outputs.update(encoded=outs[0], noised=outs[1], decoded=outs[2])
            self.keep_outputs = keep
            extra_logs = losses.pop('_extra', [])
            lr = self.get_current_learning_rate()

            def read_logs():   # waits for the step's scalars (the reference's .item() calls)
                out = [(k.strip(), v) for k, v in losses.items()]
                for name, v in extra_logs:
                    out.append((name, v.item() if torch.is_tensor(v) else v))
                out.append(('lr', lr))
                return out
            if self.deferred_logs and self.writer is None and not dump:
                logs = DeferredLogs(read_logs)   # train.deferred_logs: read when the caller first looks at them (train.py: one step later)
            else:
                logs = read_logs()
                self._log_side(step, logs)
        elif ready:
            L = self.hidden.config.message_length
            messages = torch.randint(0, 2, (self.real_H.shape[0], L), device=self.device).float()
            losses, _ = self.hidden.validate_on_batch([self.real_H, messages])
            logs = [(k.strip(), v) for k, v in losses.items()]
        # ---- finally (IRNrhi_model.py:690-698)
        if step % self.save_interval == 10 and self.rank <= 0 and train:
            self.save(self.global_step)
        if self.real_H is not None:
            if self.previous_images is not None:
                self.previous_previous_images = self.previous_images
            self.previous_images = self.real_H.clone().detach()
        return logs, debug_logs

    def _log_side(self, step, logs):
        """TensorBoard scalars of every float in `logs` and, every image_dump_interval steps, the stitched sheet
        input | watermarked | 10 x |difference| | attacked | predicted mask | mask (IRNcrop_model.py:399-400,421-437)"""
        if self.writer is not None:
            for name, v in logs:
                if isinstance(v, float):
                    self.writer.add_scalar(name, v, global_step=self.global_step)
            self.writer.flush()
        n = self.image_dump_interval
        if n and self.image_dump_dir and self.rank <= 0 and step % n == 10 % n and self.last_outputs.get("encoded") is not None:
            from ..utils import postprocess, stitch_images
            o = self.last_outputs
            x, enc = self.real_H[:4], o["encoded"][:4].clamp(0, 1)
            cols = [postprocess(enc), postprocess((10 * (x - enc).abs()).clamp(0, 1))]
            if o.get("attacked") is not None:
                cols += [postprocess(o["attacked"][:4].clamp(0, 1)), postprocess(o["pred"][:4].expand(-1, 3, -1, -1)),
                         postprocess(self.mask[:4].expand(-1, 3, -1, -1))]
            else:
                cols.append(postprocess(o["noised"][:4].clamp(0, 1)))
            sheet = stitch_images(postprocess(x), *cols, img_per_row=1)
            os.makedirs(self.image_dump_dir, exist_ok=True)
            sheet.save(os.path.join(self.image_dump_dir, str(step).zfill(5) + ".png"))

    def evaluate(self, *args, **kwargs):
        return self.optimize_parameters(self.global_step, train=False)

    # ------------------------------------------------------------------ checkpoints
    def _nets(self):
        nets = [(self.netG.encoder, 'encoder'), (self.netG.decoder, 'decoder'), (self.discriminator, 'discriminator')]
        if self.localizer is not None:
            nets.append((self.localizer, 'localizer'))
        return nets

    def save(self, iter_label):
        path = _get(self.opt, 'path', 'models', default=None)
        if path is None:
            return []
        return [self.save_network(net, label, iter_label, model_path=path) for net, label in self._nets()]

    def load(self):
        for net, label in self._nets():
            p = _get(self.opt, 'path', 'pretrain_model_' + label, default=None)
            if p and os.path.exists(p):
                self.load_network(p, net, _get(self.opt, 'path', 'strict_load', default=True))
