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


class _AttackCycle:
    """the attack cycles deterministically with the step."""

    def __init__(self, layers):
        self.layers = layers
        self.k = 0
        self.name = "NotChosenYet"

    def fwd(self, image, id=None):
        i = self.k % len(self.layers) if id is None else id
        layer = self.layers[i]
        if isinstance(layer, Resize):
            y, c = layer.fwd(image, resize_ratio=0.7)
        elif isinstance(layer, Crop):
            H, W = image.shape[2], image.shape[3]
            y, c = layer.fwd(image, apex=(H // 8, H // 8 + int(0.75 * H), W // 8, W // 8 + int(0.75 * W)))
        else:
            y, c = layer.fwd(image)
        self.name = getattr(layer, "name", type(layer).__name__)
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
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32, None: torch.bfloat16}[_get(train_opt, 'compute_dtype')]

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
        self.hidden = Hidden(cfg, self.device, self.attack, None, compute_dtype=dtype, grad_sync=grad_sync)
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
            self.optimizers.append(self.optimizer_localizer)
            self.localizer_weight = _get(train_opt, 'localizer_weight', default=1.0)
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
        if isinstance(batch, dict):
            imgs, mask = batch.get('GT', batch.get('imgs')), batch.get('mask')
        elif isinstance(batch, (tuple, list)):
            imgs = batch[0]
            mask = batch[1] if len(batch) > 1 and torch.is_tensor(batch[1]) and batch[1].dim() >= 4 else None
        else:
            imgs = batch
        imgs = imgs.to(self.device, torch.float32, non_blocking=True)
        if imgs.dim() == 5:
            B, C, T, H, W = imgs.shape
            imgs = imgs.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
            if mask is not None:
                mask = mask.permute(0, 2, 1, 3, 4).reshape(B * T, 1, H, W)
        self.real_H = imgs.contiguous()
        self.mask = mask.to(self.device, torch.float32).contiguous() if mask is not None else None

    # ------------------------------------------------------------------ localisation branch
    def _localise(self, encoded, images):
        """IRNcrop_model.py:344-393 on the HIP kernels.  Returns (gradient wrt `encoded`, logs)."""
        net = self.localizer
        prev = self.previous_images
        mask = self.mask
        if mask is None:
            mask = torch.zeros(encoded.shape[0], 1, encoded.shape[2], encoded.shape[3], device=encoded.device)
        inside = ((encoded >= 0) & (encoded <= 1)).float()      # kept for reference: STE clamp passes grad everywhere
        fwd_img = ops.quant(encoded.clamp(0, 1))                  # clamp_with_grad (:320-322) + Quantization (:345)
        tampered = fwd_img * (1 - mask) + prev * mask           # splice (:348)
        attacked, cA = self.attack.fwd(tampered)
        attacked_q = ops.quant(attacked)                          # Quantization (:373)
        net.refresh_packs()   # all conv weights of the localiser packed in one launch, valid until its optimiser step
        try:
            pred, cU = net.fwd(attacked_q)
            # reference applies BCEWithLogits to the sigmoid output (:378,391-393)
            loss = F.binary_cross_entropy_with_logits(pred, mask)
            g_pred = (torch.sigmoid(pred) - mask) / pred.numel() * self.localizer_weight
            grads = engine.grad_dict(net)
            g_att = net.bwd(cU, g_pred, grads, accumulate=False, need_input_grad=True)
        finally:
            net.invalidate_packs()
        if self.grad_sync is not None:
            self.grad_sync(net.flat_grads)
        self._clip(net.flat_grads)
        self.optimizer_localizer.step()
        g_tamp = self.attack.bwd(cA, g_att)                        # Quantization backward = identity
        g_enc = g_tamp * (1 - mask)                              # splice; STE clamp + Quantization: identity
        del inside
        return g_enc.contiguous(), [('CE', loss), ('Kind', self.attack.name)]

    def _clip(self, flat):
        if self.gradient_clipping:
            norm = ops.sumsq(flat).sum().sqrt()
            flat.mul_(torch.clamp(self.gradient_clipping / (norm + 1e-6), max=1.0))  # clip_grad_norm_ without a host sync

    # ------------------------------------------------------------------ the step
    def optimize_parameters(self, step, latest_values=None, train=True, eval_dir=None):
        self.global_step = self.global_step + 1
        logs, debug_logs = [], []
        self.real_H = torch.clamp(self.real_H, 0, 1)
        ready = self.previous_images is not None and self.previous_previous_images is not None
        if ready and train:
            B = self.real_H.shape[0]
            L = self.hidden.config.message_length
            messages = torch.randint(0, 2, (B, L), device=self.device).float()
            self.attack.k = step
            extra = self._localise if self.use_localizer else None
            losses, _ = self.hidden.train_on_batch([self.real_H, messages], extra_encoded_grad=extra,
                                                   clip=self._clip if self.gradient_clipping else None)
            extra_logs = losses.pop('_extra', [])
            logs = [(k.strip(), v) for k, v in losses.items()]
            for name, v in extra_logs:
                logs.append((name, v.item() if torch.is_tensor(v) else v))
            logs.append(('lr', self.get_current_learning_rate()))
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
