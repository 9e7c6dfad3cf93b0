"""The reference's IRNrhiModel.optimize_parameters AS WRITTEN (models/IRNrhi_model.py:425-560): the JPEG-simulation step that trains
a QF_predictor ("localizer"), an FBCNN ("generator") and a spectral-norm Discriminator on six quality copies of a batch -- SURVEY 8f
row 1's "literal IRNrhi step" -- on the HIP layer toolkit (glayers.py; csrc/gconv.hip, gelem.hip, losses.hip, optim.hip).

    bayar_ori, QF = localizer(real_H);  CE(QF, label) -> clip -> AdamW(localizer)                                       :452-463
    simulated = clamp(generator(real_H[0:bs].repeat(6), label / 5));  SmoothL1(simulated, real_H)                       :465-476
    bayar_s, QF_s = localizer(simulated);  + 5 SmoothL1(bayar_s, bayar_ori);  CE(QF_s, label)                           :479-485
    (BCE(D(real_H), 1) + BCE(D(simulated.detach()), 0)) / 2 -> clip -> AdamW(discriminator)                             :488-501
    l_simul + 0.01 CE + 0.01 BCE(D(simulated), 1) -> clip -> AdamW(generator);  PSNR(simulated, real_H)                 :503-527

What is not carried over: `self.netG` (a second FBCNN the reference constructs, :181, whose training lines are commented out, :465,
:516-520), the DiffJPEG copies `self.diff_jpeg` that only feed the stitched image dump (:376-387,536-556), DistributedDataParallel
wrappers (the build's data parallelism is distributed.GradSync).  Logs are the reference's (name, value) pairs; they are read back
with ONE host synchronisation per step instead of five `.item()` calls.
"""
import torch

from .. import glayers as G
from .. import ops
from .base_model import BaseModel
from .conditional_jpeg_generator import FBCNN, QF_predictor
from .networks import Discriminator


def _get(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


class IRNrhiLiteralModel(BaseModel):
    QUALITIES = 6

    def __init__(self, opt):
        super().__init__(opt)
        if self.device.type != "cuda":
            raise RuntimeError("IRNrhiLiteralModel runs on the MI355X HIP path only (opt['gpu_ids'] must not be None)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        train_opt = opt.get('train') or {}
        self.train_opt = train_opt
        net_opt = opt.get('network') or {}
        dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16, "f32": torch.float32, None: torch.float32}[
            _get(train_opt, 'compute_dtype')]
        nc, nb = _get(net_opt, 'nc', default=[32, 64, 128, 256]), _get(net_opt, 'nb', default=4)
        self.generator = FBCNN(nc=list(nc), nb=nb, dtype=dtype).to(self.device)                                  # :162
        self.localizer = QF_predictor(in_nc=3, classes=self.QUALITIES, nc=list(nc), nb=nb, dtype=dtype).to(self.device)   # :166
        self.discriminator = Discriminator(in_channels=3, use_SRM=False, dtype=dtype).to(self.device)            # :177
        wd = _get(train_opt, 'weight_decay_G', default=0.0) or 0.0
        betas = (_get(train_opt, 'beta1', default=0.9), _get(train_opt, 'beta2', default=0.999))
        lr_d = _get(train_opt, 'lr_D', default=1e-4)
        # :281-335 -- torch.optim.AdamW(lr=lr_D, weight_decay=wd_G, betas) for each of the three networks
        self._pack_plan = ops.PackPlan()
        self.optimizer_generator = G.FlatAdamW(self.generator, lr_d, betas, weight_decay=wd)
        self.optimizer_discriminator = G.FlatAdamW(self.discriminator, lr_d, betas, weight_decay=wd)
        self.optimizer_localizer = G.FlatAdamW(self.localizer, lr_d, betas, weight_decay=wd)
        self.gradient_clipping = _get(train_opt, 'gradient_clipping', default=None)
        self.global_step = 0
        self.real_H = self.label = None
        self.last = {}

    # ------------------------------------------------------------------ data
    def feed_data(self, batch):
        """batch = (imgs, label) with imgs a list of six [bs,3,H,W] tensors, one per JPEG quality class (:369-379)"""
        imgs, _ = batch
        if len(imgs) != self.QUALITIES:
            self.real_H = None
            return
        self.real_H = torch.cat([i.to(self.device, non_blocking=True).float() for i in imgs], dim=0)
        bs = imgs[0].shape[0]
        self.label = torch.arange(self.QUALITIES, device=self.device).repeat_interleave(bs)     # [0]*bs + [1]*bs + ... (:370-376 for bs = 4)

    def _clip(self, optimizer):
        if self.gradient_clipping:
            optimizer.clip_grad_norm_(self.gradient_clipping)

    # ------------------------------------------------------------------ the step
    def optimize_parameters(self, step, latest_values=None, train=True, eval_dir=None):
        prev = G._PLAN
        G.set_pack_plan(self._pack_plan)       # the 3x3 layers' packed weights: persistent, re-packed by one launch per optimiser step
        try:
            return self._optimize_parameters(step, latest_values, train, eval_dir)
        finally:
            G.set_pack_plan(prev)

    def _optimize_parameters(self, step, latest_values=None, train=True, eval_dir=None):
        self.global_step += 1
        if self.real_H is None:
            return [], []
        real_H = ops.clamp01_fwd(self.real_H)                                                    # :430
        batch_size = real_H.shape[0] // self.QUALITIES
        label = self.label
        for net in (self.generator, self.localizer, self.discriminator):
            net.train()

        # ---- localizer on the real copies (:452-463)
        bayar_ori, QF_r2 = self.localizer(real_H)
        bayar_ori = bayar_ori.detach()
        l_qf_r = G.cross_entropy_loss(QF_r2, label)
        self.optimizer_localizer.zero_grad()
        l_qf_r.backward()
        self._clip(self.optimizer_localizer)
        self.optimizer_localizer.step()
        self.optimizer_localizer.zero_grad()

        # ---- simulate the six qualities from the first copy (:465-485)
        label_input = (label / 5).float().unsqueeze(1)
        simulation_input = real_H[0:batch_size].repeat(self.QUALITIES, 1, 1, 1)
        simulated_jpeg, _simul_feats = self.generator(simulation_input, label_input)
        simulated_jpeg = G.clamp01(simulated_jpeg)
        l_simul_l1 = G.smooth_l1_loss(simulated_jpeg, real_H)
        bayar_simul, QF_simul = self.localizer(simulated_jpeg)
        l_simul_bayar = G.smooth_l1_loss(bayar_simul, bayar_ori)
        l_simul_l1 = l_simul_l1 + 5.0 * l_simul_bayar
        l_QF_simul = G.cross_entropy_loss(QF_simul, label)

        # ---- discriminator (:488-501)
        dis_real = self.discriminator(real_H)
        dis_fake = self.discriminator(simulated_jpeg.detach())
        dis_loss = (G.bce_loss(dis_real, 1.0) + G.bce_loss(dis_fake, 0.0)) / 2
        self.optimizer_discriminator.zero_grad()
        dis_loss.backward()
        self._clip(self.optimizer_discriminator)
        self.optimizer_discriminator.step()
        self.optimizer_discriminator.zero_grad()

        # ---- generator: fidelity + quality classification + adversarial (:503-525)
        gen_fake = self.discriminator(simulated_jpeg)
        FW_GAN = G.bce_loss(gen_fake, 1.0)
        l_simul_sum = l_simul_l1 + l_QF_simul * 0.01 + FW_GAN * 0.01
        l_simul_sum.backward()
        self._clip(self.optimizer_generator)
        self.optimizer_generator.step()
        self.optimizer_generator.zero_grad()

        PSSIMU = ops.psnr255(simulated_jpeg.detach(), real_H)                                    # :527
        vals = torch.stack([l_simul_bayar.detach(), FW_GAN.detach(), l_qf_r.detach(), PSSIMU.reshape(()), l_QF_simul.detach(),
                            dis_loss.detach(), l_simul_sum.detach()]).cpu().tolist()             # the step's one host synchronisation
        names = ['l_simul_bayar', 'FW_GAN', 'lQF', 'PSSIMU', 'qfsimu']                           # :482,506,529-532, in the order appended
        logs = list(zip(names, vals[:5]))
        self.last = {"dis_loss": vals[5], "l_simul_sum": vals[6], "simulated": simulated_jpeg.detach()}
        return logs, []

    # ------------------------------------------------------------------ checkpoints (base_model.py:77-115)
    def save(self, iter_label):
        for net, name in ((self.generator, 'generator'), (self.localizer, 'localizer'), (self.discriminator, 'discriminator')):
            self.save_network(net, name, iter_label)
