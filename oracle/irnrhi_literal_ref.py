"""Oracle (test infrastructure): the reference's literal IRNrhi step (/root/reference/models/IRNrhi_model.py:425-560) composed on the
CPU from the functional networks of oracle/f1_ref.py (each pinned to reference-generated fixtures), torch's own loss modules
(nn.SmoothL1Loss, nn.BCELoss, nn.CrossEntropyLoss -- the classes the reference instantiates at :147-156) and torch.optim.AdamW
(:281-335).  The reference's step itself cannot run here (it needs CUDA, DistributedDataParallel and a '/home/qcying' dump
directory), so this composition is "parity unpinned" as a whole; every piece it calls is pinned.  Only tests/ may import this."""
import torch
import torch.nn as nn

from . import f1_ref


class LiteralRef:
    def __init__(self, gen_sd, loc_sd, dis_sd, nb, lr, betas=(0.9, 0.999), weight_decay=0.0, clip=None):
        self.g, self.l, self.d = f1_ref.params(gen_sd), f1_ref.params(loc_sd), f1_ref.params(dis_sd)
        self.nb, self.clip = nb, clip

        def leaves(sd):
            return [v for v in sd.values() if v.requires_grad]

        def opt(sd):
            return torch.optim.AdamW(leaves(sd), lr=lr, betas=betas, weight_decay=weight_decay)

        self.og, self.ol, self.od = opt(self.g), opt(self.l), opt(self.d)
        self.leaves = leaves
        self.l1, self.bce, self.ce = nn.SmoothL1Loss(), nn.BCELoss(), nn.CrossEntropyLoss()

    def _clip(self, sd):
        if self.clip:
            nn.utils.clip_grad_norm_(self.leaves(sd), self.clip)

    def step(self, imgs):
        real_H = torch.clamp(torch.cat(imgs, 0), 0, 1)
        bs = imgs[0].shape[0]
        label = torch.arange(6).repeat_interleave(bs)
        bayar_ori, qf = f1_ref.qf_predictor(self.l, real_H, self.nb)
        bayar_ori = bayar_ori.clone().detach()
        l_qf_r = self.ce(qf, label)
        self.ol.zero_grad()
        l_qf_r.backward()
        self._clip(self.l)
        self.ol.step()
        self.ol.zero_grad()

        label_input = (label / 5).float().unsqueeze(1)
        sim, _ = f1_ref.fbcnn(self.g, real_H[0:bs].repeat(6, 1, 1, 1), label_input, self.nb)
        sim = torch.clamp(sim, 0, 1)
        l_simul_l1 = self.l1(sim, real_H)
        bayar_s, qf_s = f1_ref.qf_predictor(self.l, sim, self.nb)
        l_bayar = self.l1(bayar_s, bayar_ori)
        l_simul_l1 = l_simul_l1 + 5.0 * l_bayar
        l_qf_s = self.ce(qf_s, label)

        dr = f1_ref.discriminator(self.d, real_H, training=True)
        df = f1_ref.discriminator(self.d, sim.detach(), training=True)
        dis_loss = (self.bce(dr, torch.ones_like(dr)) + self.bce(df, torch.zeros_like(df))) / 2
        self.od.zero_grad()
        dis_loss.backward()
        self._clip(self.d)
        self.od.step()
        self.od.zero_grad()

        gf = f1_ref.discriminator(self.d, sim, training=True)
        fw_gan = self.bce(gf, torch.ones_like(gf))
        total = l_simul_l1 + l_qf_s * 0.01 + fw_gan * 0.01
        total.backward()
        self._clip(self.g)
        self.og.step()
        self.og.zero_grad()

        a, b = (sim.detach() * 255.0).int().float(), (real_H * 255.0).int().float()
        mse = torch.mean((a - b) ** 2)
        psnr = 0.0 if mse == 0 else float(20 * torch.log10(torch.tensor(255.0)) - 10 * torch.log10(mse))
        return {"l_simul_bayar": l_bayar.item(), "FW_GAN": fw_gan.item(), "lQF": l_qf_r.item(), "PSSIMU": psnr, "qfsimu": l_qf_s.item(),
                "dis_loss": dis_loss.item(), "l_simul_sum": total.item(), "simulated": sim.detach()}
