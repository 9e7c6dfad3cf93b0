"""BaseModel -- the checkpoint / learning-rate surface of the reference's models/base_model.py:8-150
that train.py and the model classes rely on: per-network `{iter}_{label}.pth` state_dict files
(`module.` prefixes stripped on load, tensors moved to CPU on save), `{iter}.state` training state,
warm-up aware `update_learning_rate`, `get_current_learning_rate`."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device('cuda' if opt['gpu_ids'] is not None else 'cpu')
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []

    def feed_data(self, data):
        pass

    def optimize_parameters(self, latest_values):
        pass

    def save(self, label):
        pass

    def load(self):
        pass

    # ---- learning rate
    def _set_lr(self, lr_groups_l):
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(optimizer.param_groups, lr_groups):
                group['lr'] = lr

    def _get_init_lr(self):
        return [[g['initial_lr'] for g in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        for s in self.schedulers:
            s.step()
        if cur_iter < warmup_iter:
            self._set_lr([[v / warmup_iter * cur_iter for v in groups] for groups in self._get_init_lr()])

    def get_current_learning_rate(self):
        return self.optimizers[0].param_groups[0]['lr']

    def get_network_description(self, network):
        network = getattr(network, "module", network)
        return str(network), sum(p.numel() for p in network.parameters())

    # ---- checkpoints
    def save_network(self, network, network_label, iter_label, save_dir=None, model_path=None):
        if model_path is None:
            model_path = self.opt['path']['models']
        if save_dir is None:
            save_path = os.path.join(model_path, '{}_{}.pth'.format(iter_label, network_label))
        else:
            save_path = os.path.join(save_dir, '{}_latest.pth'.format(network_label))
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        network = getattr(network, "module", network)
        state = OrderedDict((k, v.detach().cpu().clone()) for k, v in network.state_dict().items())
        torch.save(state, save_path)
        return save_path

    def load_network(self, load_path, network, strict=True):
        network = getattr(network, "module", network)
        loaded = torch.load(load_path, map_location="cpu")
        clean = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in loaded.items())
        network.load_state_dict(clean, strict=strict)  # copies in place: the flat parameter buffers stay valid

    def save_training_state(self, epoch, iter_step):
        state = {'epoch': epoch, 'iter': iter_step,
                 'schedulers': [s.state_dict() for s in self.schedulers],
                 'optimizers': [o.state_dict() for o in self.optimizers]}
        path = os.path.join(self.opt['path']['training_state'], '{}.state'.format(iter_step))
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save(state, path)
        return path

    def resume_training(self, resume_state):
        assert len(resume_state['optimizers']) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(resume_state['schedulers']) == len(self.schedulers), 'Wrong lengths of schedulers'
        for o, s in zip(self.optimizers, resume_state['optimizers']):
            o.load_state_dict(s)
        for sch, s in zip(self.schedulers, resume_state['schedulers']):
            sch.load_state_dict(s)
