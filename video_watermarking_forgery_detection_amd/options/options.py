"""YAML options -> NoneDict, the keys of the reference's options/options.py:9-101 that the step reads
(`train.*`, `datasets.train.GT_size/batch_size`, `path.*`, `gpu_ids`, `dist`, `is_train`).  A missing
key reads as None (NoneDict, options.py:86-101)."""
import os

import yaml


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def parse(opt_path, is_train=True):
    with open(opt_path, "r") as f:
        opt = yaml.safe_load(f)
    opt['is_train'] = is_train
    if opt.get('gpu_ids') is not None and 'CUDA_VISIBLE_DEVICES' not in os.environ and opt.get('set_visible_devices'):
        os.environ['CUDA_VISIBLE_DEVICES'] = ','.join(str(x) for x in opt['gpu_ids'])  # options.py:13-15
    root = opt.get('path', {}).get('root') or os.getcwd()
    name = opt.get('name', 'experiment')
    path = opt.setdefault('path', {})
    exp = os.path.join(root, 'experiments', name)
    path.setdefault('experiments_root', exp)
    path.setdefault('models', os.path.join(exp, 'models'))
    path.setdefault('training_state', os.path.join(exp, 'training_state'))
    path.setdefault('log', exp)
    opt.setdefault('dist', False)
    return dict_to_nonedict(opt)
