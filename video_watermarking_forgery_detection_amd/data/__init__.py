"""create dataset and dataloader -- mirror of the reference's data/__init__.py:7-28: in distributed training the global batch is split
evenly over the ranks (`batch_size // world_size`, asserted divisible), `drop_last=True`, no shuffling beside the sampler."""
import torch
import torch.utils.data


def create_dataloader(dataset, dataset_opt, opt=None, sampler=None):
    phase = opt['phase'] if opt is not None and opt['phase'] else 'train'
    if phase == 'train':
        if opt['dist']:
            world_size = torch.distributed.get_world_size()
            num_workers = dataset_opt['n_workers']
            assert dataset_opt['batch_size'] % world_size == 0
            batch_size = dataset_opt['batch_size'] // world_size
            shuffle = False
        else:
            num_workers = dataset_opt['n_workers'] * len(opt['gpu_ids'] or [0])
            batch_size = dataset_opt['batch_size']
            shuffle = sampler is None
        # pin_memory: the reference passes False (data/__init__.py:25); here the default is True -- feed_data's non_blocking copy then does not
        # wait for the step in flight (a pageable source makes the copy a per-step drain of the stream); `pin_memory: false` restores it
        pin = dataset_opt['pin_memory']
        return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, sampler=sampler,
                                           drop_last=True, pin_memory=torch.cuda.is_available() if pin is None else bool(pin))
    return torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=1, pin_memory=True)


from .Dataloader import DVDataset  # noqa: E402,F401
from .data_sampler import DistIterSampler  # noqa: E402,F401
from .masks import generate_stroke_mask  # noqa: E402,F401
