"""CreateDataLoader with the reference's names (data/data_loader.py, data/custom_dataset_data_loader.py)."""
import os
import random

import torch
import torch.utils.data
from torch.utils.data import SubsetRandomSampler

from .audio_dataset import AudioDataset, AudioTestDataset, GpuFeeder


def CreateDataset(opt):
    if opt.phase == 'train':
        dataset = AudioDataset(opt)
    elif opt.phase == 'test':
        dataset = AudioTestDataset(opt)
    else:
        raise ValueError("phase [%s] not recognized." % opt.phase)
    print("dataset [%s] was created" % (dataset.name()))
    return dataset


class _FedLoader:
    """A DataLoader whose batches pass through the GPU feeder stage on the consuming process."""

    def __init__(self, loader, feeder):
        self.loader, self.feeder = loader, feeder

    def __iter__(self):
        for batch in self.loader:
            yield self.feeder(batch) if self.feeder is not None else batch

    def __len__(self):
        return len(self.loader)


class CustomDatasetDataLoader:
    """Train: shuffled train / validation split (indices saved next to the checkpoints) behind worker processes that
    only read PCM, then the GPU feeder stage; test: the segments already live on the GPU."""

    def name(self):
        return 'CustomDatasetDataLoader'

    def _loader(self, indices, feeder):
        opt = self.opt
        return _FedLoader(torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, sampler=SubsetRandomSampler(indices),
                                                      num_workers=int(opt.nThreads), pin_memory=True), feeder)

    def _split(self, n):
        opt = self.opt
        given = getattr(opt, 'val_indices', None)
        if given is not None:
            val = [int(i) for i in torch.load(given)]
            held = set(val)
            return [i for i in range(n) if i not in held], val
        order = list(range(n))
        if not opt.serial_batches:
            random.Random(opt.seed).shuffle(order)
        cut = int(opt.validation_split * n)
        folder = os.path.join(opt.checkpoints_dir, opt.name)
        os.makedirs(folder, exist_ok=True)
        torch.save(order[:cut], os.path.join(folder, 'validation_indices.pt'))
        return order[cut:], order[:cut]

    def initialize(self, opt):
        self.opt = opt
        self.dataset = CreateDataset(opt)
        n = len(self.dataset)
        if opt.phase == "train":
            self.train_indices, self.val_indices = self._split(n)
            feeder = GpuFeeder(opt)
            self.dataloader = self._loader(self.train_indices, feeder)
            self.eval_dataloder = self._loader(self.val_indices, feeder) if self.val_indices else None
            self.data_lenth = min(len(self.train_indices), opt.max_dataset_size)
            self.eval_data_lenth = len(self.val_indices)
        else:
            self.dataloader = _FedLoader(torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, num_workers=0,
                                                                     shuffle=False), None)
            self.eval_dataloder, self.eval_data_lenth = None, 0
            self.data_lenth = min(n, opt.max_dataset_size)

    def load_data(self):
        return self.dataloader

    def eval_data(self):
        return self.eval_dataloder

    def eval_data_len(self):
        return self.eval_data_lenth

    def __len__(self):
        return self.data_lenth


def CreateDataLoader(opt):
    data_loader = CustomDatasetDataLoader()
    print(data_loader.name())
    data_loader.initialize(opt)
    return data_loader
