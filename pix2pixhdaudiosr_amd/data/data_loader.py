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
    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt):
        self.opt = opt
        self.dataset = CreateDataset(opt)
        dataset_size = len(self.dataset)
        indices = list(range(dataset_size))
        if opt.phase == "train":
            split = int(opt.validation_split * dataset_size)
            if getattr(opt, 'val_indices', None) is not None:
                self.val_indices = torch.load(opt.val_indices)
                self.train_indices = sorted(set(indices) - set(int(i) for i in self.val_indices))
            else:
                if not opt.serial_batches:
                    random.seed(opt.seed)
                    random.shuffle(indices)
                self.train_indices, self.val_indices = indices[split:], indices[:split]
                out_dir = os.path.join(opt.checkpoints_dir, opt.name)
                os.makedirs(out_dir, exist_ok=True)
                torch.save(self.val_indices, os.path.join(out_dir, 'validation_indices.pt'))
            self.data_lenth = min(len(self.train_indices), opt.max_dataset_size)
            feeder = GpuFeeder(opt)
            mk = lambda idx: _FedLoader(torch.utils.data.DataLoader(
                self.dataset, batch_size=opt.batchSize, sampler=SubsetRandomSampler(idx), num_workers=int(opt.nThreads),
                pin_memory=True), feeder)
            self.dataloader = mk(self.train_indices)
            self.eval_dataloder = mk(self.val_indices) if len(self.val_indices) != 0 else None
            self.eval_data_lenth = len(self.val_indices)
        else:
            self.data_lenth = min(dataset_size, opt.max_dataset_size)
            # the segments already live on the GPU: no workers, no pinning
            self.dataloader = _FedLoader(torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, num_workers=0,
                                                                     shuffle=False), None)
            self.eval_dataloder = None
            self.eval_data_lenth = 0

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
