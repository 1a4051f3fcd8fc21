"""BaseModel: bookkeeping shared by the models (reference models/base_model.py): option capture, device
choice, checkpoint save / tolerant load with the reference's file naming ``{epoch}_net_{label}.pth``."""
import os

import torch

from .. import _ops


class BaseModel(torch.nn.Module):
    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt, self.isTrain, self.gpu_ids = opt, opt.isTrain, list(opt.gpu_ids)
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        # the reference falls back to the CPU without gpu_ids; this build has no CPU path and says so at first use
        self.device = torch.device('cuda', self.gpu_ids[0]) if self.gpu_ids else torch.device('cpu')
        self.Tensor = torch.cuda.FloatTensor if self.gpu_ids else torch.Tensor

    def save_network(self, network, network_label, epoch_label, gpu_ids):
        """fp32 CPU state_dict under the reference's file name; the network itself stays on the GPU."""
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        torch.save({k: v.detach().float().cpu().contiguous() for k, v in network.state_dict().items()}, path)   # (K-major parameters are permuted views)

    def load_network(self, network, network_label, epoch_label, save_dir=''):
        """Three-tier tolerant load of the reference (base_model.py:51-89): exact -> keys that exist here ->
        keys whose shapes match; reports layers left at their initial values."""
        path = os.path.join(save_dir or self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        if not os.path.isfile(path):
            print('%s not exists yet!' % path)
            if network_label == 'G':
                raise FileNotFoundError('Generator must exist!')
            return
        loaded = torch.load(path, map_location='cpu')
        own = network.state_dict()
        try:
            network.load_state_dict(loaded)
        except Exception:
            subset = {k: v for k, v in loaded.items() if k in own}
            try:
                network.load_state_dict(subset)
                if getattr(self.opt, 'verbose', False):
                    print('Pretrained network %s has excessive layers; Only loading layers that are used' % network_label)
            except Exception:
                print('Pretrained network %s has fewer layers; The following are not initialized:' % network_label)
                merged = dict(own)
                for k, v in loaded.items():
                    if k in own and v.size() == own[k].size():
                        merged[k] = v
                missing = sorted({k.split('.')[0] for k, v in own.items()
                                  if k not in loaded or v.size() != loaded[k].size()})
                print(missing)
                network.load_state_dict(merged)
        _ops.bump_weight_epoch()
        # captured steps read the bf16 forward images the optimiser's update kernel keeps (optim.FlatAdam): weights that
        # changed any other way need a fresh capture (and one eager update to refresh the images)
        self._graph_state = None
