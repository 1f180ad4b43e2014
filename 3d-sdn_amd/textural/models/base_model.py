"""BaseModel of the textural branch (reference: textural/models/base_model.py): option bookkeeping, checkpoint file
naming `{epoch}_net_{G,D,E}.pth` holding plain state_dicts, and the tolerant loader (strips a DataParallel `module.`
prefix, loads the intersection when keys or shapes differ)."""
import os
from collections import OrderedDict

import torch


class BaseModel(torch.nn.Module):
    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.Tensor = torch.cuda.FloatTensor if self.gpu_ids else torch.Tensor
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def test(self):
        pass

    def get_image_paths(self):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        return self.input

    def get_current_errors(self):
        return {}

    def save(self, label):
        pass

    def save_network(self, network, network_label, epoch_label, gpu_ids):
        """base_model.py:47-52; the weights are copied to the host without moving the live network off the GPU."""
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        torch.save(OrderedDict((k, v.detach().cpu()) for k, v in network.state_dict().items()), path)

    def load_network(self, network, network_label, epoch_label, save_dir=''):
        """base_model.py:55-95."""
        path = os.path.join(save_dir or self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        if not os.path.isfile(path):
            print('%s not exists yet!' % path)
            if network_label == 'G':
                raise FileNotFoundError('Generator must exist!')
            return
        print(path)
        loaded = torch.load(path, map_location='cpu')
        if next(iter(loaded.keys())).startswith('module'):
            loaded = OrderedDict((k[7:], v) for k, v in loaded.items())
        own = network.state_dict()
        if set(loaded) == set(own) and all(loaded[k].shape == own[k].shape for k in own):
            network.load_state_dict(loaded)
            return
        usable = {k: v for k, v in loaded.items() if k in own and v.shape == own[k].shape}
        missing = sorted({k.split('.')[0] for k in own if k not in usable})
        if len(usable) < len(loaded):
            print('Pretrained network %s has excessive layers; Only loading layers that are used' % network_label)
        if missing:
            print('Pretrained network %s has fewer layers; The following are not initialized:' % network_label)
            print(missing)
        own.update(usable)
        network.load_state_dict(own)

    def get_z_random(self, batchSize, nz, random_type='gauss'):
        dev = 'cuda' if self.gpu_ids else 'cpu'
        if random_type == 'uni':
            return torch.rand(batchSize, nz, device=dev) * 2.0 - 1.0
        return torch.randn(batchSize, nz, device=dev)

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name in self.model_names:
            if isinstance(name, str):
                net = getattr(self, 'net' + name)
                n = sum(p.numel() for p in net.parameters())
                if verbose:
                    print(net)
                print('[Network %s] Total number of parameters : %.3f M' % (name, n / 1e6))
        print('-----------------------------------------------')
