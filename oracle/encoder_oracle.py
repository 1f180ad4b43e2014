"""CPU restatement of the derender3d encoder (TEST INFRASTRUCTURE ONLY: imported by tests/ and tests/golden/ -- the product
path is 3d-sdn_amd/geometric/derender3d/models/{resnet,derenderer}.py on the HIP kernels).

The reference's Derenderer (geometric/derender3d/models/derenderer.py:7-65) is `torchvision.models.resnet18` with its
avgpool / fc replaced, followed by three Linear layers.  torchvision (pinned 0.2.1 by the reference's environment.yml) is
an un-vendored third-party dependency and is absent from this image, so its ResNet-18 is restated here from the published
architecture (torchvision/models/resnet.py, v0.2.1: conv7x7 s2 p3 - bn - relu - maxpool3x3 s2 p1 - 4 stages of 2
BasicBlocks [64, 128, 256, 512], strides [1, 2, 2, 2], 1x1 stride-s downsample + bn where the shape changes - avgpool - fc;
init: conv weights N(0, sqrt(2 / (k*k*out_channels))), bn weight 1 / bias 0).  **Parity of the backbone is therefore
unpinned** (no torchvision to run); the Derenderer head IS pinned: tests/golden/make_encoder_golden.py runs the
reference's own Derenderer class on top of `RefResNet18` below.

  resnet18_features(sd, x, training, ...)   functional forward on a state_dict (any float dtype: the tests run it in fp64)
  derenderer_forward(sd, images, mroi, droi, training)   derenderer.py:37-65 restated on a state_dict
  RefResNet18                                 nn.Module with torchvision's attribute names / creation order / init, whose
                                              forward is the functional restatement (used to host the reference class)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))


def _bn(sd, prefix, x, training, momentum, eps, update):
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    if not update:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'], training, momentum, eps)


def resnet18_features(sd, x, training=False, momentum=0.1, eps=1e-5, prefix='', update_running=False, taps=None,
                      relu_masks=None):
    """x [N, 3, H, W] -> [N, 512] (after global average pooling).  sd: torchvision-keyed state_dict (tensors of x's dtype;
    with training=True and update_running=True the running statistics in sd are updated in place, as the modules would).
    relu_masks: optional list of 0/1 tensors, one per ReLU in execution order (stem, then bn1-relu and output-relu of each
    block); ReLU(t) is then evaluated as t * mask -- the tests use it to compare BACKWARD passes under identical
    activation patterns (a forward difference of 1e-5 otherwise flips a few units, each flip a finite gradient change)."""
    p = prefix
    bn = lambda name, t: _bn(sd, p + name, t, training, momentum, eps, update_running)
    masks = iter(relu_masks) if relu_masks is not None else None

    class _R:   # F.relu or the masked product
        @staticmethod
        def relu(t):
            return F.relu(t) if masks is None else t * next(masks).to(t.dtype)
    F_relu = _R.relu
    x = F.conv2d(x, sd[p + 'conv1.weight'], None, 2, 3)
    x = F_relu(bn('bn1', x))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps['stem'] = x
    inplanes = 64
    for li, (planes, stride) in enumerate(STAGES, 1):
        for bi in range(2):
            s = stride if bi == 0 else 1
            name = 'layer%d.%d.' % (li, bi)
            idt = x
            if bi == 0 and (s != 1 or inplanes != planes):
                idt = bn(name + 'downsample.1', F.conv2d(x, sd[p + name + 'downsample.0.weight'], None, s, 0))
            out = F_relu(bn(name + 'bn1', F.conv2d(x, sd[p + name + 'conv1.weight'], None, s, 1)))
            out = bn(name + 'bn2', F.conv2d(out, sd[p + name + 'conv2.weight'], None, 1, 1))
            x = F_relu(out + idt)
            if taps is not None:
                taps[name[:-1]] = x
        inplanes = planes
    return x.mean(dim=(2, 3))


OUT_SIZES = (('_theta_deltas', 2), ('_translation2ds', 2), ('_log_scales', 3), ('_log_depths', 1))


def derenderer_forward(sd, images, mroi_norms, droi_norms, training=False, num_classes=8, grid_size=4, update_running=False):
    """derenderer.py:37-65 on a state_dict with the reference's keys (net.*, fc1, fc2, _fc3)."""
    x = resnet18_features(sd, images, training, prefix='net.', update_running=update_running)
    x = F.relu(F.linear(x, sd['net.fc.weight'], sd['net.fc.bias']))
    x = torch.cat([x, mroi_norms, droi_norms], dim=1)
    x = F.relu(F.linear(x, sd['fc1.weight'], sd['fc1.bias']))
    x = F.relu(F.linear(x, sd['fc2.weight'], sd['fc2.bias']))
    x = F.linear(x, sd['_fc3.weight'], sd['_fc3.bias'])
    sizes = [w for _, w in OUT_SIZES] + [num_classes, num_classes * grid_size ** 3 * 3]
    td, t2, ls, ld, cp, ffd = torch.split(x, sizes, dim=1)
    return {'_theta_deltas': td / torch.norm(td, p=2, dim=1, keepdim=True), '_translation2ds': t2, '_log_scales': ls,
            '_log_depths': ld, '_class_probs': F.softmax(cp, dim=1), '_ffd_coeffs': ffd.reshape(-1, num_classes, grid_size ** 3 * 3)}


class _Block(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample


class RefResNet18(nn.Module):
    """torchvision 0.2.1 `resnet18()`: same attribute names, creation order and initialisation; forward = the functional
    restatement above on this module's own state, then `self.avgpool` / `self.fc` as torchvision's forward applies them
    (the reference replaces both, derenderer.py:26-27)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, (planes, stride) in enumerate(STAGES, 1):
            down = None
            if stride != 1 or inplanes != planes:
                down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
            blocks = [_Block(inplanes, planes, stride, down), _Block(planes, planes, 1, None)]
            inplanes = planes
            setattr(self, 'layer%d' % li, nn.Sequential(*blocks))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        taps = {}
        resnet18_features(sd, x, self.training, update_running=True, taps=taps)
        for m in self.modules():   # F.batch_norm does not count batches; the modules would
            if isinstance(m, nn.BatchNorm2d) and self.training:
                m.num_batches_tracked += 1
        x = self.avgpool(taps['layer4.1'])
        return self.fc(x.view(x.size(0), -1))
