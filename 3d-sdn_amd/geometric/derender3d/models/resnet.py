"""ResNet-18 with torchvision-compatible parameter names (conv1, bn1, layer{1..4}.{0,1}.{conv,bn}{1,2},
layer{2..4}.0.downsample.{0,1}, fc), so checkpoints written by the reference
(`derenderer.net.*`, geometric/derender3d/models/derenderer.py:25-27) load unchanged.  torchvision is not
installed in this image, and its pretrained weights need a download, so this is a local definition; pretrained
weights come from a file (see resnet18).

The modules are parameter containers; forward() runs on the HIP kernels (sdn_hip/bnnet.py: MFMA implicit-GEMM
convolutions, fused BatchNorm + residual + ReLU passes, max / average pooling), forward and backward, train and eval
mode.  CPU tensors raise NotImplementedError -- there is no torch.nn fallback."""
import torch
import torch.nn as nn

from sdn_hip import bnnet as hb


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        # relu(bn2(conv2(relu(bn1(conv1(x))))) + identity): every conv an MFMA implicit GEMM, every bn (+ add + relu) one pass
        identity = x if self.downsample is None else hb.batch_norm(self.downsample[1], hb.conv2d(self.downsample[0], x))
        out = hb.batch_norm(self.bn1, hb.conv2d(self.conv1, x), relu=True)
        return hb.batch_norm(self.bn2, hb.conv2d(self.conv2, out), res=identity, relu=True)


class ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super(ResNet18, self).__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2, 1)
        self.layer2 = self._make_layer(128, 2, 2)
        self.layer3 = self._make_layer(256, 2, 2)
        self.layer4 = self._make_layer(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        """torchvision's ResNet.forward on the HIP kernels (sdn_hip/bnnet.py); `avgpool` is the AdaptiveAvgPool2d(1) the
        Derenderer patches in (derenderer.py:26), `fc` a plain library GEMM."""
        if not isinstance(self.avgpool, nn.AdaptiveAvgPool2d) or self.avgpool.output_size not in (1, (1, 1)):
            raise NotImplementedError('ResNet18 on the HIP path pools to 1x1 (derenderer.py:26)')
        x = hb.max_pool_3x3_s2(hb.batch_norm(self.bn1, hb.conv2d(self.conv1, x), relu=True))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for block in layer:
                x = block(x)
        return self.fc(hb.global_avg_pool(x))


def resnet18(pretrained=False):
    """torchvision.models.resnet18(pretrained) (derenderer.py:25).  pretrained=True loads the torchvision-format
    state_dict named by SDN_RESNET18_WEIGHTS; without that file it raises unless SDN_ALLOW_RANDOM_INIT=1 (benchmarks and
    tests) -- never a silent random encoder."""
    net = ResNet18()
    if pretrained:
        import os
        import warnings
        path = os.environ.get('SDN_RESNET18_WEIGHTS')
        if path:
            net.load_state_dict(torch.load(path, map_location='cpu'), strict=True)
        elif os.environ.get('SDN_ALLOW_RANDOM_INIT') == '1':
            warnings.warn('resnet18(pretrained=True): SDN_RESNET18_WEIGHTS is not set; RANDOM initialisation '
                          '(SDN_ALLOW_RANDOM_INIT=1)', RuntimeWarning)
        else:
            raise RuntimeError('resnet18(pretrained=True) needs the ImageNet weights: set SDN_RESNET18_WEIGHTS to a '
                               'torchvision-format state_dict file, or SDN_ALLOW_RANDOM_INIT=1 (benchmarks only)')
    return net
