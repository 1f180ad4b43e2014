"""Derenderer: ResNet-18 image encoder + a 3-layer MLP whose output row is cut into the per-object quantities the
decoder needs (pose delta, 2-D offset, log scale, log depth, class distribution, one FFD coefficient set per class).

Reference behaviour: geometric/derender3d/models/derenderer.py:7-65.  The attribute names `net`, `fc1`, `fc2`, `_fc3`
and the order in which the layers are created are kept, so reference checkpoints load (`state_dict` keys unchanged)
and a seeded initialisation draws the same numbers.  torchvision is not a dependency: `resnet.py` holds a
key-compatible ResNet-18.
"""
import torch
import torch.nn as nn

from .resnet import resnet18

# (name, width) of the fixed-width leading columns of the output row; the class and FFD blocks follow
_FIXED_COLUMNS = (('_theta_deltas', 2), ('_translation2ds', 2), ('_log_scales', 3), ('_log_depths', 1))


class Derenderer(nn.Module):
    in_size = 4        # two normalised rois (mask roi, detection roi) of 2 numbers each
    hidden_size = 256

    def __init__(self, num_classes=8, grid_size=4):
        super().__init__()
        self.num_classes, self.grid_size = num_classes, grid_size
        coeffs_per_class = 3 * grid_size ** 3
        self.out_sizes = dict(_FIXED_COLUMNS)
        self.out_sizes['_class_probs'] = num_classes
        self.out_sizes['_ffd_coeffs'] = num_classes * coeffs_per_class

        hidden = Derenderer.hidden_size
        backbone = resnet18(pretrained=True)
        backbone.avgpool = nn.AdaptiveAvgPool2d(1)       # any crop size, not only 224 x 224
        backbone.fc = nn.Linear(512, hidden)
        self.net = backbone
        self.relu = nn.ReLU(inplace=True)
        widths = (hidden + Derenderer.in_size, hidden, hidden, sum(self.out_sizes.values()))
        self.fc1, self.fc2, self._fc3 = (nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, images, mroi_norms, droi_norms):
        h = torch.cat((self.relu(self.net(images)), mroi_norms, droi_norms), dim=1)
        for layer in (self.fc1, self.fc2):
            h = self.relu(layer(h))
        row = self._fc3(h)

        blob, start = {}, 0
        for name, width in self.out_sizes.items():
            blob[name] = row[:, start:start + width]
            start += width
        # unit (cos, sin) pose delta; class distribution; one coefficient set per class
        delta = blob['_theta_deltas']
        blob['_theta_deltas'] = delta / torch.norm(delta, p=2, dim=1, keepdim=True)
        blob['_class_probs'] = torch.softmax(blob['_class_probs'], dim=1)
        blob['_ffd_coeffs'] = blob['_ffd_coeffs'].reshape(row.shape[0], self.num_classes, -1)
        return blob
