import os

os.environ.pop('NEURAL_RENDERER_UNSAFE', None)   # the suite tests the default (safe) rule; the K1 tests flip the switch themselves
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric'),
          os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if p not in sys.path:
        sys.path.insert(0, p)


# the tests run the networks with seeded random weights: ImageNet files cannot be downloaded here (see
# textural/models/networks.py load_pretrained, geometric/derender3d/models/resnet.py resnet18)
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
