"""maskrcnn: the two custom ops of the reference's Mask R-CNN (geometric/maskrcnn/nms, roialign) on HIP kernels.
The network itself (model.py) is the reference's torch code and out of scope."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # the reference's sibling modules (model.py, config.py, ...) stay importable
