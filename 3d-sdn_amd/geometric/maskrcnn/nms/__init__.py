"""nms: drop-in for geometric/maskrcnn/nms (nms_wrapper.nms, pth_nms.pth_nms) without the compiled `_ext` module."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # the reference's sibling modules (model.py, config.py, ...) stay importable
