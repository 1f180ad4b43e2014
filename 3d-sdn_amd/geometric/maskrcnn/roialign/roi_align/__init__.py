"""roi_align: CropAndResizeFunction / CropAndResize / RoIAlign on HIP kernels (reference: roialign/roi_align)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # the reference's sibling modules (model.py, config.py, ...) stay importable
