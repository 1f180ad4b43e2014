"""textural.models: the part of the reference's textural/models package that runs on the HIP kernels."""
# Drop-in composition: a checkout of the reference keeps its own sibling modules of this package (datasets, data loaders,
# ...) -- with this directory placed BEFORE the reference's on sys.path, the package spans both directories and the names
# defined here win (pkgutil.extend_path).
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
