"""roialign: drop-in for geometric/maskrcnn/roialign."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # the reference's sibling modules (model.py, config.py, ...) stay importable
