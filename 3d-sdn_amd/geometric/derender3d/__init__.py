"""derender3d: the geometric branch's encoder / decoder package (API of the reference's geometric/derender3d)."""
# Drop-in composition: a checkout of the reference keeps its own sibling modules of this package (datasets, data loaders,
# ...) -- with this directory placed BEFORE the reference's on sys.path, the package spans both directories and the names
# defined here win (pkgutil.extend_path).
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)


class TargetType:
    """What a run predicts and renders, as bit flags that callers combine with `|` and test with `&`
    (names and values as the reference defines them, geometric/derender3d/__init__.py:1-10: they appear in its command
    lines and checkpoints' option dumps)."""
    geometry, reproject, normal, depth = (1 << bit for bit in range(4))

    # the training stages of scripts/main.py, named by what they supervise
    pretrain, finetune = geometry, reproject
    full = geometry | reproject
    extend = full | normal | depth
