"""Lab: run pytest with a lab build of the library (tools/build_lab_variant.sh) in place of the product one.
usage: python tools/lab/pytest_with_lib.py lab/<name>.so <pytest arguments...>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd')]
import sdn_hip  # noqa: E402

sdn_hip.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
