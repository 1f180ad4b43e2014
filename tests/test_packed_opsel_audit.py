"""The built library, disassembled: no kernel carries a packed fp32
instruction whose low result lane reads the HIGH half of a source pair (tools/audit_packed_opsel.py; the r06 gfx950 finding
described in csrc/conv_narrow.hip and csrc/raster_bwd.hip).  Needs no GPU: llvm-objdump reads the code objects."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_no_kernel_of_the_library_reads_a_high_half_in_packed_fp32_math():
    import audit_packed_opsel as au
    lib = os.path.join(ROOT, '3d-sdn_amd', 'lib', 'libsdn_hip.so')
    if not os.path.exists(lib) or not os.path.exists(au.OBJDUMP):
        pytest.skip('library not built or llvm-objdump missing')
    totals = {}
    counts = au.audit(lib, totals)
    assert not counts, counts
    # the audit sees the library: k_edge_rows keeps its two sums per owner in explicit two-wide math
    assert any('k_edge_rows' in k for k in totals), 'the audit found no packed fp32 math at all: is the disassembly empty?'
