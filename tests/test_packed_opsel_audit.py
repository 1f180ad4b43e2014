"""The built library, disassembled: the kernels that the product's schedules run beside MFMA kernels carry no packed fp32
instruction whose low result lane reads the HIGH half of a source pair (tools/audit_packed_opsel.py; the r06 gfx950 finding
described in csrc/conv_narrow.hip and csrc/raster_bwd.hip).  Needs no GPU: llvm-objdump reads the code objects."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_kernels_that_share_the_chip_with_mfma_kernels_are_free_of_high_half_operands():
    import audit_packed_opsel as au
    lib = os.path.join(ROOT, '3d-sdn_amd', 'lib', 'libsdn_hip.so')
    if not os.path.exists(lib) or not os.path.exists(au.OBJDUMP):
        pytest.skip('library not built or llvm-objdump missing')
    counts = au.audit(lib)
    bad = {k: n for k, n in counts.items() if any(tag in k for tag in au.MUST_BE_CLEAN)}
    assert not bad, bad
    # the audit sees the library: the exact-fp32 forward of the 3-8 channel heads is known to carry such operands (it is not on the
    # default route: the MFMA head kernel took its layers in r05 / r06) -- if this count drops to zero the disassembly found nothing
    assert any('k_conv_narrow_fwd' in k for k in counts), 'the audit found no packed fp32 math at all: is the disassembly empty?'
