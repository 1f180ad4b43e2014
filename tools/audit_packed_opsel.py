"""Post-build audit: packed fp32 instructions whose LOW result lane reads the HIGH half of a source pair.

r06 finding (csrc/conv_narrow.hip, csrc/raster_bwd.hip; tools/lab/head_race*.py, tools/lab/pk_race.py): on gfx950 a `v_pk_fma_f32` /
`v_pk_mul_f32` with such an operand (`op_sel` bit set for a source: inline asm `op_sel:[0,1,0]`, or hipcc's own `op_sel:[1,0]` for a
value that an LDS read delivered in the high half of a pair) returned wrong results when a wave of an MFMA kernel shared the SIMD
-- exact when alone on the chip.  hipcc's SLP vectoriser emits these forms freely from scalar code.  The library is kept FREE of
them: everything but conv_*.hip is built with -fno-slp-vectorize (csrc/Makefile), the fp32 kernels of conv_narrow.hip make the
splat values opaque, the explicit two-wide math of k_edge_rows likewise.  This script is the check:

    python tools/audit_packed_opsel.py [libsdn_hip.so]      -> per-kernel counts; exit 1 if any kernel has one

(tests/test_packed_opsel_audit.py runs it on the built library.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PK = re.compile(r'\bv_pk_(mul|add|fma)_f32\b')
SEL = re.compile(r'op_sel:\[([01,]+)\]')


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        copy = os.path.join(d, 'lib.so')
        with open(lib, 'rb') as f, open(copy, 'wb') as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, '--offloading', copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        out = []
        for name in sorted(os.listdir(d)):
            if 'amdgcn' in name:
                out.append(subprocess.run([OBJDUMP, '-d', os.path.join(d, name)], check=True, capture_output=True, text=True).stdout)
        return '\n'.join(out)


def audit(lib, totals=None):
    """-> {kernel: flagged instructions}; totals (optional dict): kernel -> all packed fp32 instructions seen"""
    counts, kernel = {}, None
    for line in disassemble(lib).splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if m:
            kernel = m.group(1)
            continue
        if kernel and PK.search(line):
            if totals is not None:
                totals[kernel] = totals.get(kernel, 0) + 1
            s = SEL.search(line)
            if s and '1' in s.group(1):
                counts[kernel] = counts.get(kernel, 0) + 1
    return counts


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, '3d-sdn_amd', 'lib', 'libsdn_hip.so')
    totals = {}
    counts = audit(lib, totals)
    for k, n in sorted(counts.items(), key=lambda kv: -kv[1]):
        print('%5d  %s' % (n, k))
    print('%d kernels use packed fp32 math (%d instructions); %d of them carry high-half operands' % (
        len(totals), sum(totals.values()), len(counts)))
    return 1 if counts else 0


if __name__ == '__main__':
    sys.exit(main())
