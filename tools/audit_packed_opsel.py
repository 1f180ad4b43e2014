"""Post-build audit: packed fp32 instructions whose LOW result lane reads the HIGH half of a source pair.

r06 finding (csrc/conv_narrow.hip, csrc/raster_bwd.hip; tools/lab/head_race*.py, tools/lab/pk_race.py): on gfx950 a `v_pk_fma_f32` /
`v_pk_mul_f32` with such an operand (`op_sel` bit set for a source: inline asm `op_sel:[0,1,0]`, or hipcc's own `op_sel:[1,0]` for a
value that an LDS read delivered in the high half of a pair) returned wrong results when a wave of an MFMA kernel shared the SIMD
-- exact when alone on the chip.  hipcc emits these forms freely (SLP-vectorised scalar code), and most of them have never been seen
to fail; the kernels that the product's schedules run BESIDE MFMA kernels (the train step's side streams) are kept free of them, and
this script is the check:

    python tools/audit_packed_opsel.py [libsdn_hip.so]      -> per-kernel counts; exit 1 if a kernel of MUST_BE_CLEAN has any

(tests/test_packed_opsel_audit.py runs it on the built library.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
# kernels that run on a side stream beside the MFMA conv kernels in the product's default schedules (textural train step), or that
# were hit once already
MUST_BE_CLEAN = ('k_edge_rows', 'k_wgrad_narrow_row', 'k_wgrad_narrowILi1E', 'k_conv_narrow_fwdILi1E', 'k_in_', 'k_act_bwd',
                 'k_weights_multi', 'k_split_planes', 'k_reflect_fold', 'k_l1_', 'k_segment_', 'k_avgpool3s2', 'k_assemble_nhwc',
                 'k_wgrad_head_mfma', 'k_conv_head_mfma')
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


def audit(lib):
    counts, kernel = {}, None
    for line in disassemble(lib).splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if m:
            kernel = m.group(1)
            continue
        if kernel and PK.search(line):
            s = SEL.search(line)
            if s and '1' in s.group(1):
                counts[kernel] = counts.get(kernel, 0) + 1
    return counts


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, '3d-sdn_amd', 'lib', 'libsdn_hip.so')
    counts = audit(lib)
    bad = {k: n for k, n in counts.items() if any(tag in k for tag in MUST_BE_CLEAN)}
    for k, n in sorted(counts.items(), key=lambda kv: -kv[1]):
        print('%5d  %s%s' % (n, k, '   <-- must be clean' if k in bad else ''))
    print('%d kernels carry high-half operands in packed fp32 math; %d of them on the must-be-clean list' % (len(counts), len(bad)))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
