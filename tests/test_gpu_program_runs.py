"""sdn_program_run gathers runs of pack / unpack / small-copy records into ONE k_weights_multi launch whose tensors are
processed concurrently (csrc/fast_program.hip).  A record that overlaps -- by BYTE RANGE, not only by base pointer -- what an
earlier record of the run writes (or writes what an earlier one reads) must end the run, so that program order is kept
(ADVICE r04).  The records here are built through the public Builder, as any sdn_program client would."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(prog, ext):
    st = torch.cuda.current_stream().cuda_stream
    prog.run({}, ext, st)
    torch.cuda.synchronize()


@pytest.mark.parametrize('n', [1024, 16384, 1 << 20])   # copies up to 64 KB join a run; 4 MB is a plain memcpy record
def test_copy_chain_through_overlapping_sub_ranges_keeps_program_order(n):
    from sdn_hip import program as pg
    x = torch.arange(n, dtype=torch.float32, device=DEV)
    y = torch.full((2 * n,), -1.0, device=DEV)
    z = torch.full((n,), -2.0, device=DEV)
    w = torch.full((n,), -3.0, device=DEV)
    b = pg.Builder()
    # y[n/2 : 3n/2] <- x;  z <- y[n : 2n] (second half of it written by the first record);  w <- z;  x_copy untouched
    b.op(pg.OP_COPY, buf=[b.ext('y_mid'), b.ext('x')], l=[4 * n])
    b.op(pg.OP_COPY, buf=[b.ext('z'), b.ext('y_hi')], l=[4 * n])
    b.op(pg.OP_COPY, buf=[b.ext('w'), b.ext('z')], l=[4 * n])
    prog = b.finish()
    for _ in range(3):
        y.fill_(-1.0)
        z.fill_(-2.0)
        w.fill_(-3.0)
        _run(prog, {'x': x, 'y_mid': y.data_ptr() + 4 * (n // 2), 'y_hi': y.data_ptr() + 4 * n, 'z': z, 'w': w})
        want_z = torch.cat([x[n // 2:], torch.full((n // 2,), -1.0, device=DEV)])
        assert torch.equal(z, want_z)
        assert torch.equal(w, want_z)


def test_write_after_read_of_a_sub_range_keeps_program_order():
    from sdn_hip import program as pg
    n = 8192
    a = torch.arange(2 * n, dtype=torch.float32, device=DEV)
    keep = a.clone()
    out = torch.zeros(n, device=DEV)
    zeros = torch.zeros(n, device=DEV)
    b = pg.Builder()
    b.op(pg.OP_COPY, buf=[b.ext('out'), b.ext('a_mid')], l=[4 * n])      # reads a[n/2 : 3n/2]
    b.op(pg.OP_COPY, buf=[b.ext('a_lo'), b.ext('zeros')], l=[4 * n])     # then overwrites a[0 : n] (overlaps what was read)
    prog = b.finish()
    _run(prog, {'out': out, 'a_mid': a.data_ptr() + 4 * (n // 2), 'a_lo': a, 'zeros': zeros})
    assert torch.equal(out, keep[n // 2:n // 2 + n])
    assert float(a[:n].abs().max()) == 0.0 and torch.equal(a[n:], keep[n:])
