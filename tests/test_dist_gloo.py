"""CPU, world_size 2 over gloo: the sharding + single all_gather of sdn_hip/dist.py (the N > 1 path of bench.py).
Gathered maps must equal the single-process result exactly, for even and uneven shards."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from sdn_hip import dist as sd  # noqa: E402


def test_shard_ranges_partition_the_items():
    for n in (0, 1, 7, 16, 640):
        for w in (1, 2, 3, 8):
            blocks = [sd.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        sd.shard_range(4, 2, 2)


def _ship(x):
    """tensors -> numpy arrays before a result crosses the queue: a torch tensor travels as a file descriptor that the
    receiver fetches from the SENDER's socket, so a worker that has exited by the time the parent reads the queue loses it
    (FileNotFoundError / ConnectionResetError in q.get -- seen once the workers got faster than the parent)"""
    if isinstance(x, torch.Tensor):
        return ('__tensor__', x.detach().cpu().numpy())
    if isinstance(x, (list, tuple)):
        return type(x)(_ship(v) for v in x)
    return x


def _land(x):
    if isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], str) and x[0] == '__tensor__':
        return torch.from_numpy(x[1])
    if isinstance(x, (list, tuple)):
        return type(x)(_land(v) for v in x)
    return x


def _fake_render(lo, hi):
    """stand-in for the per-object render: deterministic maps that depend only on the item index"""
    idx = torch.arange(lo, hi, dtype=torch.float32)
    return (idx[:, None, None, None] * 10 + torch.arange(5.0)[None, :, None, None]
            + torch.linspace(0, 1, 6 * 6).reshape(1, 1, 6, 6))


def _worker(rank, world, port, n_items, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        out = sd.render_sharded(_fake_render, n_items)
        q.put(_ship((rank, out)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('n_items', [16, 7, 1])
def test_two_ranks_gather_equals_single_process(n_items):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(_land(q.get(timeout=120)) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_render(0, n_items)
    for r in range(world):
        assert got[r].shape == want.shape
        assert torch.equal(got[r], want), 'rank %d' % r


def test_single_process_passthrough():
    maps = _fake_render(0, 5)
    assert sd.gather_maps(maps, 5) is maps
    with pytest.raises(ValueError):
        sd.gather_maps(maps, 6)


# ---- configs[4]: the frame-sharded two-stage pipeline (bench.edit_pipeline runs its stages through sd.run_frames)
def _stage_a(f):
    """stand-in for derender3d inference + compositing of frame f: maps and a record that depend on f only"""
    g = torch.Generator().manual_seed(5000 + f)
    return torch.rand(5, 6, 9, generator=g) + f, {'frame': f, 'objects': f % 3 + 1}


def _stage_b(f, maps, rec):
    """stand-in for input assembly + fake_inference: must be handed frame f's own maps and record"""
    assert rec['frame'] == f
    want, _ = _stage_a(f)
    assert torch.equal(maps, want)
    return float(maps.sum()) * rec['objects']


def _pipeline_worker(rank, world, port, n_frames, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        gathered, outs, (lo, hi) = sd.run_frames(n_frames, _stage_a, _stage_b, lambda: torch.zeros(0, 5, 6, 9))
        q.put(_ship((rank, float(gathered.double().sum()), tuple(gathered.shape), outs, (lo, hi))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [6, 5, 1])
def test_frame_pipeline_is_independent_of_the_world_size(n_frames):
    """`gathered_maps_checksum` of bench.py's configs[4] block must not depend on the number of ranks (the driver compares its
    N = 1 and N = 8 lines): even shards, uneven shards, and a rank without any frame."""
    single, outs1, _ = sd.run_frames(n_frames, _stage_a, _stage_b, lambda: torch.zeros(0, 5, 6, 9))
    assert tuple(single.shape) == (n_frames, 5, 6, 9) and len(outs1) == n_frames
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((_land(q.get(timeout=120)) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    checksum = float(single.double().sum())
    merged = []
    for rank, cs, shape, outs, (lo, hi) in got:
        assert cs == checksum and shape == (n_frames, 5, 6, 9)      # bit-identical on every rank
        assert (lo, hi) == sd.shard_range(n_frames, rank, 2) and len(outs) == hi - lo
        merged += outs
    assert merged == outs1                                           # every frame processed exactly once, in order


def test_frame_pipeline_batched_stage_b_equals_the_per_frame_one():
    """run_frames(batch=k): stage B receives groups of up to k consecutive frames of the shard and returns one output per frame"""
    single, outs1, _ = sd.run_frames(7, _stage_a, _stage_b, lambda: torch.zeros(0, 5, 6, 9))
    seen = []

    def stage_b_batch(fs, maps, recs):
        seen.append(list(fs))
        return [_stage_b(f, m, r) for f, m, r in zip(fs, maps, recs)]
    batched, outs3, _ = sd.run_frames(7, _stage_a, stage_b_batch, lambda: torch.zeros(0, 5, 6, 9), batch=3)
    assert torch.equal(batched, single) and outs3 == outs1 and seen == [[0, 1, 2], [3, 4, 5], [6]]
    with pytest.raises(ValueError):
        sd.run_frames(4, _stage_a, lambda fs, maps, recs: [0.0], lambda: torch.zeros(0, 5, 6, 9), batch=2)


# ---- the overlapped exchange (sd.MapExchange): step k's gather is in flight while step k + 1 "renders"
def _exchange_worker(rank, world, port, n_items, steps, q, mode=None):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = sd.shard_range(n_items, rank, world)
        ex = sd.MapExchange(n_items, (5, 6, 6), torch.float32, 'cpu', mode=mode)
        got, pending = [], None
        for k in range(steps):
            local = _fake_render(lo, hi) + 1000.0 * k          # step k's maps of this rank's shard
            h = ex.post(local)
            local.fill_(-1.0)                                  # the caller may reuse its buffer right after post()
            if pending is not None:
                got.append(ex.wait(pending).clone())           # step k - 1 arrives while step k was "rendered"
            pending = h
        got.append(ex.wait(pending).clone())
        q.put(_ship((rank, got, ex.bytes_sent_per_post)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_items', [16, 7])
def test_overlapped_exchange_equals_the_blocking_gather(n_items):
    world, steps = 2, 5
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_items, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((_land(q.get(timeout=120)) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs, nbytes in got:
        assert len(outs) == steps and nbytes == max(sd.shard_sizes(n_items, world)) * 5 * 6 * 6 * 4
        for k, o in enumerate(outs):
            want = _fake_render(0, n_items) + 1000.0 * k
            assert torch.equal(o, want), 'rank %d step %d' % (rank, k)   # order of steps and of items preserved, bit for bit


@pytest.mark.parametrize('world,n_items', [(2, 16), (2, 7), (3, 8), (3, 2)])
def test_direct_link_exchange_equals_the_blocking_gather(world, n_items):
    """MapExchange(mode='p2p'): W - 1 isend / irecv pairs per rank instead of the collective (the fallback for an RCCL that
    rings the all_gather) -- even shards, uneven shards, a rank with no item at all; bit-equal to gather_maps."""
    steps = 4
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_items, steps, q, 'p2p')) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((_land(q.get(timeout=120)) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = sd.shard_sizes(n_items, world)
    for rank, outs, nbytes in got:
        assert len(outs) == steps and nbytes == sizes[rank] * (world - 1) * 5 * 6 * 6 * 4   # own rows only, once per peer
        for k, o in enumerate(outs):
            assert torch.equal(o, _fake_render(0, n_items) + 1000.0 * k), 'rank %d step %d' % (rank, k)


def _guard_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = sd.shard_range(8, rank, world)
        ex = sd.MapExchange(8, (5, 6, 6), torch.float32, 'cpu', depth=2)
        h0 = ex.post(_fake_render(lo, hi))
        h1 = ex.post(_fake_render(lo, hi) + 1)
        try:
            ex.post(_fake_render(lo, hi) + 2)       # a third exchange in flight would overwrite slot 0's buffers
            third = 'accepted'
        except RuntimeError as e:
            third = str(e)
        a = ex.wait(h0).clone()
        try:
            ex.wait(h0)
            again = 'accepted'
        except RuntimeError as e:
            again = str(e)
        h2 = ex.post(_fake_render(lo, hi) + 2)      # slot 0 is free again
        b, c = ex.wait(h1).clone(), ex.wait(h2).clone()
        q.put(_ship((rank, third, again, [bool(torch.equal(t, _fake_render(0, 8) + k)) for k, t in enumerate((a, b, c))])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_exchange_refuses_to_reuse_a_slot_in_flight():
    """post() number depth + 1 before the first wait() used to overwrite a send buffer the collective may still
    be reading; now it raises, and a handle can be waited for once."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((_land(q.get(timeout=120)) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, third, again, diffs in got:
        assert 'never wait()ed' in third and 'already waited' in again
        assert diffs == [True, True, True]


def test_exchange_mode_selection(monkeypatch):
    monkeypatch.delenv('SDN_EXCHANGE', raising=False)
    assert sd.exchange_mode() == 'all_gather' and sd.exchange_mode('p2p') == 'p2p'
    monkeypatch.setenv('SDN_EXCHANGE', 'p2p')
    assert sd.exchange_mode() == 'p2p' and sd.MapExchange(3, (1,)).mode == 'p2p'
    with pytest.raises(ValueError):
        sd.exchange_mode('ring')


def test_exchange_single_process_is_the_identity():
    ex = sd.MapExchange(5, (5, 6, 6))
    maps = _fake_render(0, 5)
    assert ex.wait(ex.post(maps)) is maps
    with pytest.raises(ValueError):
        ex.post(_fake_render(0, 4))
