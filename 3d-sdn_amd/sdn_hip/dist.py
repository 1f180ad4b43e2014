"""Multi-GPU layer of the hot path: one process per GPU, objects / frames sharded by rank, ONE collective.

The reference has no distributed code at all (single-process nn.DataParallel only: geometric/scripts/main.py:182,631,
textural/models/models.py:16-17).  Its unit of work -- one object's render (derender3d/models/__init__.py:161 has no
cross-object dependence) or one frame of the textural branch -- is independent, so the MI355X design is: rank r takes
the contiguous block [r*n/W, (r+1)*n/W) of the items, renders it with the local HIP kernels, and the rendered maps
([n_r, 5, R, R] fp32: mask, normal xyz, depth) are exchanged with a single all_gather over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  Shards may be uneven: they are padded to the
largest shard for the collective and trimmed afterwards.  No other data-path communication exists.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced block of [0, n_items) for `rank`: sizes differ by at most one, earlier ranks get the extra."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('rank %d of %d' % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_maps(local_maps, n_items, group=None):
    """all_gather of per-item maps.  local_maps: [n_local, ...] holding this rank's shard_range(n_items, rank, world)
    block.  Returns [n_items, ...] on every rank, in global item order (bit-identical to a single-rank run, since items
    are independent)."""
    if not dist.is_available() or not dist.is_initialized():
        if local_maps.shape[0] != n_items:
            raise ValueError('single process holds %d of %d items' % (local_maps.shape[0], n_items))
        return local_maps
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_items, world)
    if local_maps.shape[0] != sizes[rank]:
        raise ValueError('rank %d holds %d items, its shard has %d' % (rank, local_maps.shape[0], sizes[rank]))
    biggest = max(sizes)
    local = local_maps.contiguous()
    if sizes[rank] != biggest:
        pad = torch.zeros((biggest - sizes[rank],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    if all(s == biggest for s in sizes):
        return out
    return torch.cat([out[r * biggest:r * biggest + sizes[r]] for r in range(world)], 0)


EXCHANGE_MODES = ('all_gather', 'p2p')


def exchange_mode(mode=None):
    """'all_gather' (one RCCL collective; the library picks ring / direct) or 'p2p' (W - 1 isend / irecv pairs per rank: every
    shard crosses exactly one xGMI link, no ring).  None -> $SDN_EXCHANGE or 'all_gather'."""
    import os
    mode = mode or os.environ.get('SDN_EXCHANGE') or 'all_gather'
    if mode not in EXCHANGE_MODES:
        raise ValueError('exchange mode %r not in %r' % (mode, EXCHANGE_MODES))
    return mode


class MapExchange:
    """The same all_gather, OVERLAPPED with the next step's rendering (r04), with a direct-link fallback (r05).

    At the measured rate (one 16-object frame per 0.9 ms per GPU) the blocking gather_maps is the scaling limiter: 47.2 MB
    leave every rank per step, each GPU receives 7 x 47.2 = 330 MB -- 0.31 ms over its seven 153 GB/s xGMI links at best,
    about 2.2 ms through a ring.  Nothing consumes the gathered maps before the NEXT step's maps exist (the reference
    composites / edits frames after the whole batch, geometric/scripts/main.py:541-607), so the exchange of step k runs while
    step k + 1 renders:

        h = ex.post(maps_k)        # copy into a persistent send slot, enqueue the exchange (async: RCCL runs it on the
                                   # process group's own stream, ordered behind the copy)
        ...render step k + 1...
        all_maps_k = ex.wait(h)    # the CURRENT stream waits for the exchange (no host block with RCCL); a view of the
                                   # receive slot, valid until `depth` further post() calls

    mode 'all_gather': one `all_gather_into_tensor` (uneven shards padded to the largest, as in gather_maps).
    mode 'p2p' (selectable with SDN_EXCHANGE=p2p): W - 1 `isend` / `irecv` pairs per rank, batched
    (`batch_isend_irecv` = one RCCL group): rank r sends its shard to r + k and receives r - k's for k = 1..W-1, each straight
    into its place of the receive slot -- xGMI is point to point, so every shard crosses exactly ONE link and the seven
    links of a GPU carry the seven transfers at once (no ring, no padding: uneven shards send their own row count).
    Both produce tensors bit-identical to gather_maps'.

    `depth` send / receive slots (default 2) make the buffers of an exchange in flight immune to the next post(); a slot is
    only reused once its previous exchange has been wait()ed -- post() raises RuntimeError otherwise (a third
    post before the first wait would overwrite a send buffer RCCL may still be reading).  Without an initialised process
    group post / wait degrade to the identity (single process)."""

    def __init__(self, n_items, item_shape, dtype=torch.float32, device=None, group=None, depth=2, mode=None):
        self.n_items, self.group, self.depth = int(n_items), group, int(depth)
        if self.depth < 1:
            raise ValueError('depth %d' % self.depth)
        self.mode = exchange_mode(mode)
        self.active = dist.is_available() and dist.is_initialized()
        self._k = 0
        self._pending = [None] * self.depth     # per slot: the handle of an exchange that has not been wait()ed
        if not self.active:
            self.world, self.rank, self.sizes = 1, 0, [self.n_items]
            return
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.sizes = shard_sizes(self.n_items, self.world)
        self.biggest = max(self.sizes)
        self.even = all(s == self.biggest for s in self.sizes)
        tail = tuple(int(v) for v in item_shape)
        self._send = [torch.zeros((self.biggest,) + tail, dtype=dtype, device=device) for _ in range(self.depth)]
        self._recv = [torch.empty((self.world * self.biggest,) + tail, dtype=dtype, device=device) for _ in range(self.depth)]
        item_bytes = self._send[0][0].numel() * self._send[0].element_size() if self.biggest else 0
        if self.mode == 'p2p':
            self.bytes_sent_per_post = item_bytes * self.sizes[self.rank] * (self.world - 1)
        else:
            self.bytes_sent_per_post = item_bytes * self.biggest

    def _peer(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def post(self, local_maps):
        if not self.active:
            if local_maps.shape[0] != self.n_items:
                raise ValueError('single process holds %d of %d items' % (local_maps.shape[0], self.n_items))
            return (None, local_maps)
        n = self.sizes[self.rank]
        if local_maps.shape[0] != n:
            raise ValueError('rank %d holds %d items, its shard has %d' % (self.rank, local_maps.shape[0], n))
        slot = self._k % self.depth
        if self._pending[slot] is not None:
            raise RuntimeError('MapExchange: post() number %d would reuse slot %d whose exchange was never wait()ed -- at most '
                               '%d exchanges may be in flight (raise `depth` or wait() first)' % (self._k, slot, self.depth))
        self._k += 1
        send, recv, B = self._send[slot], self._recv[slot], self.biggest
        send[:n].copy_(local_maps)          # (rows behind n stay zero: the padding of an uneven shard)
        if self.mode == 'p2p':
            recv[self.rank * B:self.rank * B + n].copy_(send[:n])
            ops = []
            for k in range(1, self.world):
                dst, src = (self.rank + k) % self.world, (self.rank - k) % self.world
                if n:
                    ops.append(dist.P2POp(dist.isend, send[:n], self._peer(dst), self.group))
                if self.sizes[src]:
                    ops.append(dist.P2POp(dist.irecv, recv[src * B:src * B + self.sizes[src]], self._peer(src), self.group))
            work = dist.batch_isend_irecv(ops) if ops else []
        else:
            work = [dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)]
        handle = (work, slot)
        self._pending[slot] = handle
        return handle

    def wait(self, handle):
        work, slot = handle
        if work is None:
            return slot
        if self._pending[slot] is not handle:
            raise RuntimeError('MapExchange: this exchange was already waited for (its slot has been reused)')
        for w in work:
            w.wait()
        self._pending[slot] = None
        out = self._recv[slot]
        if self.even:
            return out
        return torch.cat([out[r * self.biggest:r * self.biggest + self.sizes[r]] for r in range(self.world)], 0)


def render_sharded(render_fn, n_items, group=None):
    """render_fn(lo, hi) -> [hi - lo, ...] maps of items lo..hi-1 on this rank's GPU; returns all n_items maps."""
    if dist.is_available() and dist.is_initialized():
        lo, hi = shard_range(n_items, dist.get_rank(group), dist.get_world_size(group))
    else:
        lo, hi = 0, n_items
    return gather_maps(render_fn(lo, hi), n_items, group)


def run_frames(n_frames, stage_a, stage_b, empty, group=None, batch=1, stage_a_all=None):
    """The frame-sharded two-stage pipeline of configs[4] (geometric/scripts/main.py:375-622 -> textural/edit_vkitti.py:105):
        stage_a(f) -> (maps [C, H, W], record)   for the frames f of THIS rank's shard (rendering + compositing),
        ONE all_gather of the ranks' maps [f_r, C, H, W]  (the path's only exchange),
        stage_b(f, maps_f, record) -> output      again for this rank's frames, on the gathered tensor.
    batch > 1: stage B runs on groups of up to `batch` consecutive frames of the shard -- stage_b(frames, [maps_f], [records])
    -> one output per frame (frames are independent, textural/edit_vkitti.py:105 loops over them one by one; batching them
    only changes how full the GPU is).
    stage_a_all (optional, replaces stage_a): called ONCE with the list of this rank's frames -> [(maps, record), ...] in that
    order, so that a rank can issue the device work of all its frames before it reads anything back (one host
    synchronisation per rank instead of several per frame).
    `empty` builds a [0, C, H, W] tensor for a rank without frames.  Returns (gathered [n_frames, C, H, W], outputs of this
    rank's frames, (lo, hi)).  Frame f's results depend on f only, so `gathered` is the same for every world size."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(n_frames, rank, world)
    local, records = [], []
    if stage_a_all is not None:
        got = list(stage_a_all(list(range(lo, hi))))
        if len(got) != hi - lo:
            raise ValueError('stage_a_all returned %d results for %d frames' % (len(got), hi - lo))
        for m, rec in got:
            local.append(m)
            records.append(rec)
    else:
        for f in range(lo, hi):
            m, rec = stage_a(f)
            local.append(m)
            records.append(rec)
    local = torch.stack(local) if local else empty()
    gathered = gather_maps(local, n_frames, group) if world > 1 else local
    if batch <= 1:
        outs = [stage_b(f, gathered[f], records[f - lo]) for f in range(lo, hi)]
    else:
        outs = []
        for g0 in range(lo, hi, batch):
            fs = list(range(g0, min(hi, g0 + batch)))
            got = list(stage_b(fs, [gathered[f] for f in fs], [records[f - lo] for f in fs]))
            if len(got) != len(fs):
                raise ValueError('stage_b returned %d outputs for %d frames' % (len(got), len(fs)))
            outs += got
    return gathered, outs, (lo, hi)
