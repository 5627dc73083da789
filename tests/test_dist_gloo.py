"""CPU, world_size 2 over gloo: the N>1 plumbing (shard ranges, weight broadcast, label-map gather)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quber_amd import arch, dist as qdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        specs = arch.param_specs()
        small = {k: v for i, (k, v) in enumerate(specs.items()) if i < 40}     # a slice keeps the test fast
        sd = None
        if rank == 0:
            full = arch.init_state_dict(seed=5)
            sd = {k: full[k] for k in small}
        got = qdist.broadcast_state_dict(sd, small, src=0)
        ref = arch.init_state_dict(seed=5)
        ok_w = all(np.array_equal(got[k], ref[k]) for k in small)
        s, e = qdist.shard_range(n_frames, rank, world)
        # every rank "refines" its shard: label map b is filled with the global frame index
        local = torch.stack([torch.full((4, 6), float(i)) for i in range(s, e)]) if e > s else torch.zeros((0, 4, 6))
        counts = [qdist.shard_range(n_frames, r, world)[1] - qdist.shard_range(n_frames, r, world)[0] for r in range(world)]
        allmaps = qdist.gather_label_maps(local, counts, dst=0)
        # the asynchronous form bench.py uses (the gather of step i travels while step i + 1 computes): the send buffer is
        # copied, so overwriting `local` right after the call must not change what arrives
        mine = local.clone()
        handle = qdist.gather_label_maps(mine, counts, dst=0, async_op=True)
        mine.fill_(-7.0)
        later = handle.wait()
        # int16 on the wire (bench.py: label values are integers <= 1200), f32 again on the destination
        wired = qdist.gather_label_maps(local, counts, dst=0, async_op=True, wire_dtype=qdist.label_wire_dtype()).wait()
        if rank == 0:
            ok_g = allmaps.shape == (n_frames, 4, 6) and all(float(allmaps[i, 0, 0]) == i for i in range(n_frames))
            ok_g = ok_g and torch.equal(later, allmaps) and torch.equal(wired, allmaps) and wired.dtype == allmaps.dtype
        else:
            ok_g = allmaps is None and later is None
        q.put((rank, ok_w, ok_g, (s, e)))
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 16, 129):
        for w in (1, 2, 3, 8):
            rs = [qdist.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1:3] == (True, True) and res[1][1:3] == (True, True)
    assert res[0][3] == (0, 3) and res[1][3] == (3, 5)
