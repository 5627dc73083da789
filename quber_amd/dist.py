"""Data-parallel sharding of frames over the GPUs of one node (one process per GPU, torch.distributed).

Frames are independent, so the data path has no exchange step: rank r refines its own shard.
The only collectives are (1) a broadcast of the flat weight vector from the rank that owns the checkpoint and
(2) a gather of the refined label maps (and per-frame counts) to rank 0.  Backend "nccl" is RCCL over xGMI on
ROCm; the same code runs on "gloo" for the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [start, stop) of `n_items` for `rank` (first n%world ranks get one extra)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def _comm_device(device):
    """gloo (CPU rehearsals of the multi-rank path) moves host tensors; nccl = RCCL moves device tensors."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)


def broadcast_state_dict(sd, specs, src=0, device="cpu"):
    """sd: name -> numpy array on `src` (ignored elsewhere); specs: ordered name -> (shape, kind).  Returns the
    full state_dict on every rank after ONE broadcast of the concatenated fp32 vector."""
    total = int(sum(int(np.prod(s)) for s, _ in specs.values()))
    device = _comm_device(device)
    if dist.get_rank() == src:
        flat = torch.from_numpy(np.concatenate([np.asarray(sd[k], np.float32).ravel() for k in specs])).to(device)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out, o = {}, 0
    for k, (shape, _) in specs.items():
        n = int(np.prod(shape))
        out[k] = host[o:o + n].reshape(shape)
        o += n
    return out


def label_wire_dtype(label_divisor=1000, n_thing_classes=1, top_k=200):
    """Smallest integer type that carries every value a panoptic map can hold (-1, class * divisor [+ instance id <= top_k]):
    int16 for the refiner's one thing class (labels <= 1200) - half the bytes of the f32 maps on the wire."""
    return torch.int16 if (n_thing_classes + 1) * label_divisor + top_k < 32768 else torch.int32


def gather_label_maps(local, counts, dst=0, async_op=False, wire_dtype=None):
    """local: [b_r, H, W] label maps of this rank (b_r may differ by one between ranks); counts: frames per rank.
    Returns the concatenated [sum(b_r), H, W] tensor on `dst`, None elsewhere.
    wire_dtype (e.g. label_wire_dtype()): the maps travel in that integer type (the values are integers: lossless) and come back in
    `local`'s own dtype on `dst`.
    async_op=True: returns a handle instead, `h.wait()` -> that result.  The collective then runs on the backend's
    own stream beside whatever the caller enqueues next (the next step's kernels); `local` is copied first, so the
    caller may overwrite it at once."""
    world, rank = dist.get_world_size(), dist.get_rank()
    bmax = max(counts)
    out_dtype = local.dtype
    wired = wire_dtype is not None and wire_dtype != local.dtype
    if wired:
        # a fresh tensor (no further copy needed for the asynchronous form), handed to the backend as bytes: neither gloo nor
        # RCCL has a 16-bit integer type, and a gather only moves bytes
        local = local.to(wire_dtype).contiguous().view(torch.uint8)
        async_copy = False
    else:
        async_copy = async_op
    if local.shape[0] < bmax:
        pad = torch.cat([local, local.new_full((bmax - local.shape[0],) + tuple(local.shape[1:]), 255 if wired else -1)])
    else:
        pad = local.clone() if async_copy else local
    pad = pad.contiguous().to(_comm_device(pad.device))
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    work = dist.gather(pad, bufs, dst=dst, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        got = torch.cat([b[:c] for b, c in zip(bufs, counts)])
        return got.view(wire_dtype).to(out_dtype) if wired else got

    if not async_op:
        return finish()
    return _PendingGather(work, finish, pad)


class _PendingGather:
    def __init__(self, work, finish, keep):
        self.work, self.finish, self.keep = work, finish, keep      # `keep`: the send buffer stays alive until the wait

    def wait(self):
        self.work.wait()
        self.keep = None
        return self.finish()
