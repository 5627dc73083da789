"""GPU: quber_amd/dist.py on a ONE-rank RCCL process group (backend "nccl" is RCCL on ROCm) - what a one-GPU box allows of
SURVEY.md 8e: the calls, dtypes and stream semantics of ProcessGroupNCCL that the gloo tests (tests/test_dist_gloo.py, world 2)
cannot show - device tensors on the wire, int16 label maps handed over as bytes (RCCL has no 16-bit integer type), the
asynchronous gather whose send buffer the caller overwrites at once.  RCCL refuses two ranks on one device, so a second rank
needs a second GPU: that part is the driver's 8-GPU run.  The group is created in this process (no child: a process that has
initialised the GPU starts no other program on the GPU boxes) and destroyed in a `finally`."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(port):
    import torch.distributed as dist
    from quber_amd import arch, dist as qdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        out = {"backend": dist.get_backend()}
        specs = arch.param_specs()
        small = {k: v for i, (k, v) in enumerate(specs.items()) if i < 40}
        full = arch.init_state_dict(seed=5)
        got = qdist.broadcast_state_dict({k: full[k] for k in small}, small, src=0, device=dev)
        out["weights"] = all(np.array_equal(got[k], full[k]) for k in small)
        # label maps as the post-processing writes them: f32 on the device, values -1, 1000 ... 1200
        g = torch.Generator().manual_seed(3)
        local = torch.randint(1000, 1201, (3, 48, 64), generator=g).float()
        local[0, :4] = -1.0
        local = local.to(dev)
        want = local.clone()
        same = qdist.gather_label_maps(local, [3], dst=0)
        out["sync"] = bool(torch.equal(same, want)) and same.device.type == "cuda"
        mine = local.clone()
        h = qdist.gather_label_maps(mine, [3], dst=0, async_op=True)
        mine.fill_(-7.0)                       # the send buffer was copied: this must not reach the destination
        out["async"] = bool(torch.equal(h.wait(), want))
        wire = qdist.label_wire_dtype()
        out["wire"] = str(wire)
        mine = local.clone()
        h = qdist.gather_label_maps(mine, [3], dst=0, async_op=True, wire_dtype=wire)
        mine.fill_(-7.0)
        w = h.wait()
        torch.cuda.synchronize()
        out["wired"] = bool(torch.equal(w, want)) and w.dtype == want.dtype
        # another batch size through the same group (the padding of a short shard needs a second rank: tests/test_dist_gloo.py)
        out["ragged"] = bool(torch.equal(qdist.gather_label_maps(local[:2], [2], dst=0, wire_dtype=wire), want[:2]))
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        objs = [None]
        dist.all_gather_object(objs, {"rank": 0})
        dist.barrier()
        out["reduce"] = float(t.item()) == 1.25 and objs == [{"rank": 0}]
        return out
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_group_broadcast_and_gather():
    if not torch.distributed.is_nccl_available():
        pytest.fail("torch.distributed has no nccl (RCCL) backend on this box")
    res = _run(_free_port())
    assert res["backend"] == "nccl" and res["wire"] == "torch.int16"
    for k in ("weights", "sync", "async", "wired", "ragged", "reduce"):
        assert res[k], (k, res)
