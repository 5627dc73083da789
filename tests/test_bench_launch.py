"""bench.py's launch contract: `--gpus N` starts its own N ranks and never reports a rank count it did
not run.  CPU only: --dry exercises rendezvous, the weight broadcast, the label-map gather and the JSON line over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def test_self_launch_two_ranks_dry():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "2", "--batch", "3"],
                       env=_env(QUBER_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry"] is True and j["scaling"] == "weak"
    assert [x["rank"] for x in j["rccl_ranks"]] == [0, 1]
    assert all(x["world_size"] == 2 for x in j["rccl_ranks"])
    assert len({x["weights"] for x in j["rccl_ranks"]}) == 1        # every rank holds rank 0's weights after the broadcast
    # the N > 1 line explains itself: per-rank step medians, what the (overlapped) gather cost the loop, what it costs alone
    for x in j["rccl_ranks"]:
        for k in ("step_ms_median", "step_ms_max", "elapsed_ms_per_step", "gather_ms_per_step", "gather_alone_ms", "instances_out",
                  "frames", "backend", "device"):
            assert k in x, k
        assert x["backend"] == "gloo" and x["frames"] == 3 and x["gather_alone_ms"] > 0
    g = j["gather"]
    assert g["bytes_per_rank_per_step"] == 3 * 32 * 48 * 2        # int16 label maps on the wire
    assert g["ms_per_step_max_over_ranks"] == max(x["gather_ms_per_step"] for x in j["rccl_ranks"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k


def test_self_launch_eight_ranks_dry():
    """The N = 8 line of the driver's scaling run, rehearsed on the CPU: eight ranks rendezvous over gloo, receive rank 0's weights,
    gather their (asynchronous) label maps in rank order, and rank 0 prints ONE line that carries all eight."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry", "--steps", "2", "--batch", "2"],
                       env=_env(QUBER_DIST_BACKEND="gloo", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["dry"] is True and j["scaling"] == "weak"
    assert [x["rank"] for x in j["rccl_ranks"]] == list(range(8)) and all(x["world_size"] == 8 for x in j["rccl_ranks"])
    assert len({x["weights"] for x in j["rccl_ranks"]}) == 1
    assert j["gather"]["bytes_per_rank_per_step"] == 2 * 32 * 48 * 2          # int16 label maps on the wire


def test_nccl_preflight_needs_enough_gpus():
    # one rank of a 2-rank RCCL launch on a node that shows fewer GPUs than --gpus: a clear error before any process group
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "visible GPUs" in (r.stderr + r.stdout)


def test_world_size_mismatch_is_an_error():
    # a torchrun-style environment with one rank must not silently satisfy --gpus 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_single_rank_dry_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["rccl_ranks"][0]["world_size"] == 1
