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
