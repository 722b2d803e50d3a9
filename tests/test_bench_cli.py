"""bench.py's N-rank launch path (README.md:33 / train.py:96-106 of the reference start N processes): `python bench.py --gpus 2`
must become two ranks of ONE process group by itself.  No GPU here, so the ranks run the gloo rendezvous dry run
(BENCH_BACKEND=gloo): same launcher, same env contract, same census fields the real JSON line carries."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ, BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["gpus_arg"] == 2
    assert line["rccl_ranks"] == 2                                    # the size of the group the collective library formed
    assert sorted(x["rank"] for x in line["ranks"]) == [0, 1]         # gathered THROUGH the group
    assert [x["rows_M"] for x in sorted(line["ranks"], key=lambda x: x["rank"])] == [100, 101]


def test_bench_under_external_launcher():
    """The driver's form: torch.distributed.run starts the ranks, bench.py must not spawn again."""
    env = dict(os.environ, BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 2 and line["n_gpus"] == 2


def test_bench_rejects_mismatched_world():
    r = _run(["--gpus", "4"], extra_env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_refuses_more_gpus_than_present():
    """Without the dry-run backend the launcher checks the device count first and fails loudly."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("needs a node with fewer than 64 GPUs")
    r = _run(["--gpus", "64"], extra_env={"BENCH_BACKEND": "nccl"})
    assert r.returncode != 0 and "exposes" in r.stderr
