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


def test_bench_functions_use_only_names_they_can_see():
    """bench.py's finetune / recognize paths only run on a GPU box: a name that exists in one `main_*` function and was pasted into
    another (it happened: `graphed` in main_finetune) must fail HERE.  Every name a function loads is a local, an enclosing or
    module-level name, or a builtin."""
    import ast
    import builtins
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    module_names = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            module_names.add(node.name)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            module_names.update((a.asname or a.name).split(".")[0] for a in node.names)
        elif isinstance(node, (ast.Assign, ast.AnnAssign, ast.AugAssign)):
            for t in ast.walk(node):
                if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Store):
                    module_names.add(t.id)

    def bound_in(fn):
        names = {a.arg for a in fn.args.args + fn.args.kwonlyargs}
        if fn.args.vararg:
            names.add(fn.args.vararg.arg)
        if fn.args.kwarg:
            names.add(fn.args.kwarg.arg)
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                names.add(n.id)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                names.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n is not fn:
                names.add(n.name)
                names.update(a.arg for a in getattr(n, "args", ast.arguments(args=[], kwonlyargs=[], posonlyargs=[], defaults=[], kw_defaults=[])).args)
            elif isinstance(n, ast.Lambda):
                names.update(a.arg for a in n.args.args)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                names.add(n.name)
            elif isinstance(n, ast.comprehension):
                names.update(t.id for t in ast.walk(n.target) if isinstance(t, ast.Name))
        return names

    problems = []
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        visible = bound_in(fn) | module_names
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in visible:
                problems.append((fn.name, n.id, n.lineno))
    assert not problems, problems
