"""Runtime helpers of the pretraining loop, behind the reference's `Dino.modules.utils` names
(Dino/modules/utils.py: clip_gradients 132-141, cancel_gradients_last_layer 144-149, restart_from_checkpoint 152-184,
cosine_iter_scheduler 200-210, fix_random_seeds 226-232, SmoothedValue/MetricLogger 235-411, dist helpers 434-510,
trunc_normal_ 523-561, get_params_groups 643-654, has_batchnorms 657-662).  Only what train.py's pretraining path uses.
"""
from __future__ import annotations

import argparse
import datetime
import math
import os
import sys
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


# ------------------------------------------------------------------------------------------------- init
def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated normal by inverse-CDF sampling: u ~ U(2*Phi(lo)-1, 2*Phi(hi)-1), x = erfinv(u)*std*sqrt(2)+mean."""
    phi = lambda v: 0.5 * (1.0 + math.erf(v / math.sqrt(2.0)))
    lo, hi = phi((a - mean) / std), phi((b - mean) / std)
    with torch.no_grad():
        tensor.uniform_(2 * lo - 1, 2 * hi - 1)
        tensor.erfinv_()
        tensor.mul_(std * math.sqrt(2.0))
        tensor.add_(mean)
        tensor.clamp_(min=a, max=b)
    return tensor


def fix_random_seeds(seed=31):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


# -------------------------------------------------------------------------------------------- schedules
def cosine_iter_scheduler(base_value, final_value, niter, warmup_iters=0, start_warmup_value=0):
    """Per-ITERATION cosine schedule with linear warm-up (float64 numpy array of length niter)."""
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_iters > 0 else np.array([])
    steps = np.arange(niter - warmup_iters)
    body = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * steps / len(steps)))
    schedule = np.concatenate((warm, body))
    assert len(schedule) == niter
    return schedule


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-EPOCH-parameterised cosine schedule of the finetune script (modules/utils.py:187-198)."""
    warmup_iters = int(warmup_epochs * niter_per_ep)
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    steps = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.concatenate((warm, final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * steps / len(steps)))))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def bool_flag(s):
    if s.lower() in {"off", "false", "0"}:
        return False
    if s.lower() in {"on", "true", "1"}:
        return True
    raise argparse.ArgumentTypeError("invalid value for a boolean flag")


# ------------------------------------------------------------------------------------------- parameters
def get_params_groups(model):
    """Two AdamW groups: weight-decayed tensors, and biases / 1-D tensors with weight_decay 0."""
    decayed, plain = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (plain if (name.endswith(".bias") or p.dim() == 1) else decayed).append(p)
    return [{"params": decayed}, {"params": plain, "weight_decay": 0.}]


def has_batchnorms(model):
    kinds = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)
    return any(isinstance(m, kinds) for m in model.modules())


def _arena_of(model):
    m = model.module if hasattr(model, "module") else model
    return getattr(m, "arena", None)


def clip_gradients(model, clip):
    """Per-TENSOR gradient clipping (each tensor on its own L2 norm), on the device; returns the norms (one sync)."""
    arena = _arena_of(model)
    if arena is None:
        raise RuntimeError("clip_gradients needs a ccd_amd model whose parameters live in an arena")
    from .. import ops
    cs, cb, cl = arena.opt_tables()
    norm2 = torch.zeros(len(arena.segments), dtype=torch.float32, device=arena.device)
    ops.seg_sumsq(arena.grad, cs, cb, cl, norm2)
    ops.clip_scale(arena.grad, cs, cb, cl, norm2, clip)
    norms = norm2.sqrt().tolist()
    return [n for n, p in zip(norms, arena.params.values()) if p.grad is not None]


def cancel_gradients_last_layer(epoch, model, freeze_last_layer):
    if epoch >= freeze_last_layer:
        return
    arena = _arena_of(model)
    if arena is not None:
        arena.skip_substrings.add("last_layer")   # the fused optimizer skips these tensors this iteration (== grad None)
        return
    for n, p in model.named_parameters():
        if "last_layer" in n:
            p.grad = None


def restart_from_checkpoint(ckp_path, run_variables=None, **kwargs):
    if not os.path.isfile(ckp_path):
        return
    print("Found checkpoint at {}".format(ckp_path))
    checkpoint = torch.load(ckp_path, map_location="cpu", weights_only=False)
    for key, value in kwargs.items():
        if key in checkpoint and value is not None:
            try:
                msg = value.load_state_dict(checkpoint[key], strict=False)
                print("=> loaded '{}' from checkpoint '{}' with msg {}".format(key, ckp_path, msg))
            except TypeError:
                try:
                    value.load_state_dict(checkpoint[key])
                    print("=> loaded '{}' from checkpoint: '{}'".format(key, ckp_path))
                except ValueError:
                    print("=> failed to load '{}' from checkpoint: '{}'".format(key, ckp_path))
        else:
            print("=> key '{}' not found in checkpoint: '{}'".format(key, ckp_path))
    if run_variables is not None:
        for var_name in run_variables:
            if var_name in checkpoint:
                run_variables[var_name] = checkpoint[var_name]


# ---------------------------------------------------------------------------------------------- meters
class SmoothedValue:
    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.6f} ({global_avg:.6f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        t = t.tolist()
        self.count, self.total = int(t[0]), t[1]

    median = property(lambda self: torch.tensor(list(self.deque)).median().item())
    avg = property(lambda self: torch.tensor(list(self.deque), dtype=torch.float32).mean().item())
    global_avg = property(lambda self: self.total / self.count)
    max = property(lambda self: max(self.deque))
    value = property(lambda self: self.deque[-1])

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(float(v))

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.6f}"), SmoothedValue(fmt="{avg:.6f}")
        n = len(iterable)
        width = len(str(n))
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                msg = [header, f"[{i:{width}d}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    msg.append(f"max mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}")
                print(self.delimiter.join(msg))
            end = time.time()
        total = time.time() - start
        print("{} Total time: {} ({:.6f} s / it)".format(header, str(datetime.timedelta(seconds=int(total))),
                                                         total / max(n, 1)))


# ------------------------------------------------------------------------------------------ distributed
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master):
    """print() only on the master rank unless force=True is passed."""
    import builtins
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    builtins.print = print


def init_distributed_mode(args):
    """One process per GPU; RCCL (backend name 'nccl' on ROCm) over xGMI.  Env: RANK / WORLD_SIZE / LOCAL_RANK
    (torchrun / torch.distributed.launch), SLURM_PROCID, or a single visible GPU."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % torch.cuda.device_count()
        args.world_size = int(os.environ.get("SLURM_NTASKS", 1))
    elif torch.cuda.is_available():
        print("Will run the code on one GPU.")
        args.rank, args.gpu, args.world_size = 0, 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29501")
    else:
        print("Does not support training without GPU.")
        sys.exit(1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend="nccl", init_method=getattr(args, "dist_url", None) or "env://",
                            world_size=args.world_size, rank=args.rank, device_id=torch.device("cuda", args.gpu))
    print("| distributed init (rank {}): {}".format(args.rank, getattr(args, "dist_url", "env://")), flush=True)
    dist.barrier()
    setup_for_distributed(args.rank == 0)
