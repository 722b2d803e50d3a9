#!/usr/bin/env python3
"""bench.py - images/sec of the CCD-ViT-Small pretraining step (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W       (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
The line carries `rccl_ranks` (the size of the group RCCL formed), the RCCL version and every rank's device / row count M,
gathered through the communicator: proof of how many ranks really ran.

One "step" = the full iteration of train.py:221-272 on a resident synthetic batch: student fwd
(2 views), teacher fwd, seg + DINO loss, centre update, backward, per-tensor clip, AdamW, teacher EMA.  bf16 MFMA
operands, fp32 accumulation / master weights.  Prints ONE JSON line on rank 0 (contract in the task statement) with
the extra `roofline` (dominant kernel, timed live with HIP events) and `cpu_baseline` (the CPU oracle on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TF = 2500.0       # MI355X dense bf16 MFMA peak (guide: MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (~6.3 TB/s achievable)
GF_PER_IMAGE = {"vit_small": (96.7 + 8.7, 44.56e-3 * 4), "vit_base": (167.6 + 10.1, 45.09e-3 * 4),
                "vit_tiny": (25.6 + 6.6, 43.9e-3 * 4),
                # E = 768 / 12 heads: 12 (2 * 256 * 12 E^2 + 4 * 256^2 E) = 45.9 GF per view forward, x 8 (2 views: student fwd +
                # 2 x bwd + teacher fwd); segmentation head 2.13 GF per view forward x 6; head 46.13 MF per row x 4
                "vit_base_768": (367.2 + 12.8, 46.13e-3 * 4)}   # (backbone+seg GF/img, head GF per selected row pair) SURVEY 8d


# timer kind (ccd_amd/ops.py spans) -> the kernel(s) of ccd_amd/csrc/kernels that the launches of that kind run at the default policy
KIND_KERNEL = {
    "mlp_fused": "ccd::mlp_fused_kernel<E, store_u> (mlp_fused.h)",
    "proj_mlp_fused": "ccd::mlp_fused_kernel<E, store_u, PROJ = true> (mlp_fused.h)",
    "gemm_nt_lnbwd": "ccd::rowgemm_kernel<E, ring, RG_LNBWD> (rowgemm.h)",
    "gemm_nt_resid": "ccd::gemm_row384_kernel<EPI_RESID_LN> (gemm_row384.h) / gemm_bf16_kernel<NT, EPI_RESID> (gemm.h)",
    "gemm_nt_bf16": "ccd::gemm256_kernel<EPI_BF16> (gemm256.h) / gemm_bf16_kernel<NT, EPI_BF16> (gemm.h)",
    "gemm_nt_gelu": "ccd::gemm256_kernel<EPI_GELU> (gemm256.h)",
    "gemm_nt_dgelu": "ccd::gemm256_kernel<EPI_DGELU> (gemm256.h)",
    "gemm_nt_f32": "ccd::gemm256_kernel<EPI_F32> (gemm256.h) / gemm_bf16_kernel<NT, EPI_F32> (gemm.h)",
    "gemm_nt_atomic": "ccd::gemm_bf16_kernel<NT, EPI_ATOMIC> (gemm.h)",
    "gemm_tn_atomic": "ccd::gemm_tn384_kernel<4, 2, 4> (gemm_tn384.h) / gemm_bf16_kernel<TN, EPI_ATOMIC> (gemm.h)",
    "gemm_tn_f32": "ccd::gemm_bf16_kernel<TN, EPI_F32> (gemm.h)",
    "conv_gemm": "ccd::gemm_bf16_kernel<NT, epi, GATHER> (gemm.h, implicit-GEMM convolutions)",
    "conv_wgrad": "ccd::gemm_bf16_kernel<TN, EPI_ATOMIC, GATHER> (gemm.h)",
    "attention_fwd": "ccd::attention_fwd_kernel (attention_fwd.h)",
    "attention_bwd": "ccd::attention_bwd_dq_kernel + ccd::attention_bwd_dkv_tr_kernel (attention_bwd.h)",
    "layernorm_fwd": "ccd::ln_fwd_kernel (layernorm.h)",
    "layernorm_bwd": "ccd::ln_bwd_kernel (layernorm.h)",
    "dino_loss_fwd": "ccd::dino_loss_fwd_kernel (loss.h)",
    "dino_loss_bwd": "ccd::dino_loss_bwd_kernel (loss.h)",
}


def cpu_baseline(arch, steps=2):
    """The CPU oracle (oracle/, a restatement of the reference's CPU path) on BASELINE config #1: B=8, fp32."""
    from ccd_amd.synthetic import make_batch
    from oracle import ccd_oracle as O
    spec = O.Spec(norm_last_layer=False, drop_path_rate=0.0, **O.ARCH[arch])
    student, teacher = O.build_pair(spec, seed=0)
    center, opt = torch.zeros(1, spec.out_dim), O.AdamWState()
    batch = make_batch(8, seed=0)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        rec = O.train_iteration(student, teacher, center, opt, batch, 1, 1e-5, 0.04, 0.9995)
        center = rec["center"]
        if i:
            times.append(time.perf_counter() - t0)
    sec = sorted(times)[len(times) // 2]
    return {"value": round(8.0 / sec, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{arch} B=8 fp32, {steps} timed steps after 1 warm-up, median {sec:.2f} s/step (oracle/ccd_oracle.py)"}


def pmc_traffic(kind, a):
    """HBM bytes per launch of the dominant GEMM kind from the committed PMC passes of this same command
    (tools/pmc_traffic.sh -> profiles/pmc_traffic.json; counters cannot be read from inside the process).
    Only valid for the default workload the passes were taken on; otherwise null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if not os.path.isfile(path) or a.arch != "vit_small" or a.batch != 256 or a.epoch >= 30:
        return None
    with open(path) as f:
        table = json.load(f)
    # the passes are only evidence for the kernels they were taken on: the file carries a digest of the kernel sources
    # (tools/pmc_traffic.py stamps it), and the line reports null as soon as any kernel source has changed since
    if table.get("_kernel_sources_sha256") != kernel_sources_digest():
        return None
    rec = table.get(kind)
    if kind.startswith("_"):                # whole-step totals are plain numbers
        return rec
    return rec["bytes_per_launch"] if rec else None


def kernel_sources_digest():
    """sha256 over ccd_amd/csrc (sorted file names + contents): identifies the kernels a PMC pass was taken on."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ccd_amd", "csrc")
    h = hashlib.sha256()
    for d, _, fs in sorted(os.walk(root)):
        for f in sorted(fs):
            if f.endswith((".h", ".hip", ".sh")):
                h.update(os.path.relpath(os.path.join(d, f), root).encode())
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def emit(line):
    """Print the result as the LAST line of stdout: librccl logs its path through C stdio, which is block-buffered on a
    pipe and would otherwise be flushed after Python's output at exit."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(line), flush=True)


def finetune_cpu_baseline(arch, B=8, steps=2):
    """The CPU oracle of the finetune step (oracle/finetune_oracle.py) on this host: B=8, fp32, dropout off."""
    from oracle import ccd_oracle as O
    from oracle import finetune_oracle as FO
    spec = FO.FtSpec(vit=O.Spec(**O.ARCH[arch]))
    net = FO.init_finetune(spec, seed=0)
    opt = O.AdamWState()
    g = torch.Generator().manual_seed(0)
    img = torch.randn(B, 3, 32, 128, generator=g)
    targets = FO.str2tensor(["benchmark%d" % i for i in range(B)])
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        FO.train_iteration(net, opt, img, targets, 1e-4)
        if i:
            times.append(time.perf_counter() - t0)
    sec = sorted(times)[len(times) // 2]
    return {"value": round(B / sec, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{arch} finetune B={B} fp32, {steps} timed steps after 1 warm-up, median {sec:.2f} s/step "
                      "(oracle/finetune_oracle.py)"}


def main_finetune(a, world, rank, dev, use_dist):
    """BASELINE config #5: CCD_vision_model finetune step (train_finetune.py:262-289) - ViT encoder fwd/bwd, Mlp, 6-layer
    NRTR decoder, TFLoss, AdamW - on a resident synthetic labelled batch; dropout 0.1 and drop_path 0.1 ON as shipped."""
    from ccd_amd import finetune as ft, ops
    from ccd_amd.parallel import DataParallel
    torch.manual_seed(0)
    model = ft.build_model(ft.FinetuneConfig(arch=a.arch, drop_path_rate=0.1), dev)
    net = DataParallel(model) if use_dist else model
    opt = ft.make_optimizer(model)
    B = a.batch
    g = torch.Generator().manual_seed(1000 + rank)
    images = torch.randn(B, 3, 32, 128, generator=g).to(dev)
    words = ["".join(model.label_convertor.idx2char[int(c)] for c in torch.randint(0, 90, (int(n),), generator=g))
             for n in torch.randint(3, 16, (B,), generator=g)]
    labels = model.label_convertor.str2tensor(words).to(dev)

    def step():
        return ft.training_iteration(net, opt, images, labels, 1e-4)[0]

    warm_summary = None                       # see main(): every launch timed in the last warm-up step, then one kind only
    for i in range(a.warmup):
        if not a.no_kernel_timer and i == a.warmup - 1:
            ops.TIMER = ops.KernelTimer()
        loss = step()
    if ops.TIMER is not None:
        torch.cuda.synchronize()
        warm_summary = ops.TIMER.summary()
    dominant = max(warm_summary.items(), key=lambda kv: kv[1]["ms"])[0] if warm_summary else None
    timer = None if a.no_kernel_timer else ops.KernelTimer(only={dominant} if dominant else None)
    ops.TIMER = timer
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    final_loss = loss.item()
    assert final_loss == final_loss and abs(final_loss) < 1e4, f"non-finite loss {final_loss}"
    census = rank_census(world, rank, dev, 25 * B)
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        ips = B * world * a.steps / elapsed
        E = model.backbone.embed_dim
        L, D, T = model.decoder.dec_spec.L, 512, 25
        # FLOPs per image (multiply-add = 2, backward = 2x forward): backbone 12.09 GF/view fwd (SURVEY 8d, E=384),
        # Mlp, per-layer K/V over the 256 tokens, decoder rows (self/cross projections, FFN), attention products
        vit_fwd = {"vit_small": 12.09e9, "vit_base": 20.95e9, "vit_tiny": 3.2e9}.get(a.arch, 12.09e9)
        enc = 2 * 256 * (E * 512 + 512 * D) + 2 * 256 * L * 2 * D * D
        dec = L * (2 * T * (4 * D * D + 2 * D * D + 2 * D * 256) + 4 * T * T * D + 4 * T * 256 * D) + 2 * T * D * 92
        gf_img = 3 * (vit_fwd + enc + dec) / 1e9
        line = {"metric": "images/sec (32x128 crops) CCD finetune step (ViT encoder + NRTR decoder)", "value": round(ips, 2),
                "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"CCD_vision_model finetune {a.arch} bf16, bs={B}/GPU, 6-layer NRTR decoder, T=25, "
                                       "dropout 0.1, drop_path 0.1, AdamW, synthetic labelled batch in HBM (BASELINE config #5)",
                           "global_batch": B * world, "parallelism": f"dp{world}", "gflop_per_image": round(gf_img, 1),
                           "step_frac_of_mfma_peak": round(ips / world * gf_img / 1e3 / PEAK_BF16_TF, 4),
                           "final_loss": round(final_loss, 4)}}
        line.update(census)
        if timer is not None:
            summ = timer.summary()
            key, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            avg_ms = d["ms"] / d["launches"]
            tflops = d["flops"] / d["launches"] / avg_ms / 1e9
            gbs = d["bytes"] / d["launches"] / avg_ms / 1e6
            intensity = d["flops"] / max(d["bytes"], 1.0)
            hbm_bound = intensity < PEAK_BF16_TF * 1e12 / (PEAK_HBM_GBS * 1e9)
            line["roofline"] = {"bound": "hbm" if hbm_bound else "mfma", "kernel": KIND_KERNEL.get(key, key), "kind": key,
                                "achieved": round(gbs if hbm_bound else tflops, 1),
                                "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TF,
                                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                                "frac": round(gbs / PEAK_HBM_GBS if hbm_bound else tflops / PEAK_BF16_TF, 4), "traffic": None,
                                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": d["launches"] // a.steps,
                                }
            table, per = (warm_summary, 1) if warm_summary else (summ, a.steps)
            line["roofline"].update({
                "by_kind_source": "last warm-up step (every GEMM launch timed)" if warm_summary else "timed region",
                "gemm_ms_per_step": round(sum(v["ms"] for v in table.values()) / per, 3),
                "by_kind_ms_per_step": {k: round(v["ms"] / per, 3) for k, v in sorted(table.items())},
                "by_kind_tflops": {k: round(v["flops"] / v["ms"] / 1e9, 1) for k, v in sorted(table.items())}})
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = finetune_cpu_baseline(a.arch)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


def main_recognize(a, world, rank, dev, use_dist):
    """Inference of the recogniser: DINO_Finetune.forward_test (dino_vision.py:233-262) = encoder forward + Mlp + 25
    greedy decoding steps of the 6-layer decoder on a resident synthetic batch; replicas only (no collective)."""
    from ccd_amd import finetune as ft
    torch.manual_seed(0)
    model = ft.build_model(ft.FinetuneConfig(arch=a.arch, drop_path_rate=0.1), dev)
    model.eval()
    B = a.batch
    images = torch.randn(B, 3, 32, 128, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)

    def step():
        with torch.no_grad():
            return model(images, None, return_loss=False)

    for _ in range(a.warmup):
        probs = step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        probs = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    assert tuple(probs.shape) == (B, 25, 92) and bool(torch.isfinite(probs).all())
    census = rank_census(world, rank, dev, 25 * B)
    if rank == 0:
        line = {"metric": "images/sec (32x128 crops) CCD text recognition, greedy decoding (25 steps)",
                "value": round(B * world * a.steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"DINO_Finetune.forward_test {a.arch} bf16, bs={B}/GPU, 6-layer NRTR decoder, 25 greedy "
                                       f"steps, HIP graph {'on' if os.environ.get('CCD_DECODE_GRAPH', '1') != '0' else 'off'}",
                           "global_batch": B * world, "parallelism": f"replicas x{world}"}}
        line.update(census)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command under torch.distributed.run with one rank per GPU
    (what README.md:33 of the reference does with `torch.distributed.launch`), rendezvous on 127.0.0.1 and an OS-chosen port.
    The JSON line is printed by rank 0 of the child job; this process only forwards the exit code."""
    import socket
    import subprocess
    if os.environ.get("BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < n:
        sys.exit(f"bench.py: --gpus {n} but this node exposes {torch.cuda.device_count()} GPU(s)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def rank_census(world, rank, dev, rows):
    """What the collective library itself saw: the size of the group, its version, and every rank's (device, selected rows M)
    gathered THROUGH it - evidence in the JSON line that N ranks really joined one RCCL communicator."""
    info = {"rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1}
    if dist.is_initialized():
        mine = torch.tensor([rank, dev.index if dev.type == "cuda" else -1, int(rows)], dtype=torch.int64, device=dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        info["ranks"] = [{"rank": int(t[0]), "device": int(t[1]), "rows_M": int(t[2])} for t in got]
        if dist.get_backend() == "nccl":
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    return info


def rendezvous_dry_run(a, world, rank):
    """BENCH_BACKEND=gloo: no GPU work - the ranks the launcher started join ONE process group, run the barrier / max-over-ranks
    timing skeleton of the real bench and rank 0 prints the census.  Exists so that the N-rank launch path of this file is
    exercised in the GPU-less build container (tests/test_bench_cli.py); it never produces a throughput number."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    dist.barrier()
    t0 = time.perf_counter()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    census = rank_census(world, rank, dev, rows=100 + rank)
    dist.destroy_process_group()
    if rank == 0:
        emit({"dry_run": True, "backend": "gloo", "n_gpus": world, "gpus_arg": a.gpus, **census})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--arch", default="vit_small")
    ap.add_argument("--out-dim", type=int, default=65536)
    ap.add_argument("--workload", default="pretrain", choices=["pretrain", "finetune", "recognize"],
                    help="pretrain = the BASELINE metric (default); finetune = BASELINE config #5 (SURVEY 8f row 1); "
                         "recognize = greedy-decoding inference of the finetuned model (forward_test)")
    ap.add_argument("--epoch", type=int, default=1, help="pseudo-epoch handed to the model: >= 30 takes the predicted-mask "
                    "branch (dino_vision.py:64-70; the batch then carries its characters in the images)")
    ap.add_argument("--graph", action="store_true", help="pretrain, 1 GPU: replay the step as ONE HIP graph "
                    "(pretrain.GraphedTrainingStep) - the small-batch regime, where the host's launches bound the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    a = ap.parse_args()
    if os.environ.get("BENCH_NOTIMER") == "1":
        a.no_kernel_timer = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)               # `python bench.py --gpus N`: become the launcher of N ranks of this command
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("BENCH_BACKEND", "nccl")       # "gloo": rendezvous dry run without GPUs (tests/test_bench_cli.py)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"      # the latter: 1-rank RCCL smoke of the N>1 path
    if backend == "gloo":
        return rendezvous_dry_run(a, world, rank)
    if torch.cuda.device_count() < world:
        sys.exit(f"bench.py: --gpus {world} but this node exposes {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.workload == "finetune":
        return main_finetune(a, world, rank, dev, use_dist)
    if a.workload == "recognize":
        return main_recognize(a, world, rank, dev, use_dist)

    from ccd_amd import ops, pretrain
    from ccd_amd.loss.Dino_loss import DINOLoss
    from ccd_amd.parallel import DataParallel
    from ccd_amd.synthetic import make_batch, make_text_like_batch

    torch.manual_seed(0)
    student, teacher = pretrain.build_networks(arch=a.arch, out_dim=a.out_dim, drop_path_rate=0.1,
                                               norm_last_layer=False, device=dev)
    if use_dist:
        student = torch.nn.SyncBatchNorm.convert_sync_batchnorm(student)      # train.py:96-98
        if world == 1:           # BENCH_FORCE_DIST=1: the N > 1 code path on one rank - every collective is really issued
            from ccd_amd import seghead
            seghead.FORCE_SYNC = True
        model = DataParallel(student, reduce_at_world1=(world == 1))
    else:
        model = student
    dino_loss = DINOLoss(a.out_dim, 2, 0.04, 0.04, 0, 100).to(dev)
    opt = pretrain.make_optimizer(student, clip_grad=3.0)
    B = a.batch
    images, masks, metrics = (make_text_like_batch if a.epoch >= 30 else make_batch)(B, seed=1000 + rank, device=dev)
    lr, wd, mom = 0.0005 * B * world / 256.0, 0.04, 0.9995

    graphed = None
    if a.graph:
        if use_dist:
            raise SystemExit("--graph captures a single-process step")
        graphed = pretrain.GraphedTrainingStep(student, teacher, dino_loss, opt, eager_steps=2)
        a.warmup = max(a.warmup, 3)          # two eager iterations, then the capturing call: all inside the warm-up

    def step():
        if graphed is not None:
            return graphed(images, masks, metrics, epoch=a.epoch, lr=lr, wd=wd, momentum=mom)
        return pretrain.training_iteration(model, teacher, dino_loss, opt, images, masks, metrics, epoch=a.epoch, lr=lr,
                                           wd=wd, momentum=mom)

    # Kernel timing (HIP events on the launching stream): the last warm-up step times EVERY GEMM launch - that picks the
    # dominant kind and fills the per-kind table; the timed region then only brackets the launches of that one kind
    # (two event records per launch on ~590 launches cost 1.3 ms of a 61 ms step; on ~48 launches 0.1 ms).
    # (--graph: events cannot be read out of a replayed graph - the table comes from the second EAGER warm-up step, the timed
    # region replays untimed and `roofline` says so)
    warm_summary = None
    timed_warm = 1 if graphed is not None else a.warmup - 1
    for i in range(a.warmup):
        if not a.no_kernel_timer and i == timed_warm:
            ops.TIMER = ops.KernelTimer()
        loss = step()
        if ops.TIMER is not None:
            torch.cuda.synchronize()
            warm_summary = ops.TIMER.summary()
            ops.TIMER = None
    dominant = max(warm_summary.items(), key=lambda kv: kv[1]["ms"])[0] if warm_summary else None
    timer = None if a.no_kernel_timer or graphed is not None else ops.KernelTimer(only={dominant} if dominant else None)
    ops.TIMER = timer
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    if graphed is not None and warm_summary:          # the per-kind table of the eager step stands in for the timed region
        timer = ops.KernelTimer()
        timer.summary = lambda: {k: dict(v, ms=v["ms"] * a.steps, launches=v["launches"] * a.steps, flops=v["flops"] * a.steps,
                                         bytes=v["bytes"] * a.steps) for k, v in warm_summary.items()}
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    final_loss = loss.item()
    assert final_loss == final_loss and abs(final_loss) < 1e4, f"non-finite loss {final_loss}"

    # every rank: its selected-row count M (ranks hold different batches, so M differs), gathered through the collective library
    from ccd_amd import engine
    ids = ops.ccl_label(masks)
    m_local = int(engine.Selection(torch.cat([ids, ops.warp_idmap(ids, metrics)]), B).M)
    census = rank_census(world, rank, dev, m_local)
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        ips = B * world * a.steps / elapsed
        # whole-step MFMA fraction: SURVEY 8(d) FLOP table with the measured rows-per-image of this batch
        m_rows = m_local / B
        body, per_row = GF_PER_IMAGE.get(a.arch, GF_PER_IMAGE["vit_small"])
        gf_img = body + per_row * 2 * m_rows
        arch_name = {"vit_small": "CCD-ViT-Small", "vit_base": "CCD-ViT-Base", "vit_tiny": "CCD-ViT-Tiny",
                     "vit_base_768": "CCD-ViT-Base (768/12)"}.get(a.arch, a.arch)
        line = {"metric": f"images/sec (32x128 crops, 2 views) {arch_name} pretrain step", "value": round(ips, 2),
                "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"CCD_pretrain_{a.arch} bf16, bs={B}/GPU, 2 views 32x128, out_dim={a.out_dim}, "
                                       f"drop_path 0.1, " + ("predicted-mask branch (pseudo-epoch %d), text-like " % a.epoch
                                                            if a.epoch >= 30 else "dataset-mask branch (pseudo-epoch %d), " % a.epoch)
                                       + "synthetic batch in HBM",
                           "global_batch": B * world, "parallelism": f"dp{world}",
                           "rows_per_image_per_view": round(m_rows, 3), "gflop_per_image": round(gf_img, 1),
                           "step_frac_of_mfma_peak": round(ips / world * gf_img / 1e3 / PEAK_BF16_TF, 4),
                           "final_loss": round(final_loss, 4)}}
        line.update(census)
        if graphed is not None:
            line["config"]["workload"] += ", step replayed as one HIP graph"
            line["hip_graph"] = {"captures": graphed.captures, "replays": graphed.replays,
                                 "roofline_source": "eager warm-up step (HIP events cannot be read out of a graph replay)"}
        if timer is not None:
            summ = timer.summary()
            key, d = max(summ.items(), key=lambda kv: kv[1]["ms"])          # the GEMM kind with the largest total time
            avg_ms = d["ms"] / d["launches"]
            tflops = d["flops"] / d["launches"] / avg_ms / 1e9
            gbs = d["bytes"] / d["launches"] / avg_ms / 1e6
            # which roof binds this kind: arithmetic intensity of its ALGORITHMIC work against the machine ridge
            ridge = PEAK_BF16_TF * 1e12 / (PEAK_HBM_GBS * 1e9)                # 312 flop/B on MI355X
            intensity = d["flops"] / max(d["bytes"], 1.0)
            hbm_bound = intensity < ridge
            line["roofline"] = {"bound": "hbm" if hbm_bound else "mfma",
                                "kernel": KIND_KERNEL.get(key, key), "kind": key,
                                "achieved": round(gbs if hbm_bound else tflops, 1),
                                "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TF,
                                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                                "frac": round(gbs / PEAK_HBM_GBS if hbm_bound else tflops / PEAK_BF16_TF, 4),
                                "traffic": pmc_traffic(key, a),
                                # HBM bytes of ONE WHOLE STEP from the same PMC passes (every dispatch, fetch + write; docs/LAB_NOTEBOOK.md section 4d has
                                # the budget by tensor) - null unless the shipped kernel sources are the ones the passes ran on
                                "step_bytes": pmc_traffic("_step_bytes", a),
                                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                                "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
                                "arithmetic_intensity_flop_per_byte": round(intensity, 1),
                                "mfma_tflops": round(tflops, 1), "mfma_frac": round(tflops / PEAK_BF16_TF, 4),
                                "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
                                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": d["launches"] // a.steps,
                                }
            table, per = (warm_summary, 1) if warm_summary else (summ, a.steps)
            for k, v in table.items():      # the loss kernels are launched for the worst case 2 * 26 * B rows and read M on the device
                if k.startswith("dino_loss"):
                    live = 2.0 * m_local / (2 * 26 * B)
                    v["flops"], v["bytes"] = v["flops"] * live, v["bytes"] * live
            line["roofline"].update({
                "by_kind_source": "last warm-up step (every GEMM / attention / LayerNorm / loss launch timed)" if warm_summary else "timed region",
                "timed_ms_per_step": round(sum(v["ms"] for v in table.values()) / per, 3),
                "gemm_ms_per_step": round(sum(v["ms"] for k, v in table.items() if k.startswith(("gemm", "mlp", "proj_mlp", "conv"))) / per, 3),
                "by_kind_ms_per_step": {k: round(v["ms"] / per, 3) for k, v in sorted(table.items())},
                "by_kind_tflops": {k: round(v["flops"] / v["ms"] / 1e9, 1) for k, v in sorted(table.items())},
                "by_kind_algorithmic_gbs": {k: round(v["bytes"] / v["ms"] / 1e6, 1) for k, v in sorted(table.items())}})
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.arch)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
