"""The N>1 path on CPU: world_size-2 gloo process groups exercising the gradient reducer, the rank-0 parameter
broadcast, the centre all-reduce semantics (Dino_loss.py:133-143) and a full data-parallel iteration of the tiny
model (HIP kernels executed by the CPU SIMT executor in every rank)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


class _FakeArena:
    def __init__(self, total, ranges):
        self.total, self._ranges = total, ranges
        self.grad = torch.zeros(total)

    def range_of(self, prefix):
        return self._ranges[prefix]


def _reducer_worker(rank, world, port):
    _init(rank, world, port)
    from ccd_amd.parallel import GradReducer
    ranges = {"head.": (700, 1000), "blocks.1.": (300, 500), "blocks.0.": (100, 300), "patch.": (0, 64)}
    arena = _FakeArena(1000, ranges)
    arena.grad.copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1))
    red = GradReducer(arena, bucket_elems=250)
    for p in ("head.", "blocks.1.", "blocks.0.", "patch."):      # the order the backward pass announces them in
        red.mark_ready(p)
    red.finish()                                                  # covers the gaps [64,100) and [500,700) as well
    want = torch.arange(1000, dtype=torch.float32) * (1 + 2) / 2.0
    assert torch.allclose(arena.grad, want), (arena.grad - want).abs().max()
    # second round re-uses the reducer
    arena.grad.fill_(float(rank))
    red.mark_ready("head.")
    red.finish()
    assert torch.allclose(arena.grad, torch.full((1000,), 0.5))
    dist.destroy_process_group()


def _center_worker(rank, world, port):
    _init(rank, world, port)
    from backends import Backend
    from ccd_amd.loss.Dino_loss import DINOLoss
    from oracle import ccd_oracle as O
    with Backend("sim"):
        K = 512
        g = torch.Generator().manual_seed(10 + rank)
        m_local = 5 + 3 * rank                                      # ranks select different numbers of rows
        t = torch.zeros(2 * 26 * 2, K)
        t[:2 * m_local] = torch.randn(2 * m_local, K, generator=g)
        loss = DINOLoss(K, 2, 0.04, 0.04, 0, 10)
        loss.center.copy_(torch.linspace(-1, 1, K).view(1, K))
        loss.update_center(t, torch.tensor([m_local], dtype=torch.int32))
        # reference semantics: all_reduce(sum of LOCAL rows) / (LOCAL row count * world)  - reproduced, not "fixed"
        sums = [torch.zeros(1, K) for _ in range(world)]
        dist.all_gather(sums, t[:2 * m_local].sum(0, keepdim=True))
        want = torch.linspace(-1, 1, K).view(1, K) * 0.9 + (sum(sums) / (2 * m_local * world)) * 0.1
        assert torch.allclose(loss.center, want, atol=1e-5), (loss.center - want).abs().max()
        want2 = O.center_update(torch.linspace(-1, 1, K).view(1, K), t[:2 * m_local], world,
                                all_reduce=lambda x: dist.all_reduce(x))
        assert torch.allclose(loss.center, want2, atol=1e-5)
    dist.destroy_process_group()


def _dp_step_worker(rank, world, port):
    _init(rank, world, port)
    from backends import Backend
    import model_checks as mc
    from ccd_amd import pretrain
    from ccd_amd.loss.Dino_loss import DINOLoss
    from ccd_amd.parallel import DataParallel
    from ccd_amd.synthetic import make_batch
    with Backend("sim") as b:
        torch.manual_seed(3 + 100 * rank)                             # deliberately different initial weights per rank
        np.random.seed(3)
        student, teacher = mc.tiny_networks(b.device)
        model = DataParallel(student)                                 # broadcasts rank 0's parameters
        flat = student.arena.flat.clone()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1]), "parameters not broadcast from rank 0"
        teacher.backbone.load_state_dict(student.backbone.state_dict())
        teacher.head.load_state_dict(student.head.state_dict())
        dino_loss = DINOLoss(512, 2, 0.04, 0.04, 0, 40)
        opt = pretrain.make_optimizer(student, clip_grad=3.0)
        images, masks, metrics = make_batch(1, seed=11 + rank)       # every rank its own shard (1 image: CPU executor)
        # local gradients without synchronisation
        s_out = student(images, metrics, masks, 1)
        with torch.no_grad():
            t_out = teacher(images, metrics, None, None, clusters=s_out["zero"])
        from ccd_amd import ops
        s_out["gt"] = [masks, ops.warp_idmap(ops.mask_to_idmap(masks), metrics)]
        c0 = dino_loss.center.clone()
        loss = dino_loss(s_out, t_out, 1)
        student.arena.zero_grad()
        loss.backward()
        local = student.arena.grad.clone()
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        mean = (both[0] + both[1]) / 2
        dino_loss.center.copy_(c0)
        # the real iteration: same forward/backward, gradients averaged bucket by bucket during backward
        bn_state = {k: v.clone() for k, v in student.state_dict().items() if "running_" in k or "num_batches" in k}
        seen = {}
        orig_step = opt.step
        opt.step = lambda: seen.setdefault("grad", student.arena.grad.clone()) is None or orig_step()
        pretrain.training_iteration(model, teacher, dino_loss, opt, images, masks, metrics, 1, 2e-4, 0.05, 0.99)
        err = (seen["grad"] - mean).abs().max().item()
        scale = mean.abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (err, scale)               # atomics re-order fp32 sums between the two passes
        flat = student.arena.flat.clone()
        dist.all_gather(gathered, flat)
        assert torch.allclose(gathered[0], gathered[1], atol=1e-7), "replicas diverged after the optimizer step"
    dist.destroy_process_group()


def _syncbn_worker(rank, world, port):
    """Native SegHead with SyncBatchNorm on 2 ranks (one image each) == one process on both images: same logits,
    same running statistics, and the ranks' weight gradients sum to the full-batch gradient."""
    _init(rank, world, port)
    import copy
    from backends import Backend
    import kernel_checks as kc
    from ccd_amd import seghead as sh
    from ccd_amd.modules.segmentor import SegHead
    with Backend("sim") as b:
        torch.manual_seed(5)
        E = 64
        head = SegHead(in_channels=E)
        with torch.no_grad():
            for m in head.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        ref = copy.deepcopy(head).float()
        g = torch.Generator().manual_seed(9)
        taps = [kc.rnd((2 * 256, E), g).to(kc.BF) for _ in range(3)]
        dl = kc.rnd((2, 2, 32, 128), g) / 4096
        # single-process reference on both images (torch fp32 library ops)
        rin = [t.float().view(2, 8, 32, E).permute(0, 3, 1, 2).contiguous() for t in taps]
        want = ref.cls(ref.unpool2(ref.unpool1(ref.mlahead(*rin))))
        want.backward(dl)
        # this rank's shard through the HIP path with synchronised statistics
        head = torch.nn.SyncBatchNorm.convert_sync_batchnorm(head)
        mine = [t[rank * 256:(rank + 1) * 256].clone().requires_grad_(True) for t in taps]
        calls = []
        real_all_reduce = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real_all_reduce(*a, **k))[1]
        logits = sh.seg_head_forward(head, mine, 1)
        fwd_calls = len(calls)
        kc.close(logits, want[rank:rank + 1], 3e-2, 3e-2, "syncbn/logits")
        logits.backward(dl[rank:rank + 1])
        dist.all_reduce = real_all_reduce
        # statistics of independent layers travel together: one exchange per level and direction
        assert fwd_calls <= 4 and len(calls) - fwd_calls <= 4, calls
        refp, refb = dict(ref.named_parameters()), dict(ref.named_buffers())
        for name, buf in head.named_buffers():
            if not name.startswith("conv_mla") and "num_batches" not in name:
                kc.close(buf, refb[name], 2e-2, 2e-3, f"syncbn/{name}")
        for name, p in head.named_parameters():
            if name.startswith("conv_mla") or name in ("unpool1.0.bias", "unpool2.0.bias"):
                continue
            total = p.grad.clone()
            dist.all_reduce(total)
            c = kc._cos(total, refp[name].grad)
            ratio = float(total.norm() / refp[name].grad.norm())
            assert c > 0.99 and 0.95 < ratio < 1.05, (name, c, ratio)
    dist.destroy_process_group()


def _finetune_dp_worker(rank, world, port):
    """Finetune path under data parallelism: rank-0 broadcast, gradients == mean of the ranks' local gradients (the
    decoder / encoder ranges are announced by FinetuneFn's hooks, the backbone blocks by BackboneFn's), replicas equal."""
    _init(rank, world, port)
    from backends import Backend
    import model_checks as mc
    from ccd_amd import finetune as ft
    from ccd_amd.parallel import DataParallel
    with Backend("sim") as b:
        mc._register_test_arch()
        torch.manual_seed(3 + 100 * rank)
        model = ft.build_model(ft.FinetuneConfig(arch="vit_test2", drop_path_rate=0.0, decoder_n_layers=1,
                                                 decoder_max_seq_len=10), b.device, dropout=0.0)
        net = DataParallel(model)
        flat = model.arena.flat.clone()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1]), "parameters not broadcast from rank 0"
        opt = ft.make_optimizer(net)
        g = torch.Generator().manual_seed(50 + rank)
        img = torch.randn(1, 3, 32, 128, generator=g)
        labels = model.label_convertor.str2tensor([mc.FT_WORDS[rank]])
        loss, _ = model(img, labels)
        model.arena.zero_grad()
        loss.backward()                                               # hooks fire, but nothing waits: grads stay local
        net.reducer._works, net.reducer._pending, net.reducer._done = [], [], []
        local = model.arena.grad.clone()
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.barrier()
        dist.all_gather(both, local)
        seen = {}
        orig_step = opt.step
        opt.step = lambda: seen.setdefault("grad", model.arena.grad.clone()) is None or orig_step()
        ft.training_iteration(net, opt, img, labels, 2e-4)
        mean = (both[0] + both[1]) / 2
        err, scale = (seen["grad"] - mean).abs().max().item(), mean.abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (err, scale)
        flat = model.arena.flat.clone()
        dist.all_gather(gathered, flat)
        assert torch.allclose(gathered[0], gathered[1], atol=1e-7), "replicas diverged after the optimizer step"
    dist.destroy_process_group()


def _spawn(fn, port):
    mp.spawn(fn, args=(2, port), nprocs=2, join=True)


def test_grad_reducer_world2():
    _spawn(_reducer_worker, 29611)


def test_center_all_reduce_world2():
    _spawn(_center_worker, 29612)


def test_data_parallel_iteration_world2():
    _spawn(_dp_step_worker, 29613)


def test_seghead_syncbn_world2():
    _spawn(_syncbn_worker, 29614)


def test_finetune_data_parallel_world2():
    _spawn(_finetune_dp_worker, 29615)
