"""One CCD pretraining iteration on the MI355X-native modules (the body of train.py:221-272), shared by train.py,
bench.py, __graft_entry__.smoke() and the parity tests."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .loss.Dino_loss import DINOLoss
from .model.dino_vision import ABIDINOModel
from .modules import vision_transformer as vits
from .modules.segmentor import SegHead
from .optim import FusedClipAdamW, ema_update


def build_networks(arch="vit_small", patch_size=4, out_dim=65536, drop_path_rate=0.1, norm_last_layer=False,
                   use_bn_in_head=False, seg_channel=None, backbone_kwargs=None, head_kwargs=None, device="cuda"):
    """Student / teacher exactly in train.py:63-114's construction order (same RNG stream as the reference)."""
    bk, hk = backbone_kwargs or {}, head_kwargs or {}
    if arch in vits.__dict__:
        student_b = vits.__dict__[arch](patch_size=patch_size, drop_path_rate=drop_path_rate, **bk)
        teacher_b = vits.__dict__[arch](patch_size=patch_size, **bk)
    else:   # explicit dimensions (tests / tiny models)
        mk = lambda **kw: vits.VisionTransformer(patch_size=patch_size, qkv_bias=True, mlp_ratio=4,
                                                 norm_layer=lambda e: nn.LayerNorm(e, eps=1e-6), **bk, **kw)
        student_b, teacher_b = mk(drop_path_rate=drop_path_rate), mk()
    E = student_b.embed_dim
    student = ABIDINOModel(student_b, SegHead(in_channels=seg_channel or E, mla_channels=128, mlahead_channels=64,
                                              num_classes=2),
                           vits.DINOHead(E, out_dim, use_bn=use_bn_in_head, norm_last_layer=norm_last_layer, **hk))
    teacher = ABIDINOModel(teacher_b, None, vits.DINOHead(E, out_dim, use_bn_in_head, **hk))
    student, teacher = student.to(device), teacher.to(device)
    student.ensure_arena()
    teacher.ensure_arena()
    teacher.backbone.load_state_dict(student.backbone.state_dict())
    teacher.head.load_state_dict(student.head.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    teacher.ensure_arena()          # refreshes the bf16 mirrors after the state-dict copy
    return student, teacher


def make_optimizer(student_module, clip_grad=3.0):
    arena = student_module.ensure_arena()
    opt = FusedClipAdamW(arena, clip_grad=clip_grad)
    opt.mark_unused(student_module.unused_parameter_names())
    return opt


def _set_schedule(optimizer, lr, wd):
    for i, g in enumerate(optimizer.param_groups):
        g["lr"] = float(lr)
        if i == 0:
            g["weight_decay"] = float(wd)


FWD_SPLIT = None          # ("cu" | "xcd", teacher share): student / teacher backbone passes on CU-masked streams; None = one after the other
_FWD_SPLIT_ENV_READ = False


def _forward_split(device):
    global FWD_SPLIT, _FWD_SPLIT_ENV_READ
    if not _FWD_SPLIT_ENV_READ:
        _FWD_SPLIT_ENV_READ = True
        if FWD_SPLIT is None:
            from .streams import env_partition
            FWD_SPLIT = env_partition()
    if FWD_SPLIT is None or torch.device(device).type != "cuda":
        return None
    from .streams import partition
    return partition(device, teacher_per_xcd=FWD_SPLIT[1], layout=FWD_SPLIT[0])


def _forward_backward(student, teacher, dino_loss, optimizer, images, masks, metrics, epoch, check_finite=False):
    """train.py:221-246: both networks, the loss, zero_grad, backward (+ the gradient all-reduce of a wrapped student)."""
    split = _forward_split(images.device)
    if split is None:
        return _forward_backward_on_stream(student, teacher, dino_loss, optimizer, images, masks, metrics, epoch, check_finite, None)
    # The CU-masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags): any operation on the legacy
    # default stream - an event record is one - waits for everything queued on them and holds back everything queued after it.
    # torch's default stream IS that stream, so the pass runs on a non-blocking stream of its own and joins the caller's at the end
    # (measured with the default stream in the middle: the two partitions ran one after the other, 67 ms per step instead of 50).
    dev = images.device
    caller = torch.cuda.current_stream(dev)
    work = _work_stream(dev)
    work.wait_stream(caller)
    with torch.cuda.stream(work):
        loss = _forward_backward_on_stream(student, teacher, dino_loss, optimizer, images, masks, metrics, epoch, check_finite, split)
    caller.wait_stream(work)
    loss.record_stream(caller)
    return loss


_WORK_STREAMS = {}


def _work_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _WORK_STREAMS:
        _WORK_STREAMS[key] = torch.cuda.Stream(device=device)
    return _WORK_STREAMS[key]


def _forward_backward_on_stream(student, teacher, dino_loss, optimizer, images, masks, metrics, epoch, check_finite, split):
    metrics = metrics.float()
    if split is None:
        s_out = student(images, metrics, masks, epoch, clusters=None)
        with torch.no_grad():
            t_out = teacher(images, metrics, None, None, clusters=s_out["zero"], index=None)
    else:
        # both backbone passes side by side on disjoint compute units (ccd_amd/streams.py): the teacher's is enqueued first and
        # nobody waits for it until its tokens are pooled; the student's joins the calling stream when it returns
        from . import engine
        s_stream, t_stream, s_cus, t_cus = split
        t_mod = teacher.module if hasattr(teacher, "module") else teacher
        main = torch.cuda.current_stream(images.device)
        try:
            engine.FORWARD_HOP = engine.ForwardHop(t_stream, s_cus, join=False)
            with torch.no_grad():
                t_tokens = t_mod.backbone_tokens(images)
            engine.FORWARD_HOP = engine.ForwardHop(s_stream, t_cus, join=True)
            s_out = student(images, metrics, masks, epoch, clusters=None)
        finally:
            engine.FORWARD_HOP = None
        main.wait_stream(t_stream)
        with torch.no_grad():
            t_out = teacher(images, metrics, None, None, clusters=s_out["zero"], index=None, tokens=t_tokens)
    # gt = [masks, warped masks > 0.1]  (train.py:234-237); the warped half stays an id map on the device
    masks_image = ops.warp_idmap(ops.mask_to_idmap(masks.contiguous().float()), metrics.contiguous())
    s_out["gt"] = [masks, masks_image]
    loss = dino_loss(s_out, t_out, epoch)
    if check_finite and not math.isfinite(loss.item()):
        raise FloatingPointError("Loss is {}, stopping training".format(loss.item()))
    optimizer.zero_grad()
    loss.backward()
    if hasattr(student, "finish_gradient_sync"):
        student.finish_gradient_sync()
    return loss


def training_iteration(student, teacher, dino_loss: DINOLoss, optimizer: FusedClipAdamW, images, masks, metrics,
                       epoch, lr, wd, momentum, freeze_last_layer=1, check_finite=False):
    """student / teacher may be wrapped (ccd_amd.parallel.DataParallel); returns the loss as a device tensor."""
    s_mod = student.module if hasattr(student, "module") else student
    t_mod = teacher.module if hasattr(teacher, "module") else teacher
    _set_schedule(optimizer, lr, wd)
    loss = _forward_backward(student, teacher, dino_loss, optimizer, images, masks, metrics, epoch, check_finite)
    if epoch < freeze_last_layer:
        s_mod.arena.skip_substrings.add("last_layer")        # cancel_gradients_last_layer (modules/utils.py:144-149)
    optimizer.step()                                         # per-tensor clip + AdamW, fused
    ema_update(s_mod.arena, t_mod.arena, float(momentum))
    return loss.detach()


class GraphedTrainingStep:
    """`training_iteration` captured ONCE into a HIP graph and replayed: one hipGraphLaunch per iteration instead of ~700
    kernel launches driven from Python.  It is for the small-batch regime (64 images per GPU: the kernels of a step take less
    time than the host needs to launch them); at the headline batch of 256 the host already runs ahead of the GPU.

    What changes between iterations never enters the graph as a launch argument:
      * the batch is copied into the graph's static input tensors;
      * lr / weight decay / Adam bias corrections / frozen tensors: `FusedClipAdamW.stage_hyper` (an eager host -> device copy
        of the per-tensor table the captured AdamW kernel reads);
      * the EMA momentum and the DropPath seed: two device scalars the kernels read when they RUN (ccd_ema's d_m,
        ccd_droppath_scales' d_seed), staged the same way.  The seed sequence is the eager one's, so a graphed run draws the
        same masks as an eager run with the same torch seed;
      * the teacher temperature and the epoch < 30 branch ARE launch-time constants: a change re-captures (once per epoch
        during the temperature warm-up, once at epoch 30).
    The first `eager_steps` calls run eagerly (they are real iterations): lazily built tables (optimizer table, weight
    transposes) and the allocator's pools settle before anything is captured.  Single process only - the gradient all-reduce of
    parallel.DataParallel is launched from autograd hooks on a side stream and is not captured.  Bench-only (`bench.py --graph`):
    train.py runs the eager iteration, whose finite-loss check reads the loss every step."""

    def __init__(self, student, teacher, dino_loss: DINOLoss, optimizer: FusedClipAdamW, eager_steps=2):
        if hasattr(student, "finish_gradient_sync"):
            raise ValueError("GraphedTrainingStep captures a single-process step; run the data-parallel job eagerly")
        from .optim import HostStaging
        self.student, self.teacher, self.dino_loss, self.optimizer = student, teacher, dino_loss, optimizer
        dev = student.arena.device
        self._mom = HostStaging((2,), torch.float32, dev)
        self._seed = HostStaging((1,), torch.int64, dev)
        if int(eager_steps) < 1:     # the first capture relies on tables an eager step builds (optimizer table, weight transposes), and
            raise ValueError("GraphedTrainingStep needs at least one eager step before it captures")    # the DropPath seed draw syncs
        self.eager_left = int(eager_steps)
        self.key = self.graph = None
        self.captures = self.replays = 0

    def _key(self, images, masks, metrics, epoch):
        from . import engine
        fusion = tuple(sorted((k, v) for k, v in vars(engine.Fusion).items() if not k.startswith("_") and isinstance(v, (bool, int, type(None)))))
        return (tuple(images.shape), images.dtype, tuple(masks.shape), masks.dtype, tuple(metrics.shape), metrics.dtype,
                epoch < 30, float(self.dino_loss.teacher_temp_schedule[epoch]), bool(getattr(self.student, "training", True)), fusion)

    def _record(self, body):
        """-> (something with .replay(), DropPath seeds drawn by one pass of `body`).  Capturing launches nothing."""
        from . import engine
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        before = engine._DROPPATH_SEED["calls"]
        with torch.cuda.graph(g):
            body()
        return g, engine._DROPPATH_SEED["calls"] - before

    def _capture(self, images, masks, metrics, epoch):
        from . import engine
        self.graph = None                                    # the previous graph's pool goes back to the allocator first
        self.images, self.masks, self.metrics = images.clone(), masks.clone(), metrics.clone()
        seeds = engine._DROPPATH_SEED

        def body():
            loss = _forward_backward(self.student, self.teacher, self.dino_loss, self.optimizer, self.images, self.masks,
                                     self.metrics, epoch)
            self.optimizer.launch_step()
            ema_update(self.student.arena, self.teacher.arena, 0.0, d_m=self._mom.dev)
            self.loss = loss.detach()

        self._calls0 = seeds["calls"]
        engine.set_device_droppath_seed(self._seed.dev)
        try:
            self.graph, self._calls_per_step = self._record(body)
        finally:
            engine.set_device_droppath_seed(None)
        seeds["calls"] = self._calls0                        # capturing drew nothing: the first replay uses these seeds
        self.captures += 1

    def __call__(self, images, masks, metrics, epoch, lr, wd, momentum, freeze_last_layer=1):
        if self.eager_left > 0:
            self.eager_left -= 1
            return training_iteration(self.student, self.teacher, self.dino_loss, self.optimizer, images, masks, metrics,
                                      epoch, lr, wd, momentum, freeze_last_layer)
        from . import engine
        key = self._key(images, masks, metrics, epoch)
        if key != self.key:
            self._capture(images, masks, metrics, epoch)
            self.key = key
        self.images.copy_(images, non_blocking=True)
        self.masks.copy_(masks, non_blocking=True)
        self.metrics.copy_(metrics, non_blocking=True)
        h = self._mom.begin()
        h[0], h[1] = float(momentum), 1.0 - float(momentum)
        self._mom.commit()
        seeds = engine._DROPPATH_SEED
        delta = ((seeds["calls"] - self._calls0) * engine.DROPPATH_SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF
        self._seed.begin()[0] = delta - (1 << 64) if delta >= (1 << 63) else delta
        self._seed.commit()
        seeds["calls"] += self._calls_per_step
        _set_schedule(self.optimizer, lr, wd)
        if epoch < freeze_last_layer:
            self.student.arena.skip_substrings.add("last_layer")
        self.optimizer.stage_hyper()
        self.graph.replay()
        self.replays += 1
        return self.loss.clone()
