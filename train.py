#!/usr/bin/env python3
"""CCD self-supervised pretraining on MI355X - same CLI / YAML surface as the reference's train.py:

    python -m torch.distributed.run --nproc-per-node=8 --master-addr 127.0.0.1 train.py --config Dino/configs/CCD_pretrain_ViT_small.yaml

Only --config is honoured, like the reference (train.py:394-396 returns Config(args.config)).  One process per GPU,
RCCL over xGMI (ccd_amd.parallel.DataParallel), checkpoints in the reference's layout
({student, teacher, optimizer, epoch, iteration, dino_loss}, `module.`-prefixed network keys, train.py:197-211).
Data: `dataset.scheme: synthetic` gives seeded in-memory batches; the LMDB pipeline of the reference
(Dino/dataset/*) is outside this implementation's scope.
"""
import argparse
import datetime
import json
import logging
import math
import os
import sys
import time
from pathlib import Path

import torch
import torch.nn as nn
import torch.utils.data

from Dino.loss.Dino_loss import DINOLoss
from Dino.model.dino_vision import ABIDINOModel
from Dino.modules import utils
from Dino.modules import vision_transformer as vits
from Dino.modules.segmentor import SegHead
from Dino.modules.vision_transformer import DINOHead
from Dino.utils.utils import Config, Logger
from ccd_amd import pretrain
from ccd_amd.parallel import DataParallel
from ccd_amd.synthetic import make_batch


class SyntheticPretrainSet(torch.utils.data.Dataset):
    """(image_views [3,3,32,128], mask [32,128], theta [3,3]) with the reference dataset's contract."""

    def __init__(self, length, seed=0, chunk=256):
        self.length, self.seed, self.chunk, self._cache = length, seed, chunk, (None, None)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        c = i // self.chunk
        if self._cache[0] != c:
            self._cache = (c, make_batch(self.chunk, seed=self.seed + c))
        imgs, masks, thetas = self._cache[1]
        j = i % self.chunk
        return imgs[j], masks[j], thetas[j]


class _ScalarLog:
    """TensorBoard when it is installed, JSON lines otherwise (same tags as train.py:280-290)."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb, self.fh = SummaryWriter(log_dir=log_dir), None
        except Exception:
            self.tb, self.fh = None, open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, scalar_value, global_step):
        if self.tb is not None:
            self.tb.add_scalar(tag=tag, scalar_value=scalar_value, global_step=global_step)
        else:
            self.fh.write(json.dumps({"tag": tag, "value": float(scalar_value), "step": int(global_step)}) + "\n")
            self.fh.flush()


def _lmdb_dirs(paths):
    """The reference's `_get_dataset` walk (train.py:414-421): a root either IS an LMDB environment or holds sub-trees of them."""
    out = []
    for p in paths:
        p = str(p)
        if os.path.isfile(os.path.join(p, "data.mdb")):
            out.append(p)
            continue
        if not os.path.isdir(p):
            raise FileNotFoundError(f"dataset root {p} does not exist")
        subs = sorted(f.path for f in os.scandir(p) if f.is_dir())
        if not subs:
            raise FileNotFoundError(f"dataset root {p} holds no LMDB environment")
        out.extend(_lmdb_dirs(subs))
    return out


def _get_databaunch(config):
    """-> (loader, view_maker): `view_maker` (device augmentation, ccd_amd/dataset) is None for synthetic batches, which already
    are (image_tensors, masks, metrics)."""
    world, rank = utils.get_world_size(), utils.get_rank()
    if config.dataset_scheme == "synthetic":
        n = int(config.dataset_synthetic_samples or 64 * config.batch_size_per_gpu * world)
        ds = SyntheticPretrainSet(n, seed=int(config.seed or 0))
        sampler = torch.utils.data.DistributedSampler(ds, shuffle=True)
        return torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=config.batch_size_per_gpu, num_workers=0,
                                           pin_memory=bool(config.dataset_pin_memory), drop_last=True), None
    if config.dataset_scheme != "selfsupervised_kmeans":
        raise NotImplementedError(f"dataset.scheme {config.dataset_scheme!r}: pretraining reads `selfsupervised_kmeans` LMDB "
                                  "datasets (or `synthetic` batches)")
    from ccd_amd.dataset import DeviceViewMaker, ImageDatasetSelfSupervisedKmeans, collate_uint8
    h, w = int(config.dataset_image_height or 32), int(config.dataset_image_width or 128)
    if (h, w) != (32, 128):
        raise ValueError(f"the character-region kernels are built for 32 x 128 inputs, not {h} x {w}")
    kwargs = dict(img_h=h, img_w=w, is_training=True, data_aug=bool(config.dataset_data_aug),
                  multiscales=bool(config.dataset_multiscales), data_portion=float(config.dataset_portion or 1.0),
                  mask=True, mask_path=config.dataset_mask_path or "",
                  augmentation_severity=int(config.dataset_augmentation_severity or 1))
    parts = [ImageDatasetSelfSupervisedKmeans(path=p, **kwargs) for p in _lmdb_dirs(config.dataset_train_roots)]
    ds = parts[0] if len(parts) == 1 else torch.utils.data.ConcatDataset(parts)
    sampler = torch.utils.data.DistributedSampler(ds, shuffle=True)
    workers = int(config.dataset_num_workers or 0)
    loader = torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=config.batch_size_per_gpu, num_workers=workers,
                                         collate_fn=collate_uint8, pin_memory=bool(config.dataset_pin_memory), drop_last=True,
                                         persistent_workers=workers > 0)
    views = DeviceViewMaker(img_h=h, img_w=w, severity=kwargs["augmentation_severity"], data_aug=kwargs["data_aug"],
                            seed=int(config.seed or 0) * 1000 + rank)
    return loader, views


def train(config):
    utils.init_distributed_mode(config)
    utils.fix_random_seeds(config.seed)
    logging.info("Construct dataset.")
    loader, view_maker = _get_databaunch(config)
    config.iter_num = len(loader)
    world = utils.get_world_size()

    # ---- student / teacher (same construction order as the reference -> same initial weights for a seed)
    config.arch = config.arch.replace("deit", "vit")
    if config.arch not in vits.__dict__:
        raise ValueError(f"Unknow architecture: {config.arch}")
    student_b = vits.__dict__[config.arch](patch_size=config.patch_size, drop_path_rate=config.drop_path_rate)
    teacher_b = vits.__dict__[config.arch](patch_size=config.patch_size)
    embed_dim = student_b.embed_dim
    student = ABIDINOModel(student_b, SegHead(in_channels=config.model_seg_channel, mla_channels=128, mlahead_channels=64,
                                              num_classes=2),
                           DINOHead(embed_dim, config.out_dim, use_bn=config.use_bn_in_head,
                                    norm_last_layer=config.norm_last_layer))
    teacher = ABIDINOModel(teacher_b, None, DINOHead(embed_dim, config.out_dim, config.use_bn_in_head))
    student, teacher = student.cuda(), teacher.cuda()
    if utils.has_batchnorms(student) and world > 1:
        student = nn.SyncBatchNorm.convert_sync_batchnorm(student)
    student.ensure_arena()
    teacher.ensure_arena()
    student = DataParallel(student, device_ids=[config.gpu], find_unused_parameters=True)
    teacher = DataParallel(teacher, device_ids=[config.gpu])
    teacher.module.backbone.load_state_dict(student.module.backbone.state_dict())
    teacher.module.head.load_state_dict(student.module.head.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    print(f"Student and Teacher are built: they are both {config.arch} network.")

    global_bs = config.batch_size_per_gpu * world
    config.epochs = int(config.training_epochs * len(loader) * global_bs / config.imgnet_based) + 1
    print(f"training epochs is {config.epochs}")
    dino_loss = DINOLoss(config.out_dim, config.crops_number, config.warmup_teacher_temp, config.teacher_temp,
                         config.warmup_teacher_temp_epochs, config.epochs).cuda()
    if config.optimizer != "adamw":
        raise NotImplementedError("the shipped pretraining configs use adamw; only the fused AdamW is implemented")
    if config.use_fp16:
        raise NotImplementedError("use_fp16 is False in every shipped config; this implementation runs bf16 MFMA "
                                  "operands with fp32 master weights and needs no loss scaler")
    optimizer = pretrain.make_optimizer(student.module, clip_grad=config.clip_grad or 0.0)

    niter = config.training_epochs * len(loader)
    lr_schedule = utils.cosine_iter_scheduler(config.lr * global_bs / 256., config.min_lr, niter,
                                              warmup_iters=min(niter, int(config.warmup_epoch * config.imgnet_based / global_bs)))
    wd_schedule = utils.cosine_iter_scheduler(config.weight_decay, config.weight_decay_end, niter)
    momentum_schedule = utils.cosine_iter_scheduler(config.momentum_teacher, 1, niter)
    print("Loss, optimizer and schedulers ready.")

    to_restore = {"epoch": 0, "iteration": 0}
    ckpt_dir = os.path.join(config.output_dir, config.global_name)
    utils.restart_from_checkpoint(os.path.join(ckpt_dir, "checkpoint.pth"), run_variables=to_restore, student=student,
                                  teacher=teacher, optimizer=optimizer, dino_loss=dino_loss)
    student.module.ensure_arena()
    teacher.module.ensure_arena()
    iteration, epoch = int(to_restore["iteration"]), to_restore["epoch"]
    print(f"continue to train:{iteration}:{epoch}")
    start_time, global_epoch = time.time(), 0
    pending_loss, loss_sum, loss_n = None, None, 0

    print("Starting DINO training !")
    for train_epoch in range(config.training_epochs):
        loader.sampler.set_epoch(train_epoch)
        metric_logger = utils.MetricLogger(delimiter="  ")
        header = "Epoch: [{}/{}]".format(train_epoch, config.training_epochs)
        for batch in metric_logger.log_every(loader, 10, header):
            if iteration >= niter:
                break
            # LMDB batches arrive as uint8 samples + masks: the three views and theta are made on the device
            image_tensors, masks, metrics = batch if view_maker is None else view_maker(*batch)
            epoch = int((iteration + 1) * global_bs / config.imgnet_based)
            if epoch != global_epoch:          # pseudo-epoch boundary: sync meters, checkpoint, log.txt
                global_epoch = epoch
                # the loss of the last enqueued step is still unverified: never persist weights a NaN step has touched
                if pending_loss is not None and not math.isfinite(pending_loss.item()):
                    print("Loss is {}, stopping training".format(pending_loss.item()), force=True)
                    sys.exit(1)
                if loss_sum is not None:       # the meters see EVERY iteration's loss (summed on the device, read here)
                    metric_logger.update(loss=(loss_sum / loss_n).item())
                    loss_sum, loss_n = None, 0
                metric_logger.synchronize_between_processes()
                print("Averaged stats:", metric_logger)
                stats = {k: m.global_avg for k, m in metric_logger.meters.items()}
                save_dict = {"student": student.state_dict(), "teacher": teacher.state_dict(),
                             "optimizer": optimizer.state_dict(), "epoch": epoch, "iteration": iteration,
                             "dino_loss": dino_loss.state_dict()}
                os.makedirs(ckpt_dir, exist_ok=True)
                utils.save_on_master(save_dict, os.path.join(ckpt_dir, "checkpoint.pth"))
                if config.saveckp_freq and epoch % config.saveckp_freq == 0:
                    utils.save_on_master(save_dict, os.path.join(ckpt_dir, f"checkpoint{epoch:04}.pth"))
                if utils.is_main_process():
                    with (Path(ckpt_dir) / "log.txt").open("a") as f:
                        f.write(json.dumps({**{f"train_{k}": v for k, v in stats.items()}, "epoch": epoch}) + "\n")
                metric_logger = utils.MetricLogger(delimiter="  ")
            image_tensors = image_tensors.cuda(non_blocking=True)
            masks = masks.cuda(non_blocking=True)
            metrics = metrics.cuda(non_blocking=True)
            loss = pretrain.training_iteration(student, teacher, dino_loss, optimizer, image_tensors, masks, metrics,
                                               epoch, lr_schedule[iteration], wd_schedule[iteration],
                                               momentum_schedule[iteration], freeze_last_layer=config.freeze_last_layer)
            # finiteness is checked one iteration late so the host never waits on the step it has just enqueued
            if pending_loss is not None and not math.isfinite(pending_loss.item()):
                print("Loss is {}, stopping training".format(pending_loss.item()), force=True)
                sys.exit(1)
            pending_loss = loss
            loss_sum = loss.detach() if loss_sum is None else loss_sum + loss.detach()
            loss_n += 1
            if iteration % 10 == 0:            # one host read per print interval: the mean of the iterations since the last one
                metric_logger.update(loss=(loss_sum / loss_n).item())
                loss_sum, loss_n = None, 0
                metric_logger.update(lr=optimizer.param_groups[0]["lr"])
                metric_logger.update(wd=optimizer.param_groups[0]["weight_decay"])
            if iteration % config.training_show_iters == 0 and config.writer is not None:
                for name, val in dino_loss.last_losses.items():
                    config.writer.add_scalar("metric/" + name, val.item(), iteration)
                config.writer.add_scalar("metric/lr", optimizer.param_groups[0]["lr"], iteration)
                config.writer.add_scalar("metric/wd", optimizer.param_groups[0]["weight_decay"], iteration)
            iteration += 1
        if iteration >= niter:
            break
    torch.cuda.synchronize()
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))


def _parse_arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True, help="path to config file")
    parser.add_argument("--local_rank", "--local-rank", default=0, type=int, help="set by the launcher; ignored")
    args, _ = parser.parse_known_args()       # every other reference flag is accepted and ignored, as upstream
    return Config(args.config)


if __name__ == "__main__":
    config = _parse_arguments()
    Logger.init(config.global_workdir, config.global_name, config.global_phase)
    Logger.enable_file()
    logging.info(config)
    os.makedirs(f"./saved_models/{config.global_name}", exist_ok=True)
    config.writer = _ScalarLog(f"./tensorboard/{config.global_name}") if int(os.environ.get("RANK", 0)) == 0 else None
    train(config)
