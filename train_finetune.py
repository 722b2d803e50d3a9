#!/usr/bin/env python3
"""CCD finetune (text recognition) on MI355X - same CLI / YAML surface as the reference's train_finetune.py:

    python train_finetune.py --config Dino/configs/CCD_vision_model_ARD.yaml
    python -m torch.distributed.run --nproc-per-node=8 --master-addr 127.0.0.1 train_finetune.py --config ...

The reference wraps the model in nn.DataParallel (one process, train_finetune.py:186); here it is one process per GPU
with gradient averaging over RCCL (ccd_amd.parallel.DataParallel) - state-dict keys keep the
`module.` prefix either way, so `pretrain_checkpoint` (the teacher of a pretraining run, :190-198) and `checkpoint`
({net, optimizer, iteration}, :199-207, 382-388) load and save in the reference's layout.
Data: `dataset.scheme: supervised` reads the labelled LMDB datasets (ccd_amd/dataset/dataset_pretrain.py), `synthetic` gives
seeded labelled batches; benchmark evaluation is test.py (TextAccuracy, ccd_amd/metric/eval_acc.py).
"""
import argparse
import logging
import os
import time

import torch
import torch.distributed as dist
import torch.utils.data

from Dino.model.dino_vision import DINO_Finetune
from Dino.modules import utils
from Dino.utils.utils import Config, Logger
from ccd_amd import finetune as ft
from ccd_amd.parallel import DataParallel


class SyntheticLabelledSet(torch.utils.data.Dataset):
    """(image [3,32,128] fp32, padded label indices int64 [1, T]) with the contract of ImageDataset + AttnConvertor
    (Dino/dataset/dataset_pretrain.py:218-226): random text of 3..15 DICT90 characters per sample."""

    def __init__(self, length, convertor, seed=0):
        self.length, self.convertor, self.seed = length, convertor, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        n = int(torch.randint(3, 16, (1,), generator=g))
        word = "".join(self.convertor.idx2char[int(c)] for c in torch.randint(0, 90, (n,), generator=g))
        return torch.randn(3, 32, 128, generator=g), self.convertor.str2tensor([word])


def _loader(config, convertor, world, rank):
    bs = int(config.dataset_train_batch_size)
    if config.dataset_scheme == "supervised":
        # labelled LMDB datasets (Dino/dataset/dataset_pretrain.py:18-277; roots walked like train_finetune.py:96-121)
        from ccd_amd.dataset.dataset_pretrain import ImageDataset, collate_fn_filter_none
        from train import _lmdb_dirs
        kw = dict(img_h=int(config.dataset_image_height or 32), img_w=int(config.dataset_image_width or 128),
                  max_length=int(config.decoder_max_seq_len or 25), type=config.dataset_charset_type or "DICT90",
                  data_portion=float(config.dataset_portion or 1.0), is_training=True,
                  data_aug=bool(config.dataset_data_aug))             # the YAML's `data_aug` (dataset_pretrain.py:68-158)
        parts = [ImageDataset(path=p, **kw) for p in _lmdb_dirs(config.dataset_train_roots)]
        ds = parts[0] if len(parts) == 1 else torch.utils.data.ConcatDataset(parts)
        sampler = torch.utils.data.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None
        return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=sampler is None, sampler=sampler,
                                           num_workers=int(config.dataset_num_workers or 0), collate_fn=collate_fn_filter_none,
                                           pin_memory=bool(config.dataset_pin_memory), drop_last=True)
    if config.dataset_scheme != "synthetic":
        raise NotImplementedError(f"dataset.scheme {config.dataset_scheme!r}: finetuning reads `supervised` LMDB datasets "
                                  "(or `synthetic` labelled batches)")
    n = int(config.dataset_synthetic_samples or 32 * bs * world)
    ds = SyntheticLabelledSet(n, convertor, seed=int(config.seed or 0))
    sampler = torch.utils.data.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None
    return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=sampler is None, sampler=sampler, num_workers=0,
                                       pin_memory=bool(config.dataset_pin_memory), drop_last=True)


def word_accuracy(model, images, labels):
    """Greedy decoding accuracy on one batch (the reference evaluates with TextAccuracy on LMDB benchmarks, eval_acc.py)."""
    mod = model.module if hasattr(model, "module") else model
    mod.eval()
    with torch.no_grad():
        probs = model(images, None, return_loss=False)
    mod.train()
    idx, _ = mod.label_convertor.tensor2idx(torch.log(probs.clamp_min(1e-30)))
    truth = []
    for row in labels.tolist():
        body = row[1:]
        truth.append(body[:body.index(mod.label_convertor.end_idx)] if mod.label_convertor.end_idx in body else body)
    return sum(int(a == b) for a, b in zip(idx, truth)) / max(1, len(truth))


def main(config):
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    utils.fix_random_seeds(int(config.global_seed or config.seed or 0))
    model = DINO_Finetune(config).to(device)
    model.ensure_arena()
    model = DataParallel(model)                          # `module.` key prefix like nn.DataParallel; reducer when world > 1
    model.train()
    if config.model_pretrain_checkpoint and os.path.isfile(config.model_pretrain_checkpoint):
        logging.info(f"Read pretrain vision model from {config.model_pretrain_checkpoint}.")
        teacher = torch.load(config.model_pretrain_checkpoint, map_location="cpu", weights_only=False)["teacher"]
        dd = model.state_dict()
        missing = [n for n in dd if n not in teacher]
        model.load_state_dict({n: teacher.get(n, v) for n, v in dd.items()})
        logging.info(f"not in the pretraining checkpoint (kept at init): {len(missing)} tensors")
    if config.model_checkpoint and os.path.isfile(config.model_checkpoint):
        logging.info(f"Read vision model from {config.model_checkpoint}.")
        model.load_state_dict(torch.load(config.model_checkpoint, map_location="cpu", weights_only=False)["net"])
    model.module.ensure_arena()
    loader = _loader(config, model.module.label_convertor, world, rank)
    config.iter_num = len(loader)
    if config.optimizer != "adamw":
        raise NotImplementedError("the shipped finetune configs use adamw; only the fused AdamW is implemented")
    optimizer = ft.make_optimizer(model, lr=config.lr, weight_decay=config.weight_decay, clip_grad=config.clip_grad)
    lr_schedule = utils.cosine_scheduler(config.lr, config.min_lr, config.training_epochs, len(loader),
                                         warmup_epochs=config.warmup_epochs)
    to_restore = {"iteration": 0}
    if config.model_checkpoint:
        utils.restart_from_checkpoint(config.model_checkpoint, run_variables=to_restore, optimizer=optimizer)
    iteration = int(to_restore["iteration"])
    print(f"continue to train:{iteration}")
    out_dir = os.path.join(config.output_dir or "./saved_models/", config.global_name)
    os.makedirs(out_dir, exist_ok=True)
    total, it, t0, running, nrun = int(config.training_epochs * len(loader)), iter(loader), time.time(), None, 0
    augmenter = None
    while iteration < total:
        try:
            images, labels = next(it)
        except StopIteration:
            if loader.sampler is not None and hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(iteration // len(loader))
            it = iter(loader)
            images, labels = next(it)
        if images.dtype == torch.uint8:         # dataset.data_aug: resized uint8 samples, the imgaug-shaped pipeline runs on the device
            if augmenter is None:
                from ccd_amd.dataset.dataset_pretrain import DeviceImageAugmenter
                augmenter = DeviceImageAugmenter(images.shape[1], images.shape[2], seed=int(config.seed or 0) + rank, device=device)
            images = augmenter(images)
        else:
            images = images.to(device, non_blocking=True)
        labels = labels.squeeze(1).to(device, non_blocking=True)
        loss, attn = ft.training_iteration(model, optimizer, images, labels, lr_schedule[iteration])
        running = loss if running is None else running + loss
        nrun += 1
        if iteration % config.training_show_iters == 0:
            avg = (running / nrun).item()
            logging.info(f"iteration:{iteration}--> train loss:{avg}")
            if config.writer is not None:
                config.writer.add_scalar("metric/train_loss", avg, iteration)
                config.writer.add_scalar("metric/lr", optimizer.param_groups[0]["lr"], iteration)
            running, nrun = None, 0
        if iteration % config.training_eval_iters == 0:
            acc = word_accuracy(model, images, labels)
            logging.info(f"iteration: {iteration} synthetic-batch word accuracy: {acc:0.3f} "
                         f"({(time.time() - t0):.0f} s elapsed)")
            if config.writer is not None:
                config.writer.add_scalar("metric/eval_acc", acc, iteration)
        if iteration % config.training_save_iters == 0 and rank == 0:
            torch.save({"net": model.state_dict(), "optimizer": optimizer.state_dict(), "iteration": iteration},
                       os.path.join(out_dir, f"{iteration}.pth"))
        iteration += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.destroy_process_group()


def _parse_arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True, help="path to config file")
    parser.add_argument("--local_rank", "--local-rank", default=0, type=int, help="set by the launcher; ignored")
    args, _ = parser.parse_known_args()
    return Config(args.config)


if __name__ == "__main__":
    from train import _ScalarLog
    config = _parse_arguments()
    Logger.init(config.global_workdir, config.global_name, config.global_phase)
    Logger.enable_file()
    logging.info(config)
    config.writer = _ScalarLog(f"./tensorboard/{config.global_name}") if int(os.environ.get("RANK", 0)) == 0 else None
    main(config)
