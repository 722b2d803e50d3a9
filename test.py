#!/usr/bin/env python3
"""Benchmark evaluation of a finetuned recogniser on MI355X - the reference's test.py (CLI :30-90, loop :156-218):

    python test.py --config Dino/configs/CCD_vision_model_ARD.yaml        # model.checkpoint = a {net, ...} file

Builds DINO_Finetune(config), loads `checkpoint['net']` (DataParallel-prefixed keys, as train_finetune.py and the
published ARD / STD files store them), evaluates every entry of `dataset.test.roots` with TextAccuracy and prints the
reference's report: one line per dataset (word_num, accuracy = cwr) and the word-weighted total.
"""
import argparse
import logging
import os

import torch
import torch.utils.data

from Dino.metric.eval_acc import TextAccuracy
from Dino.model.dino_vision import DINO_Finetune
from Dino.modules import utils
from Dino.utils.utils import Config, Logger
from ccd_amd.dataset.dataset_pretrain import ImageDataset, collate_fn_filter_none
from ccd_amd.parallel import DataParallel
from train import _lmdb_dirs

EVAL_DATA_NAMES = ["IIIT5k_3000", "SVT", "IC13_1015", "IC15_1811", "SVTP", "CUTE80", "TotalText", "COCOText", "CTW", "HOST",
                   "WOST"]                                        # test.py:184-196 (the order of dataset.test.roots)


def get_test_loaders(config):
    """One loader per entry of dataset.test.roots; an entry with sub-folders is the concatenation of its LMDBs (test.py:111-121)."""
    kw = dict(img_h=int(config.dataset_image_height or 32), img_w=int(config.dataset_image_width or 128),
              max_length=int(config.decoder_max_seq_len or 25), type=config.dataset_charset_type or "DICT90", is_training=False)
    loaders = []
    for eval_root in config.dataset_test_roots:
        parts = [ImageDataset(path=p, **kw) for p in _lmdb_dirs([eval_root])]
        ds = parts[0] if len(parts) == 1 else torch.utils.data.ConcatDataset(parts)
        loaders.append(torch.utils.data.DataLoader(ds, batch_size=int(config.dataset_test_batch_size or 256), shuffle=False,
                                                   num_workers=int(config.dataset_num_workers or 0),
                                                   collate_fn=collate_fn_filter_none, pin_memory=bool(config.dataset_pin_memory),
                                                   drop_last=False))
    return loaders


def evaluate(model, loaders, config, names=None):
    names = names or EVAL_DATA_NAMES
    words = acc = 0.0
    report, results = "", []
    model.eval() if not hasattr(model, "module") else model.module.eval()
    with torch.no_grad():
        for i, loader in enumerate(loaders):
            metric = TextAccuracy(charset_path=config.dataset_charset_path, case_sensitive=bool(config.dataset_eval_case_sensitive),
                                  model_eval="vision")
            res = metric.compute(model, loader)
            results.append(res)
            acc += res["cwr"] * res["words"]
            words += res["words"]
            name = names[i] if i < len(names) else f"dataset{i}"
            report += f"dataset: {name} --> word_num: {res['words']} --> accuracy: {res['cwr']:0.3f}\n"
    report += f"total_accuracy: {acc / max(words, 1.0):0.3f}"
    return report, results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="Dino/configs/CCD_vision_model_ARD.yaml", help="path to config file")
    ap.add_argument("--checkpoint", type=str, default=None)
    ap.add_argument("--test_root", type=str, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    a = ap.parse_args()
    config = Config(a.config)
    if a.checkpoint is not None:
        config.model_checkpoint = a.checkpoint
    if a.test_root is not None:
        config.dataset_test_roots = [a.test_root]
    if a.batch_size is not None:
        config.dataset_test_batch_size = a.batch_size
    Logger.init(config.global_workdir, config.global_name, "test")
    utils.fix_random_seeds(int(config.global_seed or 0))
    logging.info("Construct dataset.")
    loaders = get_test_loaders(config)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    model = DINO_Finetune(config).to(device)
    model.ensure_arena()
    model = DataParallel(model)
    if config.model_checkpoint:
        logging.info(f"Read vision model from {config.model_checkpoint}.")
        sd = torch.load(config.model_checkpoint, map_location="cpu", weights_only=False)
        model.load_state_dict(sd["net"])
        model.module.ensure_arena()
    logging.info("eval model")
    report, _ = evaluate(model, loaders, config)
    print("-" * 80)
    print(report + "\n")


if __name__ == "__main__":
    main()
