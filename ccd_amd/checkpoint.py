"""Checkpoint interop with the reference: read the published CCD files into this implementation and
write them back in the same layout.

Two layouts exist in the reference:
* pretraining (train.py:255-262, the "CCD-ViT-Small / CCD-ViT-Base" downloads of README.md:46,53):
  {'student': sd, 'teacher': sd, 'optimizer', 'epoch', 'iteration', 'dino_loss'} - both state dicts DDP-prefixed
  (`module.backbone.blocks.0.attn.qkv.weight`, ...);
* finetuning (train_finetune.py:382-388, the "ARD / STD" downloads): {'net': sd, 'optimizer', 'iteration'} -
  nn.DataParallel-prefixed (`module.backbone...`, `module.decoder...`).

    python -m ccd_amd.checkpoint inspect  checkpoint.pth
    python -m ccd_amd.checkpoint convert  checkpoint.pth out.pth [--strip-prefix | --add-prefix] [--only teacher]

`describe()` recovers what is needed to rebuild the networks (architecture from the embedding width, DINO head width,
decoder depth), `build_pretrain_models()` / `build_finetune_model()` rebuild them and load strictly (GPU needed: the modules
have no CPU path), `export_*()` write the reference layout back.
"""
from __future__ import annotations

import argparse
import re

import torch

ARCH_BY_WIDTH = {192: "vit_tiny", 384: "vit_small", 512: "vit_base"}          # vision_transformer.py:254-280 of the reference


def _strip(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def _prefixed(sd):
    return {(k if k.startswith("module.") else "module." + k): v for k, v in sd.items()}


def _net_facts(sd):
    sd = _strip(sd)
    facts = {"tensors": len(sd), "parameters": int(sum(v.numel() for v in sd.values() if torch.is_tensor(v)))}
    pe = sd.get("backbone.pos_embed")
    if pe is not None:
        facts["embed_dim"] = int(pe.shape[-1])
        facts["arch"] = ARCH_BY_WIDTH.get(int(pe.shape[-1]), f"unknown (width {int(pe.shape[-1])})")
    blocks = [int(m.group(1)) for k in sd for m in [re.match(r"backbone\.blocks\.(\d+)\.", k)] if m]
    if blocks:
        facts["depth"] = max(blocks) + 1
    if "backbone.patch_embed.proj.weight" in sd:
        facts["patch_size"] = int(sd["backbone.patch_embed.proj.weight"].shape[-1])
    if "head.last_layer.weight_v" in sd:
        facts["out_dim"] = int(sd["head.last_layer.weight_v"].shape[0])
        facts["head_bn"] = any(k.startswith("head.mlp.") and "running_mean" in k for k in sd)
    facts["has_segmentation"] = any(k.startswith("segmentation.") for k in sd)
    layers = [int(m.group(1)) for k in sd for m in [re.match(r"decoder\.layer_stack\.(\d+)\.", k)] if m]
    if layers:
        facts["decoder_layers"] = max(layers) + 1
    return facts


def describe(ckpt):
    """ckpt: the object torch.load returned (or a path)."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    if not isinstance(ckpt, dict):
        raise ValueError("not a CCD checkpoint: expected a dict")
    if "student" in ckpt and "teacher" in ckpt:
        kind, nets = "pretrain", {"student": ckpt["student"], "teacher": ckpt["teacher"]}
    elif "net" in ckpt:
        kind, nets = "finetune", {"net": ckpt["net"]}
    else:
        raise ValueError(f"not a CCD checkpoint: keys {sorted(ckpt)[:8]} hold neither student/teacher nor net")
    out = {"kind": kind, "extra": sorted(k for k in ckpt if k not in nets)}
    for name, sd in nets.items():
        out[name] = dict(_net_facts(sd), ddp_prefixed=all(k.startswith("module.") for k in sd))
    for k in ("epoch", "iteration"):
        if k in ckpt:
            out[k] = int(ckpt[k])
    return out


def build_pretrain_models(ckpt, device="cuda"):
    """-> (student, teacher): DataParallel-wrapped ABIDINOModels holding the checkpoint's weights (strict load)."""
    from .model.dino_vision import ABIDINOModel
    from .modules import vision_transformer as vits
    from .modules.segmentor import SegHead
    from .parallel import DataParallel
    info = describe(ckpt)
    assert info["kind"] == "pretrain", "a {student, teacher} checkpoint is needed"
    s, t = info["student"], info["teacher"]
    arch, E = s["arch"], s["embed_dim"]
    if arch not in vits.__dict__:
        raise ValueError(f"embedding width {E}: no architecture of that width (have {sorted(ARCH_BY_WIDTH.values())})")
    student = ABIDINOModel(vits.__dict__[arch](patch_size=s["patch_size"]),
                           SegHead(in_channels=E, mla_channels=128, mlahead_channels=64, num_classes=2),
                           vits.DINOHead(E, s["out_dim"], use_bn=s["head_bn"], norm_last_layer=False))
    teacher = ABIDINOModel(vits.__dict__[arch](patch_size=t["patch_size"]), None, vits.DINOHead(E, t["out_dim"], t["head_bn"]))
    nets = []
    for net, key in ((student, "student"), (teacher, "teacher")):
        net = DataParallel(net.to(device))
        net.load_state_dict(_prefixed(ckpt[key]), strict=True)
        net.module.ensure_arena()
        nets.append(net)
    return tuple(nets)


def build_finetune_model(ckpt, config=None, device="cuda"):
    """-> DataParallel(DINO_Finetune) with `ckpt['net']` loaded strictly; `config` defaults to the architecture the tensors imply."""
    from .finetune import FinetuneConfig
    from .model.dino_vision import DINO_Finetune
    from .parallel import DataParallel
    info = describe(ckpt)
    assert info["kind"] == "finetune", "a {net, ...} checkpoint is needed"
    n = info["net"]
    if config is None:
        config = FinetuneConfig(arch=n["arch"], decoder_n_layers=n.get("decoder_layers", 6))
    model = DataParallel(DINO_Finetune(config).to(device))
    model.load_state_dict(_prefixed(ckpt["net"]), strict=True)
    model.module.ensure_arena()
    return model


def export_pretrain(student, teacher, path, **extra):
    """train.py:255-262's layout (student / teacher state dicts keep the `module.` prefix whatever wrapper holds them)."""
    torch.save(dict({"student": _prefixed(student.state_dict()), "teacher": _prefixed(teacher.state_dict())}, **extra), path)


def export_finetune(model, path, **extra):
    torch.save(dict({"net": _prefixed(model.state_dict())}, **extra), path)


def convert(src, dst, prefix=None, only=None):
    ckpt = torch.load(src, map_location="cpu", weights_only=False)
    info = describe(ckpt)
    fix = {None: lambda sd: sd, "strip": _strip, "add": _prefixed}[prefix]
    names = ("student", "teacher") if info["kind"] == "pretrain" else ("net",)
    if only is not None:
        if only not in names:
            raise ValueError(f"--only {only}: this checkpoint holds {names}")
        out = fix(ckpt[only])                                      # a bare state dict (what model.load_state_dict takes)
    else:
        out = dict(ckpt)
        for n in names:
            out[n] = fix(ckpt[n])
    torch.save(out, dst)
    return info


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ccd_amd.checkpoint", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("inspect"); a.add_argument("path")
    c = sub.add_parser("convert"); c.add_argument("src"); c.add_argument("dst")
    g = c.add_mutually_exclusive_group()
    g.add_argument("--strip-prefix", action="store_true"); g.add_argument("--add-prefix", action="store_true")
    c.add_argument("--only", default=None)
    args = ap.parse_args(argv)
    if args.cmd == "inspect":
        import json
        print(json.dumps(describe(args.path), indent=2))
    else:
        convert(args.src, args.dst, "strip" if args.strip_prefix else "add" if args.add_prefix else None, args.only)


if __name__ == "__main__":
    main()
