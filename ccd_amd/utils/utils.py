"""Config / Logger with the reference's behaviour (Dino/utils/utils.py: Logger 160-188, Config 191-237):
a YAML file is laid over Dino/configs/template.yaml and nested keys are flattened to `a_b_c` attributes; an unknown
attribute resolves to the dict of all attributes sharing that prefix, or None."""
from __future__ import annotations

import logging
import os
import time

import yaml

_TEMPLATE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "Dino", "configs",
                         "template.yaml")
_PHASES = ("train", "test")
_STAGES = ("pretrain-vision", "pretrain-language", "pretrain-fusion", "train-semi-supervised", "train-supervised")


class Config(object):
    def __init__(self, config_path, host=True):
        assert os.path.exists(config_path), "%s does not exists!" % config_path
        template = "Dino/configs/template.yaml" if os.path.exists("Dino/configs/template.yaml") else _TEMPLATE
        for path in (template, config_path):
            with open(path) as fh:
                self._absorb(yaml.load(fh, Loader=yaml.FullLoader) or {})
        self.global_workdir = os.path.join(self.global_workdir, self.global_name)

    def _absorb(self, tree, prefix=""):
        for key, value in tree.items():
            if isinstance(value, dict):
                self._absorb(value, f"{prefix}{key}_")
                continue
            if key == "phase":
                assert value in _PHASES
            if key == "stage":
                assert value in _STAGES
            setattr(self, f"{prefix}{key}", value)

    def __getattr__(self, item):
        # only reached when normal lookup fails: collect `item_*` attributes into a dict
        if item.startswith("__"):
            raise AttributeError(item)
        prefix = f"{item}_"
        found = {k[len(prefix):]: v for k, v in self.__dict__.items() if k.startswith(prefix)}
        return found or None

    def __repr__(self):
        rows = [f"\t({i}): {k} = {v}" for i, (k, v) in enumerate(sorted(vars(self).items()))]
        return "ModelConfig(\n" + "\n".join(rows) + "\n)"


class Logger(object):
    _handle, _root = None, None

    @staticmethod
    def init(output_dir, name, phase):
        fmt = "[%(asctime)s %(filename)s:%(lineno)d %(levelname)s {}] %(message)s".format(name)
        logging.basicConfig(level=logging.INFO, format=fmt)
        os.makedirs(output_dir, exist_ok=True)
        Logger._handle = logging.FileHandler(os.path.join(output_dir, f"{phase}.txt"))
        Logger._root = logging.getLogger()

    @staticmethod
    def enable_file():
        if Logger._handle is None or Logger._root is None:
            raise Exception("Invoke Logger.init() first!")
        Logger._root.addHandler(Logger._handle)

    @staticmethod
    def disable_file():
        if Logger._handle is None or Logger._root is None:
            raise Exception("Invoke Logger.init() first!")
        Logger._root.removeHandler(Logger._handle)
