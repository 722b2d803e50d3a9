# names the reference pulls in through `from fastai.vision import *`
import json, math, os, random, warnings
from pathlib import Path
import numpy as np
import PIL
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import tensor
from torch.utils.data import Dataset
PathOrStr = object
def ifnone(a, b):
    return b if a is None else a
