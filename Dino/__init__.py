"""`Dino.*` import paths of the reference (TongkunGuan/CCD) mapped onto the MI355X-native implementation in
`ccd_amd`, so `from Dino.modules import vision_transformer as vits`, `from Dino.model.dino_vision import
ABIDINOModel`, `from Dino.loss.Dino_loss import DINOLoss`, `from Dino.utils.utils import Config` keep working.
The pretraining path and the finetune path (8f row 1: DINO_Finetune, NRTRDecoder, TFLoss,
AttnConvertor) exist here."""
import importlib
import sys

_ALIASES = {
    "Dino.modules": "ccd_amd.modules",
    "Dino.modules.vision_transformer": "ccd_amd.modules.vision_transformer",
    "Dino.modules.segmentor": "ccd_amd.modules.segmentor",
    "Dino.modules.utils": "ccd_amd.modules.utils",
    "Dino.model": "ccd_amd.model",
    "Dino.model.dino_vision": "ccd_amd.model.dino_vision",
    "Dino.loss": "ccd_amd.loss",
    "Dino.loss.Dino_loss": "ccd_amd.loss.Dino_loss",
    "Dino.loss.ce_loss": "ccd_amd.loss.ce_loss",
    "Dino.decoder": "ccd_amd.decoder",
    "Dino.decoder.nrtr_decoder": "ccd_amd.decoder.nrtr_decoder",
    "Dino.convertor": "ccd_amd.convertor",
    "Dino.convertor.attn": "ccd_amd.convertor.attn",
    "Dino.metric": "ccd_amd.metric",
    "Dino.metric.eval_acc": "ccd_amd.metric.eval_acc",
    "Dino.dataset": "ccd_amd.dataset",
    "Dino.dataset.dataset_pretrain": "ccd_amd.dataset.dataset_pretrain",
    "Dino.dataset.datasetsupervised_kmeans": "ccd_amd.dataset.datasetsupervised_kmeans",
    "Dino.utils": "ccd_amd.utils",
    "Dino.utils.utils": "ccd_amd.utils.utils",
}
for _alias, _target in _ALIASES.items():
    _mod = importlib.import_module(_target)
    sys.modules[_alias] = _mod
    _parent, _, _leaf = _alias.rpartition(".")
    setattr(sys.modules[_parent], _leaf, _mod)
