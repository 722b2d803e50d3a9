"""ABIDINOModel: dual-view backbone + segmentation head + character-region pooling + DINO head
(Dino/model/dino_vision.py:21-115), MI355X-native.

Differences from the reference that are invisible at the boundary:
  * the character clusters live on the device as uint8 id maps (ccd_amd.engine.Selection) - the host-side
    numpy/skimage labelling loop (dino_vision.py:59-71) is the HIP kernel ccl_label_kernel;
  * the number of selected character rows M never reaches the host: 'instances_view' is backed by a worst-case
    buffer and only materialises a [2M, K] tensor if somebody indexes the output dict for it.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import engine, ops
from ..modules.vision_transformer import ArenaModule


class ClusterMaps:
    """What the student hands to the teacher as `clusters=` (reference: a dense [2B,26,32,128] tensor)."""

    def __init__(self, selection: engine.Selection):
        self.selection = selection

    def dense(self):
        return self.selection.dense()

    # a little tensor-likeness for callers that only look at the shape / move it around
    @property
    def shape(self):
        return torch.Size((2 * self.selection.batch, 26, 32, 128))

    def to(self, *a, **k):
        return self


class ModelOutput(dict):
    """Output dict whose reference-shaped entries are produced lazily (they need a host sync on M)."""

    def __init__(self, eager, lazy):
        super().__init__(eager)
        self._lazy = lazy

    def __getitem__(self, key):
        if key not in self.keys() and key in self._lazy:
            super().__setitem__(key, self._lazy[key]())
        return super().__getitem__(key)

    def __contains__(self, key):
        return super().__contains__(key) or key in self._lazy

    def raw(self, key, default=None):
        return super().get(key, default)


class ABIDINOModel(ArenaModule):
    def __init__(self, backbone, Segmentation, head):
        super().__init__()
        backbone.fc, backbone.head = nn.Identity(), nn.Identity()
        self.backbone = backbone
        self.segmentation = Segmentation
        self.head = head

    # ------------------------------------------------------------------------------------------- arena
    def _transposed_names(self):
        return ["backbone." + n for n in self.backbone._transposed_names()] + \
               ["head." + n for n in self.head._transposed_names()]

    def attach_arena(self, arena, prefix):
        super().attach_arena(arena, prefix)
        self.backbone.attach_arena(arena, prefix + "backbone.")
        self.head.attach_arena(arena, prefix + "head.")

    def unused_parameter_names(self):
        """Parameters that never take part in the forward pass (reference: their .grad stays None, so AdamW and the
        DDP reducer ignore them - 19 tensors, the reason for find_unused_parameters=True at train.py:106)."""
        pre = self.arena_prefix
        names = [pre + "backbone.cls_token"]
        names += [pre + n for n, _ in self.named_parameters() if n.startswith("segmentation.conv_mla.")]
        return names

    # ---------------------------------------------------------------------------------------- sub-steps
    def _pooled_logits(self, tokens, sel):
        rows = engine.RegionPoolFn.apply(tokens, sel)
        # (lazy: the head may leave its last product to the loss - engine.LazyLogits, ccd_head_loss_fwd / _bwd)
        return self.head.forward_rows(rows, sel.total, lazy=True)

    def attention(self, feature, clusters):
        """Reference-shaped helper (dino_vision.py:38-49): feature [N,E,8,32], clusters [N,26,32,128] -> ([N,26,E], index)."""
        n, e = feature.shape[0], feature.shape[1]
        idmap = ops.planes_to_idmap(clusters)
        tok_plane, tok_coef, present = ops.region_stats(idmap)
        tokens = feature.permute(0, 2, 3, 1).reshape(n, 256, e).float()
        onehot = (tok_plane.long().unsqueeze(-1) == torch.arange(26, device=feature.device)).float()
        w = onehot * tok_coef.unsqueeze(-1)                                  # [N,256,26], tiny glue for API parity
        return torch.bmm(w.transpose(1, 2), tokens), present.bool()

    def backbone_tokens(self, x):
        """Final-norm tokens of both views [2B,256,E] (no taps): the teacher's backbone pass on its own, so that it can be
        enqueued before the student's clusters exist (pretrain._forward_backward with a CU partition)."""
        self.ensure_arena()
        tokens, = self.backbone.tokens_and_taps(torch.cat([x[:, 1], x[:, 2]]), need_taps=False)
        return tokens

    def forward(self, x, metrics, target_mask, epoch, clusters=None, index=None, tokens=None):
        self.ensure_arena()
        B = x.shape[0]
        if tokens is None:
            views = torch.cat([x[:, 1], x[:, 2]])
            tokens, *taps = self.backbone.tokens_and_taps(views, need_taps=clusters is None)
        else:
            assert clusters is not None, "precomputed tokens are the teacher's (no segmentation taps)"
        if clusters is None:
            seg_in = [self.backbone.to_2D(t) for t in taps]
            seg = self.segmentation(seg_in)                                    # [2B,2,32,128] fp32
            if epoch < 30:
                mask = target_mask.contiguous().float()
            else:
                mask = ops.seg_to_mask(seg.detach().contiguous(), B)
            ids_src = ops.ccl_label(mask)
            ids_img = ops.warp_idmap(ids_src, metrics.contiguous().float())
            sel = engine.Selection(torch.cat([ids_src, ids_img]), B)
            logits = self._pooled_logits(tokens, sel)
            return ModelOutput(
                {"mask": seg, "image": x, "zero": ClusterMaps(sel), "logits_buf": logits, "selection": sel},
                {"instances_view": lambda: engine.logits_tensor(logits)[: 2 * sel.M], "index": lambda: sel.new_index.bool()})
        sel = clusters.selection if isinstance(clusters, ClusterMaps) else \
            engine.Selection(ops.planes_to_idmap(clusters), B)
        logits = self._pooled_logits(tokens, sel)
        return ModelOutput({"logits_buf": logits, "selection": sel},
                           {"instances_view": lambda: engine.logits_tensor(logits)[: 2 * sel.M],
                            "feature": lambda: self.backbone.to_2D(tokens).float()})


# ------------------------------------------------------------------------------------------------ finetune
class DINO_Finetune(ArenaModule):
    """Text recogniser of the finetune stage (Dino/model/dino_vision.py:134-290): ViT backbone -> Mlp "encoder" ->
    NRTR decoder -> TFLoss.  Same constructor (a config object), parameter names and init RNG order as the reference."""

    def __init__(self, config):
        super().__init__()
        from ..convertor.attn import AttnConvertor
        from ..decoder.nrtr_decoder import Mlp, NRTRDecoder
        from ..loss.ce_loss import TFLoss
        from ..modules import vision_transformer as vits
        self.label_convertor = AttnConvertor(dict_type='DICT90', max_seq_len=config.decoder_max_seq_len, with_unknown=True)
        config.arch = config.arch.replace("deit", "vit")
        if config.arch not in vits.__dict__:
            raise NotImplementedError(f"Unknow architecture: {config.arch} (HIP kernels cover vit_tiny / vit_small / vit_base)")
        self.backbone = vits.__dict__[config.arch](patch_size=config.patch_size, drop_path_rate=config.drop_path_rate)
        embed_dim = self.backbone.embed_dim
        self.encoder = Mlp(in_features=embed_dim, hidden_features=512, out_features=512, act_layer=nn.GELU, drop=0.1)
        config.decoder_num_classes = self.label_convertor.num_classes()
        config.decoder_start_idx = self.label_convertor.start_idx
        config.decoder_padding_idx = self.label_convertor.padding_idx
        self.decoder = NRTRDecoder(
            n_layers=config.decoder_n_layers, d_embedding=config.decoder_d_embedding, n_head=config.decoder_n_head,
            d_k=config.decoder_d_k, d_v=config.decoder_d_v, d_model=config.decoder_d_model, d_inner=config.decoder_d_inner,
            n_position=200, dropout=0.1, num_classes=config.decoder_num_classes, max_seq_len=config.decoder_max_seq_len,
            start_idx=config.decoder_start_idx, padding_idx=config.decoder_padding_idx)
        self.loss = TFLoss(ignore_index=self.label_convertor.padding_idx)

    # ------------------------------------------------------------------------------------------- arena
    def _transposed_names(self):
        return ["backbone." + n for n in self.backbone._transposed_names()] + \
               ["encoder." + n for n in self.encoder._transposed_names()] + \
               ["decoder." + n for n in self.decoder._transposed_names()]

    def attach_arena(self, arena, prefix):
        super().attach_arena(arena, prefix)
        self.backbone.attach_arena(arena, prefix + "backbone.")
        self.encoder.attach_arena(arena, prefix + "encoder.")
        self.decoder.attach_arena(arena, prefix + "decoder.")

    def unused_parameter_names(self):
        """cls_token and the three norm_seg LayerNorms never take part in this forward pass (their .grad stays None in
        the reference, so torch's AdamW skips them)."""
        pre = self.arena_prefix
        return [pre + "backbone.cls_token"] + [pre + f"backbone.norm_seg.{j}.{k}" for j in range(3) for k in ("weight", "bias")]

    # ------------------------------------------------------------------------------ reference surface
    def forward(self, img, text, return_loss=True, test_speed=False):
        if return_loss:
            return self.forward_train(img, text)
        if test_speed:
            return self.forward_test_speed(img)
        return self.forward_test(img)

    def extract_feat(self, img):
        """Final-norm tokens [N,256,E] (bf16, the kernels' native layout)."""
        self.ensure_arena()
        tokens, = self.backbone.tokens_and_taps(img, need_taps=False)
        return tokens

    def forward_train(self, img, img_metas):
        """img [N,3,32,128], img_metas = padded target indices int64 [N,T] -> (loss, attn [N,H,T,256])."""
        feat = self.extract_feat(img)
        targets_dict = {'padded_targets': img_metas}
        out_enc = self.encoder(feat)
        out_dec, attn = self.decoder(feat, out_enc, targets_dict, train_mode=True)
        return self.loss(out_dec, targets_dict), attn

    def forward_test(self, img):
        feat = self.extract_feat(img)
        return self.decoder(feat, self.encoder(feat), None, train_mode=False)

    def forward_test_speed(self, img):
        feat = self.extract_feat(img)
        return self.decoder(feat, self.encoder(feat), None, train_mode=False, test_speed=True)
