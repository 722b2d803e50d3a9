"""Pins the CPU oracle (oracle/) against fixtures produced by the REAL reference (tools/gen_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ccd_amd.synthetic import make_batch
from oracle import ccd_oracle as O
from oracle import ccl_np

pytestmark = pytest.mark.filterwarnings("ignore")


def _stat(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.pow(2).sum().sqrt().item()])


def _check_stats(names, stats, table, rtol, atol=1e-7, what=""):
    for n, row in zip(names, stats):
        got = _stat(table[str(n)])
        # [sum, abs-sum, l2]: the plain sum may cancel to ~0, so its error is judged against the abs-sum
        np.testing.assert_allclose(got[1:], row[1:], rtol=rtol, atol=atol, err_msg=f"{what}:{n}")
        assert abs(got[0] - row[0]) <= rtol * row[1] + atol, f"{what}:{n} sum {got[0]} vs {row[0]}"


def test_schedules(golden_dir):
    g = np.load(os.path.join(golden_dir, "sched.npz"))
    np.testing.assert_array_equal(O.cosine_iter_schedule(0.0005 * 8 / 256.0, 1e-6, 50, warmup_iters=10), g["lr"])
    np.testing.assert_array_equal(O.cosine_iter_schedule(0.04, 0.4, 50), g["wd"])
    np.testing.assert_array_equal(O.cosine_iter_schedule(0.9995, 1, 50), g["mom"])
    np.testing.assert_array_equal(O.cosine_iter_schedule(1e-3, 1e-5, 17), g["lr_nowarm"])
    np.testing.assert_array_equal(O.teacher_temp_schedule(0.04, 0.04, 0, 40), g["teacher_temp_0_40"])
    np.testing.assert_array_equal(O.teacher_temp_schedule(0.02, 0.07, 5, 12), g["teacher_temp_5_12"])


def test_ccl_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "ccl_cases.npz"))
    for name, mask, want, tie in zip(g["names"], g["masks"], g["idmaps"], g["has_tie"]):
        got = ccl_np.label_idmap(mask)
        if not tie:
            np.testing.assert_array_equal(got, want, err_msg=str(name))
        else:  # order inside a mean-column tie is unspecified in the reference (unstable argsort)
            assert (got == 255).sum() == (want == 255).sum()
            pairs = set(zip(got[got != 255].tolist(), want[want != 255].tolist()))
            assert len(pairs) == len(set(p[0] for p in pairs)), f"{name}: not a relabelling"


def test_state_keys_and_init(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for arch, cfg in O.ARCH.items():
        spec = O.Spec(out_dim=1024, seg_in=cfg["embed_dim"], norm_last_layer=False, **cfg)
        s, t = O.build_pair(spec, seed=0)
        assert [[k, list(v.shape), str(v.dtype)] for k, v in s.P.items()] == keys[arch]["student"]
        assert [[k, list(v.shape), str(v.dtype)] for k, v in t.P.items()] == keys[arch]["teacher"]
        assert s.trainable == keys[arch]["student_trainable"]


def _tiny_spec():
    return O.Spec(embed_dim=192, depth=3, heads=3, taps=(1, 2, 3), out_dim=512, head_hidden=256,
                  head_bottleneck=64, norm_last_layer=False, seg_in=192)


@pytest.mark.parametrize("B,name", [(2, "tiny_step.npz"), (8, "tiny8_step.npz")])
def test_tiny_step(golden_dir, B, name):
    g = np.load(os.path.join(golden_dir, name))
    student, teacher = O.build_pair(_tiny_spec(), seed=3)
    _check_stats(g["init_names"], g["init_stats"], student.P, rtol=0, atol=0, what="init")
    batch = make_batch(B, seed=11)
    np.testing.assert_array_equal(batch[1].numpy().astype(np.uint8), g["masks"])
    np.testing.assert_array_equal(batch[2].numpy(), g["metrics"])
    epoch, lr, wd, mom, clip, freeze = g["hyper"]
    rec = O.train_iteration(student, teacher, torch.zeros(1, 512), O.AdamWState(), batch, int(epoch), lr, wd, mom,
                            clip=clip, freeze_last_layer=int(freeze))
    s, t = rec["s_out"], rec["t_out"]
    np.testing.assert_array_equal(s["idmap"], g["zero_idmap"])
    np.testing.assert_array_equal(s["index"].numpy(), g["new_index"])
    np.testing.assert_array_equal(rec["masks_image"].numpy().astype(np.uint8), g["masks_image"])
    np.testing.assert_allclose(s["mask"].detach().numpy(), g["seg_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s["instances_view"].detach().numpy(), g["student_logits"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(t["instances_view"].detach().numpy(), g["teacher_logits"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(t["feature"].detach().numpy()[:, ::4], g["teacher_feature"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose([rec["loss"], rec["mask_loss"], rec["dino_loss"]], g["losses"], rtol=1e-6)
    np.testing.assert_allclose(rec["center"].numpy(), g["center_after"], rtol=1e-5, atol=1e-8)
    assert sorted(rec["grads_raw"]) == sorted(map(str, g["grad_names"]))
    _check_stats(g["grad_names"], g["grad_stats"], rec["grads_raw"], rtol=2e-4, atol=1e-9, what="grad")
    for k in g.files:
        if k.startswith("grad/"):
            np.testing.assert_allclose(rec["grads_raw"][k[5:]].numpy(), g[k], rtol=1e-3, atol=1e-7, err_msg=k)
    _check_stats(g["post_names"], g["post_stats"], student.P, rtol=1e-4, atol=1e-7, what="post")
    _check_stats(g["teacher_post_names"], g["teacher_post_stats"], teacher.P, rtol=1e-4, atol=1e-7, what="ema")


@pytest.mark.parametrize("arch", ["vit_base", "vit_base_768"])
def test_arch_step(golden_dir, arch):
    """BASELINE config #4's architectures: the oracle against one iteration of the real reference (tools/gen_golden.py::gen_arch)."""
    g = np.load(os.path.join(golden_dir, "arch_step.npz"))
    p = arch + "/"
    epoch, lr, wd, mom, clip, freeze, seed, B, K = g[p + "hyper"]
    spec = O.Spec(norm_last_layer=False, out_dim=int(K), seg_in=O.ARCH[arch]["embed_dim"], **O.ARCH[arch])
    student, teacher = O.build_pair(spec, seed=0)
    _check_stats(g[p + "init_names"], g[p + "init_stats"], student.P, rtol=0, atol=0, what="init")
    rec = O.train_iteration(student, teacher, torch.zeros(1, int(K)), O.AdamWState(), make_batch(int(B), seed=int(seed)), int(epoch), lr,
                            wd, mom, clip=clip, freeze_last_layer=int(freeze))
    np.testing.assert_array_equal(rec["s_out"]["idmap"], g[p + "zero_idmap"])
    np.testing.assert_array_equal(rec["s_out"]["index"].numpy(), g[p + "new_index"])
    np.testing.assert_allclose([rec["loss"], rec["mask_loss"], rec["dino_loss"]], g[p + "losses"], rtol=2e-6)
    np.testing.assert_allclose(rec["center"].numpy(), g[p + "center_after"], rtol=1e-5, atol=1e-8)
    _check_stats(g[p + "grad_names"], g[p + "grad_stats"], rec["grads_raw"], rtol=5e-4, atol=1e-9, what="grad")


def test_small_step(golden_dir):
    """CCD_pretrain_ViT_small hyper-parameters, B=8, two iterations (BASELINE config #1)."""
    g = np.load(os.path.join(golden_dir, "small_step.npz"))
    spec = O.Spec(norm_last_layer=False, **O.ARCH["vit_small"])
    student, teacher = O.build_pair(spec, seed=0)
    _check_stats(g["init_names"], g["init_stats"], student.P, rtol=0, atol=0, what="init")
    center, opt = torch.zeros(1, spec.out_dim), O.AdamWState()
    for step in range(2):
        p = f"s{step}/"
        epoch, lr, wd, mom, clip, freeze, seed = g[p + "hyper"]
        batch = make_batch(8, seed=int(seed))
        np.testing.assert_array_equal(batch[1].numpy().astype(np.uint8), g[p + "masks"])
        np.testing.assert_allclose(_stat(batch[0]), g[p + "image_stat"], rtol=1e-12)
        rec = O.train_iteration(student, teacher, center, opt, batch, int(epoch), lr, wd, mom, clip=clip,
                                freeze_last_layer=int(freeze))
        center = rec["center"]
        s, t = rec["s_out"], rec["t_out"]
        np.testing.assert_array_equal(s["idmap"], g[p + "zero_idmap"])
        np.testing.assert_array_equal(s["index"].numpy(), g[p + "new_index"])
        np.testing.assert_array_equal(rec["masks_image"].numpy().astype(np.uint8), g[p + "masks_image"])
        np.testing.assert_allclose([rec["loss"], rec["mask_loss"], rec["dino_loss"]], g[p + "losses"], rtol=2e-6)
        r, c = g[p + "rows"], g[p + "cols"]
        np.testing.assert_allclose(s["instances_view"].detach()[r][:, c].numpy(), g[p + "student_logits_sample"],
                                   rtol=1e-3, atol=2e-6)
        np.testing.assert_allclose(t["instances_view"].detach()[r][:, c].numpy(), g[p + "teacher_logits_sample"],
                                   rtol=1e-3, atol=2e-6)
        np.testing.assert_allclose(center[0, c].numpy(), g[p + "center_sample"], rtol=1e-4, atol=1e-8)
        _check_stats(g[p + "grad_names"], g[p + "grad_stats"], rec["grads_raw"], rtol=1e-3, atol=2e-7, what="grad")
        _check_stats(g[p + "grad_names"], g[p + "grad_clipped_stats"], rec["grads_clipped"], rtol=1e-3, atol=2e-7,
                     what="clipped")
        _check_stats(g[p + "post_names"], g[p + "post_stats"], student.P, rtol=1e-4, atol=1e-6, what="post")
        _check_stats(g[p + "teacher_post_names"], g[p + "teacher_post_stats"], teacher.P, rtol=1e-5, atol=1e-6,
                     what="ema")
    # predicted-mask branch: CCL + warp + row selection on the reference's own thresholded prediction
    ids = O.label_batch(g["pred/mask"])
    src = torch.from_numpy(ccl_np.idmap_to_planes(ids))
    clusters = torch.cat([src, O.warp_planes(src, torch.from_numpy(g["pred/metrics"]))])
    np.testing.assert_array_equal(ccl_np.planes_to_idmap(clusters.numpy()), g["pred/zero_idmap"])


def perturb_head_biases(P, seed, scale):
    """The fixture's deterministic head-bias perturbation (tools/gen_golden.py: perturb_head_biases) on a name -> tensor map."""
    g = torch.Generator().manual_seed(int(seed))
    with torch.no_grad():
        for n, p in P.items():
            if n.startswith("head.") and n.endswith(".bias"):
                p.add_(float(scale) * torch.randn(p.shape, generator=g))


def test_small3_steps(golden_dir):
    """The same model with NON-ZERO head biases (no pooled row is an exact zero vector entering F.normalize, so none of the
    reference's gradients is amplified rounding residue): three consecutive iterations and a fourth on the predicted-mask
    branch (epoch 30), every loss, every gradient norm, every post-step tensor against the real reference - no exclusions."""
    from ccd_amd.synthetic import make_text_like_batch
    g = np.load(os.path.join(golden_dir, "small3_step.npz"))
    spec = O.Spec(norm_last_layer=False, **O.ARCH["vit_small"])
    student, teacher = O.build_pair(spec, seed=0)
    perturb_head_biases(student.P, *g["perturb"])
    for k in teacher.P:
        if k.startswith("head."):
            teacher.P[k] = student.P[k].detach().clone()
    _check_stats(g["init_names"], g["init_stats"], student.P, rtol=0, atol=0, what="init")
    with torch.no_grad():       # the fixture's fitted last layer of the segmentation head (tools/gen_golden.py: fit_seg_classifier)
        student.P["segmentation.cls.weight"].copy_(torch.from_numpy(g["cls_weight"]))
        student.P["segmentation.cls.bias"].copy_(torch.from_numpy(g["cls_bias"]))
    center, opt = torch.zeros(1, spec.out_dim), O.AdamWState()
    for step in range(4):
        p = f"s{step}/"
        epoch, lr, wd, mom, clip, freeze, seed = g[p + "hyper"]
        batch = (make_text_like_batch if epoch >= 30 else make_batch)(8, seed=int(seed))
        np.testing.assert_array_equal(batch[1].numpy().astype(np.uint8), g[p + "masks"])
        rec = O.train_iteration(student, teacher, center, opt, batch, int(epoch), lr, wd, mom, clip=clip,
                                freeze_last_layer=int(freeze))
        center = rec["center"]
        s = rec["s_out"]
        if epoch >= 30:
            seg1 = s["mask"].detach()[:8]
            np.testing.assert_allclose(seg1.numpy(), g[p + "seg_logits_view1"], rtol=1e-3, atol=2e-5)
            np.testing.assert_array_equal((F.softmax(seg1, 1)[:, 1] > 0.5).numpy().astype(np.uint8), g[p + "pred_mask"])
            assert len(np.unique(g[p + "zero_idmap"])) > 3, "fixture: the predicted masks must hold components"
            assert float(g[p + "pred_margin_min"][0]) > 1.0, "fixture: the prediction must have real margins"
        np.testing.assert_array_equal(s["idmap"], g[p + "zero_idmap"])
        np.testing.assert_array_equal(s["index"].numpy(), g[p + "new_index"])
        np.testing.assert_allclose([rec["loss"], rec["mask_loss"], rec["dino_loss"]], g[p + "losses"], rtol=5e-6)
        _check_stats(g[p + "grad_names"], g[p + "grad_stats"], rec["grads_raw"], rtol=2e-3, atol=5e-7, what="grad")
        # (atol: Adam moves an element whose gradient is ~0 by +-lr whichever way rounding tips it - with the fitted classifier
        # the mask loss is small and a few BatchNorm biases of the segmentation head see such gradients: 2e-6 on an abs-sum of 3e-3)
        _check_stats(g[p + "post_names"], g[p + "post_stats"], student.P, rtol=1e-4, atol=5e-6, what="post")
        _check_stats(g[p + "teacher_post_names"], g[p + "teacher_post_stats"], teacher.P, rtol=1e-5, atol=1e-6, what="ema")


@pytest.mark.parametrize("tag,arch,n_layers,B", [("tiny", "vit_tiny", 2, 4), ("small", "vit_small", 6, 8)])
def test_finetune_step(golden_dir, tag, arch, n_layers, B):
    """DINO_Finetune - two AdamW iterations + greedy decoding (train_finetune.py:262-289)."""
    from oracle import finetune_oracle as FO
    g = np.load(os.path.join(golden_dir, "finetune_step.npz"))
    np.testing.assert_array_equal(FO.cosine_scheduler(0.0005, 1e-6, 3, 20, warmup_epochs=1), g["sched"])
    spec = FO.FtSpec(vit=O.Spec(**O.ARCH[arch]), n_layers=n_layers)
    net = FO.init_finetune(spec, seed=0)
    _check_stats(g[f"{tag}/init_names"], g[f"{tag}/init_stats"], net.P, rtol=0, atol=0, what="init")
    words = [str(w) for w in g["words"][:B]]
    targets = FO.str2tensor(words)
    np.testing.assert_array_equal(targets.numpy(), g[f"{tag}/targets"])
    opt = O.AdamWState()
    gen = torch.Generator().manual_seed(1234)
    for step in range(2):
        p = f"{tag}/s{step}/"
        img = torch.randn(B, 3, 32, 128, generator=gen)
        np.testing.assert_allclose(_stat(img), g[p + "image_stat"], rtol=1e-12)
        loss_ref, lr = g[p + "loss"]
        rec = FO.train_iteration(net, opt, img, targets, lr)
        np.testing.assert_allclose(rec["loss"], loss_ref, rtol=2e-6)
        np.testing.assert_allclose(rec["attn"].mean(1).numpy(), g[p + "attn_mean"], rtol=1e-3, atol=1e-6)
        _check_stats(g[p + "grad_names"], g[p + "grad_stats"], rec["grads_raw"], rtol=1e-3, atol=2e-7, what="grad")
        assert set(rec["grads_raw"]) == set(str(n) for n in g[p + "grad_names"])
        _check_stats(g[p + "post_names"], g[p + "post_stats"], net.P, rtol=1e-4, atol=1e-6, what="post")
    img = torch.randn(B, 3, 32, 128, generator=gen)
    np.testing.assert_allclose(_stat(img), g[f"{tag}/eval_image_stat"], rtol=1e-12)
    with torch.no_grad():
        _, logits, _ = FO.forward_train(net.P, spec, img, targets)
        probs = FO.forward_test(net.P, spec, img)
    np.testing.assert_allclose(logits.numpy(), g[f"{tag}/logits"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(probs.numpy(), g[f"{tag}/test_probs"], rtol=2e-3, atol=1e-5)
    np.testing.assert_array_equal(probs.argmax(-1).numpy(), g[f"{tag}/test_probs"].argmax(-1))
