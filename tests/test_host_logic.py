"""Host-side logic of the product package (no GPU, no kernels): config surface, schedules, parameter naming / init
parity with the reference, the bicubic positional resampling matrix, optimizer checkpoint layout, C-ABI symbols."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_c_abi_library_exports_every_declared_symbol():
    from ccd_amd import _lib
    header = open(os.path.join(ROOT, "include", "ccd_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(?:int|long|const char\*)\s+(ccd_\w+)\s*\(", header)))
    assert len(declared) >= 35, declared
    assert os.path.isfile(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)            # loads without a GPU; no compute call is made here
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared, "ccd_amd/_lib.py signature table out of sync with include/ccd_hip.h"
    _lib.bind(lib)
    # the library on disk was built from THESE sources (a stale in-tree .so would carry the previous number)
    src = open(os.path.join(ROOT, "ccd_amd", "csrc", "abi_impl.h")).read()
    want = int(re.search(r"int ccd_abi_version\(void\) \{ return (\d+); \}", src).group(1))
    assert want >= 11 and lib.ccd_abi_version() == want


def test_product_has_no_cpu_fallback():
    """Without the HIP library / a GPU the ops raise instead of computing something on the host."""
    from ccd_amd import _lib
    from ccd_amd.modules import vision_transformer as vits
    assert _lib._stream_override is None
    m = vits.VisionTransformer(patch_size=4, embed_dim=64, depth=1, num_heads=1, qkv_bias=True, out_indices=[1, 1, 1])
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 32, 128))
    for f in os.listdir(os.path.join(ROOT, "ccd_amd")):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "ccd_amd", f)).read().replace("the CPU oracle", ""), f


def test_schedules_match_reference(golden_dir):
    from ccd_amd.loss.Dino_loss import DINOLoss
    from ccd_amd.modules import utils
    g = np.load(os.path.join(golden_dir, "sched.npz"))
    np.testing.assert_array_equal(utils.cosine_iter_scheduler(0.0005 * 8 / 256.0, 1e-6, 50, warmup_iters=10), g["lr"])
    np.testing.assert_array_equal(utils.cosine_iter_scheduler(0.04, 0.4, 50), g["wd"])
    np.testing.assert_array_equal(utils.cosine_iter_scheduler(0.9995, 1, 50), g["mom"])
    np.testing.assert_array_equal(DINOLoss(16, 2, 0.02, 0.07, 5, 12).teacher_temp_schedule, g["teacher_temp_5_12"])


def test_state_dict_names_shapes_and_trainable_sets(golden_dir):
    from ccd_amd.model.dino_vision import ABIDINOModel
    from ccd_amd.modules import utils, vision_transformer as vits
    from ccd_amd.modules.segmentor import SegHead
    keys = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for arch, e in (("vit_tiny", 192), ("vit_small", 384), ("vit_base", 512)):
        torch.manual_seed(0)
        sb = vits.__dict__[arch](patch_size=4, drop_path_rate=0.1)
        tb = vits.__dict__[arch](patch_size=4)
        student = ABIDINOModel(sb, SegHead(in_channels=e, mla_channels=128, mlahead_channels=64, num_classes=2),
                               vits.DINOHead(e, 1024, norm_last_layer=False))
        teacher = ABIDINOModel(tb, None, vits.DINOHead(e, 1024))
        assert [[k, list(v.shape), str(v.dtype)] for k, v in student.state_dict().items()] == keys[arch]["student"]
        assert [[k, list(v.shape), str(v.dtype)] for k, v in teacher.state_dict().items()] == keys[arch]["teacher"]
        assert [n for n, p in student.named_parameters() if p.requires_grad] == keys[arch]["student_trainable"]
        groups = utils.get_params_groups(student)
        assert groups[1]["weight_decay"] == 0.0 and all(p.dim() == 1 for p in groups[1]["params"])
        assert utils.has_batchnorms(student) and not utils.has_batchnorms(teacher)
        unused = student.unused_parameter_names()
        assert len(unused) == 19 and "backbone.cls_token" in unused          # SURVEY: 19 tensors never get a gradient


def test_bicubic_resample_matrix_is_aten_bicubic():
    """The fixed 256x256 map used for pos-embed resampling == F.interpolate(bicubic, scale_factor=(8.1/16, 32.1/16))."""
    from ccd_amd.engine import bicubic_resample_matrix
    m = bicubic_resample_matrix(16, 8, 32)
    g = torch.Generator().manual_seed(0)
    pos = torch.randn(1, 256, 24, generator=g)
    want = F.interpolate(pos.reshape(1, 16, 16, 24).permute(0, 3, 1, 2), scale_factor=(8.1 / 16, 32.1 / 16),
                         mode="bicubic").permute(0, 2, 3, 1).reshape(256, 24)
    np.testing.assert_allclose((m @ pos[0]).numpy(), want.numpy(), rtol=0, atol=2e-6)
    assert abs(float(m.sum(1).mean()) - 1.0) < 1e-5        # rows are interpolation weights


def test_config_surface():
    from ccd_amd.utils.utils import Config
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        for name, arch, seg, nl in (("small", "vit_small", 384, False), ("Tiny", "vit_tiny", 192, False),
                                    ("Base", "vit_base", 512, True)):
            c = Config(f"Dino/configs/CCD_pretrain_ViT_{name}.yaml")
            assert (c.arch, c.model_seg_channel, c.norm_last_layer) == (arch, seg, nl)
            assert (c.patch_size, c.out_dim, c.crops_number, c.clip_grad, c.optimizer) == (4, 65536, 2, 3.0, "adamw")
            assert (c.dataset_image_height, c.dataset_image_width, c.training_epochs, c.imgnet_based) == (32, 128, 3, 1000000)
            assert c.not_a_key is None and isinstance(c.dataset_train, dict)
            assert c.global_workdir == os.path.join("workdir", c.global_name)
    finally:
        os.chdir(cwd)


def test_dino_import_paths_alias_the_product():
    import Dino  # noqa: F401
    from Dino.loss.Dino_loss import DINOLoss
    from Dino.model.dino_vision import ABIDINOModel
    from Dino.modules import vision_transformer as vits
    import ccd_amd.modules.vision_transformer as native
    assert vits is native and ABIDINOModel.__module__.startswith("ccd_amd") and DINOLoss.__module__.startswith("ccd_amd")


def test_finetune_state_dict_matches_reference(golden_dir):
    """DINO_Finetune's state dict has exactly the reference's keys, in its order (fixture written by
    the real reference), and the reference's AttnConvertor indices."""
    import numpy as np
    from ccd_amd.finetune import FinetuneConfig
    from ccd_amd.model.dino_vision import DINO_Finetune
    g = np.load(os.path.join(golden_dir, "finetune_step.npz"))
    for tag, arch, layers in (("tiny", "vit_tiny", 2), ("small", "vit_small", 6)):
        torch.manual_seed(0)
        m = DINO_Finetune(FinetuneConfig(arch=arch, decoder_n_layers=layers))
        assert list(m.state_dict().keys()) == [str(n) for n in g[f"{tag}/init_names"]]
        conv = m.label_convertor
        assert (conv.num_classes(), conv.start_idx, conv.end_idx, conv.padding_idx, conv.unknown_idx) == (93, 91, 91, 92, 90)
        words = [str(w) for w in g["words"]]
        np.testing.assert_array_equal(conv.str2tensor(words[:4 if tag == "tiny" else 8]).numpy(), g[f"{tag}/targets"])
    idx, scores = conv.tensor2idx(torch.nn.functional.one_hot(conv.str2tensor(["Wor1d!"])[:, 1:], 93).float() * 20)
    assert conv.idx2str(idx) == ["Wor1d!"] and len(scores[0]) == 6


def test_reference_checkpoint_layout_round_trip(golden_dir, tmp_path):
    """a checkpoint in the published layout ({student, teacher, ...}, DDP `module.` prefix, the key /
    shape table of the real reference) loads strictly into the build, exports bit-identically, and its teacher is picked
    up by the finetune model the way train_finetune.py:190-198 does it."""
    from ccd_amd.finetune import FinetuneConfig
    from ccd_amd.model.dino_vision import ABIDINOModel, DINO_Finetune
    from ccd_amd.modules import vision_transformer as vits
    from ccd_amd.modules.segmentor import SegHead
    from ccd_amd.parallel import DataParallel
    keys = json.load(open(os.path.join(golden_dir, "state_keys.json")))["vit_small"]
    gen = torch.Generator().manual_seed(7)

    def fabricate(table):
        sd = {}
        for k, shape, dt in table:
            sd["module." + k] = torch.randn(shape, generator=gen) if dt == "torch.float32" else torch.tensor(3, dtype=torch.int64)
        return sd
    ckpt = {"student": fabricate(keys["student"]), "teacher": fabricate(keys["teacher"]), "epoch": 3, "iteration": 1234}
    path = tmp_path / "checkpoint.pth"
    torch.save(ckpt, path)
    loaded = torch.load(path, map_location="cpu", weights_only=False)
    student = DataParallel(ABIDINOModel(vits.vit_small(patch_size=4, drop_path_rate=0.1),
                                        SegHead(in_channels=384, mla_channels=128, mlahead_channels=64, num_classes=2),
                                        vits.DINOHead(384, 1024, norm_last_layer=False)))
    teacher = DataParallel(ABIDINOModel(vits.vit_small(patch_size=4), None, vits.DINOHead(384, 1024)))
    for net, key in ((student, "student"), (teacher, "teacher")):
        msg = net.load_state_dict(loaded[key], strict=True)
        assert not msg.missing_keys and not msg.unexpected_keys
        out = net.state_dict()
        assert list(out) == list(loaded[key])
        assert all(torch.equal(out[k], loaded[key][k]) for k in out)
    ft = DataParallel(DINO_Finetune(FinetuneConfig(arch="vit_small")))
    dd = ft.state_dict()
    picked = [n for n in dd if n in loaded["teacher"]]
    ft.load_state_dict({n: loaded["teacher"].get(n, v) for n, v in dd.items()})
    assert len(picked) == 156 and all(n.startswith("module.backbone.") for n in picked)      # the whole ViT backbone
    assert all(torch.equal(ft.state_dict()[n], loaded["teacher"][n]) for n in picked)


def test_every_16_byte_buffer_store_holds_its_data_registers():
    """gfx950 reads the data registers of `buffer_store_dwordx4 ... soffset` a few cycles AFTER the instruction issues and LLVM's hazard
    recogniser exempts exactly that form (docs/LAB_NOTEBOOK.md section 4d: a VALU write right behind the store reached memory instead of the data).
    The fix is a convention - every 16-byte buffer store is followed by buf_store_data_hold - so it is enforced here: the raw builtin /
    instruction may only appear inside prelude_hip.h helpers, and each occurrence must be followed by the hold before the helper ends."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ccd_amd", "csrc")
    pat = re.compile(r"__builtin_amdgcn_raw(?:_ptr)?_buffer_store_b128|buffer_store_dwordx4")
    strip = lambda text: re.sub(r"//[^\n]*", "", text)
    for path in glob.glob(os.path.join(root, "**", "*.h"), recursive=True) + glob.glob(os.path.join(root, "*.hip")):
        code = strip(open(path).read())
        hits = [m.start() for m in pat.finditer(code)]
        if os.path.basename(path) != "prelude_hip.h":
            assert not hits, f"{path}: a raw 16-byte buffer store outside prelude_hip.h (use buf_store16 / buf_store16_nt / stream_store16)"
            continue
        assert hits, "prelude_hip.h no longer defines the 16-byte buffer stores?"
        for h in hits:
            end = code.index("\n}", h)                      # the helper's closing brace
            assert "buf_store_data_hold(" in code[h:end], f"prelude_hip.h: 16-byte buffer store at offset {h} without buf_store_data_hold"


def test_row_owner_kernels_stay_within_their_recorded_spill_ceilings():
    """The 512-register row-owner kernels compile without spills in their product loops only as long as nothing pushes the allocator
    over the edge (docs/LAB_NOTEBOOK.md section 4e: 227 spills from one `if`, 122 scalar spills from hoisted descriptors) - and a spill costs milliseconds
    silently.  The code object's own figures (tools/codeobj_regs.py) are checked against ceilings recorded on the round-6 tree:
    kernel pattern -> (VGPR spills, SGPR spills, scratch bytes).  A compiler bump or an edit that raises one of them fails HERE, on the
    CPU, not as a slower step on the GPU.  (The spills that are recorded sit outside the chunk loops: checked in the ISA when recorded.)"""
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import codeobj_regs
    regs = codeobj_regs.load()
    ceilings = {
        r"ccd::mlp_fused_kernel<384, false, true>": (72, 8, 292),
        r"ccd::mlp_fused_kernel<384, true, true>": (78, 101, 304),
        r"ccd::mlp_fused_kernel<384, false, false>": (0, 0, 0),
        r"ccd::mlp_fused_kernel<384, true, false>": (0, 24, 0),
        r"ccd::rowgemm_kernel<384, 3, 0, true, false, false>": (2, 12, 12),
        r"ccd::rowgemm_kernel<384, 3, 0, true, true, false>": (2, 12, 12),
        r"ccd::rowgemm_kernel<384, 3, 0, true, true, true>": (2, 24, 12),          # + a tap's LayerNorm backward in the epilogue
        r"ccd::rowgemm_kernel<384, 3, 1, true, false, false>": (0, 0, 0),
        r"ccd::rowproj_kernel<384, 2>": (0, 0, 0),
        r"ccd::attention_bwd_onepass_kernel": (0, 10, 0),
        r"ccd::attention_fwd_kernel": (0, 0, 0),
        r"ccd::cls_tail_fwd_kernel": (0, 0, 0),
        r"ccd::cls_tail_bwd_kernel<false, 4>": (0, 0, 0),
        r"ccd::cls_tail_bwd_kernel<true, 4>": (0, 0, 0),
        r"ccd::head_loss_kernel<false>": (0, 0, 0),
        r"ccd::head_loss_kernel<true>": (0, 0, 0),
        r"ccd::gemm256_kernel<5, 256, false>": (0, 0, 0),
        r"ccd::gemm_tn384_kernel<4, 2, 4, 3, 3>": (0, 0, 0),
    }
    for name, (vs, ss, sb) in ceilings.items():
        assert name in regs, f"{name}: not in the code object (renamed? update the ceiling table)"
        r = regs[name]
        got = (int(r["vgpr_spill"]), int(r["sgpr_spill"]), int(r["scratch"]))
        assert got[0] <= vs and got[1] <= ss and got[2] <= sb, f"{name}: (VGPR spills, SGPR spills, scratch B) = {got}, recorded ceiling {(vs, ss, sb)}"
        assert int(r["vgpr"]) <= 512
    # two workgroups per CU is what these kernels' launch geometry assumes: <= 256 registers
    for name in (r"ccd::attention_bwd_onepass_kernel", r"ccd::cls_tail_bwd_kernel<false, 4>", r"ccd::cls_tail_bwd_kernel<true, 4>", r"ccd::cls_tail_fwd_kernel"):
        assert int(regs[name]["vgpr"]) <= 256, (name, regs[name]["vgpr"])
