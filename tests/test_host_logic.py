"""CPU: host-side mirror of the reference interface (no GPU, no compute in the library)."""
import os

import numpy as np
import pytest
import torch

import samrs_amd
from samrs_amd import driver, synth
from samrs_amd.transforms import ResizeLongestSide
from oracle import sam_oracle as so


def test_registry_names_match_reference():
    # Generate Dataset/segment_anything/build_sam.py:47-52
    for k in ("default", "vit_h", "vit_l", "vit_b"):
        assert k in samrs_amd.sam_model_registry
    assert samrs_amd.sam_model_registry["default"] is samrs_amd.sam_model_registry["vit_h"]


def test_sam_refuses_cpu_device():
    sam = samrs_amd.sam_model_registry["vit_tiny"]()
    assert sam.image_encoder.img_size == 1024 and sam.mask_threshold == 0.0 and sam.image_format == "RGB"
    with pytest.raises(RuntimeError):
        sam.to("cpu")
    with pytest.raises(RuntimeError):
        samrs_amd.SamPredictor(sam)          # not on a device yet -> loud failure, no fallback


@pytest.mark.parametrize("hw", [(1024, 1024), (800, 800), (600, 800), (1500, 1000), (333, 1024)])
def test_resize_longest_side_matches_oracle(hw):
    t = ResizeLongestSide(1024)
    assert t.get_preprocess_shape(hw[0], hw[1], 1024) == so.get_preprocess_shape(hw[0], hw[1], 1024)
    g = torch.Generator().manual_seed(1)
    boxes = torch.rand(7, 4, generator=g, dtype=torch.float64) * min(hw)      # DOTA loader yields float64
    keep = boxes.clone()
    out = t.apply_boxes_torch(boxes, hw)
    assert out.dtype == torch.float32 and torch.equal(out, so.apply_boxes(boxes, hw))
    assert torch.equal(boxes, keep)                                            # inputs are not mutated
    pts = np.random.default_rng(0).uniform(0, min(hw), (5, 2))
    np.testing.assert_allclose(t.apply_coords(pts, hw), so.apply_coords(torch.from_numpy(pts), hw).numpy(), rtol=1e-6)
    img = synth.make_noise_image(0, *hw)
    np.testing.assert_array_equal(t.apply_image(img), so.apply_image(img))


def test_state_dict_factory_matches_contract():
    sd = synth.make_state_dict(synth.CONFIGS["vit_b"], 0)
    assert sd["image_encoder.blocks.2.attn.rel_pos_h"].shape == (127, 64)      # global block
    assert sd["image_encoder.blocks.0.attn.rel_pos_h"].shape == (27, 64)       # windowed block
    assert sd["mask_decoder.output_upscaling.0.weight"].shape == (256, 64, 2, 2)
    assert sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].shape == (2, 128)
    n_enc = sum(v.numel() for k, v in sd.items() if k.startswith("image_encoder"))
    n_dec = sum(v.numel() for k, v in sd.items() if k.startswith("mask_decoder"))
    assert n_enc == 89_670_912 and n_dec == 4_058_340                          # SURVEY.md section 8 [measured]
    sd2 = synth.make_state_dict(synth.CONFIGS["vit_b"], 0)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)                         # reproducible


def test_driver_chunking_and_sharding():
    assert driver.box_chunks(32, 20) == so.box_chunks(32, 20) == [(0, 20), (20, 32)]
    assert driver.box_chunks(40, 20) == so.box_chunks(40, 20)
    assert driver.box_chunks(0, 20) == []
    files = [f"P{i:04d}.png" for i in (5, 3, 9, 1, 7, 2, 8, 0, 6, 4)]
    parts = [driver.shard(files, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == sorted(files) and parts[0] == ["P0000.png", "P0004.png", "P0008.png"]


@pytest.mark.parametrize("hw,out", [((600, 800), (768, 1024)), ((800, 800), (1024, 1024)), ((1500, 1000), (1024, 683)),
                                    ((333, 1024), (333, 1024)), ((2048, 1365), (1024, 683))])
def test_pil_resample_restatement_is_bit_exact(hw, out):
    """The coefficient tables + integer passes used by the GPU resize (N3) reproduce PIL's BILINEAR
    resize (what ResizeLongestSide.apply_image calls, utils/transforms.py:26-31) bit for bit."""
    from PIL import Image
    from samrs_amd.transforms import resample_reference
    img = synth.make_image(7, *hw)
    ref = np.array(Image.fromarray(img).resize((out[1], out[0]), Image.BILINEAR))
    got = resample_reference(img, out[0], out[1])
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"{(got != ref).sum()} differing bytes"


def test_mean_iou_matches_reference_arithmetic():
    """driver.mean_iou == main_sam_rhbox_mask_instance.py:222-241 (float products / sums of 0-1 arrays)."""
    from samrs_amd import driver
    rng = np.random.default_rng(4)
    preds = [rng.random((3, 20, 30)) > 0.5, rng.random((2, 20, 30)) > 0.7]
    gts = [rng.random((3, 20, 30)) > 0.5, np.zeros((2, 20, 30), bool)]
    preds[1][1] = False                                  # empty union -> skipped, like the reference
    avg, area = driver.mean_iou(preds, gts)
    ious, inter, union = [], [], []
    for pm, gm in zip(preds, gts):
        for j in range(pm.shape[0]):
            g = gm[j].reshape(-1).astype(float)
            p = pm[j].reshape(-1).astype(float)
            i = float(np.sum(g * p))
            u = float(np.sum(np.array(g + p > 0)))
            if u > 0:
                inter.append(i); union.append(u); ious.append(i / u)
    assert avg == pytest.approx(np.mean(ious)) and area == pytest.approx(np.sum(inter) / np.sum(union))


def test_bench_parity_object_is_pinned_to_the_device_sources(tmp_path, monkeypatch):
    """bench.py's `parity` object (the line carried a throughput with no parity figure attached to the mode that
    produced it) reports the committed statistical sample only when it was measured on THIS build's device sources."""
    import json
    import bench
    s = {"n_masks": 256, "iou_min": 0.99961, "flips_outside_tau": 0, "classmap_diff_mean": 457.5, "classmap_diff_max": 532,
         "classmap_diff_outside_unstable": 0}
    c4 = {"n_masks": 96, "iou_min": 0.9989, "flips_outside_tau": 0}
    f = tmp_path / "parity_stats.json"
    monkeypatch.setattr(bench, "PARITY_FILE", str(f))
    assert "note" in bench.parity_of_mode(15, "c2")                                     # no file
    json.dump({"tau_frac": 0.0025, "csrc_sha16": "0" * 16, "summary": {"15": {"c2": s, "c4box": c4, "c4mask": c4}}}, open(f, "w"))
    assert "other device sources" in bench.parity_of_mode(15, "c2")["note"]             # stale: not reported
    json.dump({"tau_frac": 0.0025, "csrc_sha16": bench.csrc_sha(), "summary": {"15": {"c2": s, "c4box": c4, "c4mask": c4}}}, open(f, "w"))
    p = bench.parity_of_mode(15, "c2")
    assert p["mode"] == 15 and p["c2_iou_min"] == 0.99961 and p["classmap_px_mean"] == 457.5 and p["c4_iou_min"] == 0.9989
    assert p["classmap_bit_identical"] is False and p["c4_served_in_this_mode"] is False and p["n_masks"] == 448
    assert "note" in bench.parity_of_mode(79, "c2")                                     # a mode the sample does not hold


def test_gelu_as28_error_budget():
    """The lin1 epilogue's GELU (csrc/common.h gelu_erf2_et: erf by Abramowitz-Stegun 7.1.28, one rcp, no exp2) restated in
    fp32 numpy, operation by operation, against the exact function (`nn.GELU()`, common.py:18-26) in fp64: absolute error
    < 1e-6 everywhere, < 1.5e-7 rms under N(0, 1), and three orders of magnitude under the f16 rounding its output receives."""
    from scipy.special import erfc
    f = np.float32
    x = np.linspace(-12, 12, 600001).astype(f)
    exact = x.astype(np.float64) * 0.5 * erfc(-x.astype(np.float64) / np.sqrt(2.0))
    u = np.abs(x)
    p = (u * f(5.38297490493278e-06) + f(4.889063711743802e-05)).astype(f)
    for c in (3.8003574445610866e-05, 0.0032776263542473316, 0.02114100567996502, 0.04986734688282013, 1.0):
        p = (p * u + f(c)).astype(f)
    r = (f(1) / p).astype(f)
    for _ in range(4):
        r = (r * r).astype(f)
    got = (((u * (f(1) - r)).astype(f) + x).astype(f) * f(0.5)).astype(f)
    err = got.astype(np.float64) - exact
    w = np.exp(-x.astype(np.float64) ** 2 / 2)
    rms = np.sqrt((w * err ** 2).sum() / w.sum())
    f16 = exact.astype(np.float16).astype(np.float64) - exact
    rms16 = np.sqrt((w * f16 ** 2).sum() / w.sum())
    assert np.abs(err).max() < 1e-6 and rms < 1.5e-7 and rms16 > 500 * rms, (np.abs(err).max(), rms, rms16)
    assert got[0] == 0.0 and got[-1] == x[-1]                       # the tails are exactly max(x, 0)


def test_bench_gpus_n_without_a_launcher_becomes_the_launcher(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE) re-executes itself under torch.distributed.run with
    the same arguments, one rank per GPU, rendezvous on 127.0.0.1; under a launcher (WORLD_SIZE set) and for N = 1 it does not."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, *a, **k):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench._self_launch(4)
    assert ex.value.code == 7                                           # the launcher's status is the script's status
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]                  # same arguments, verbatim


def test_fill_rule_resolution(monkeypatch):
    """the rbox rasteriser reproduces one of the two cv2.fillPoly span rules by name; "auto" follows the cv2 that is
    installed (none here: the older rule, or SAMRS_FILL_RULE), an unknown name is refused."""
    import sys
    import types
    from samrs_amd import transforms
    monkeypatch.delenv("SAMRS_FILL_RULE", raising=False)
    monkeypatch.setitem(sys.modules, "cv2", None)                      # "import cv2" raises ImportError
    assert transforms.resolve_fill_rule("auto") == "cv2_le_451"
    assert transforms.resolve_fill_rule("cv2_ge_452") == "cv2_ge_452"
    monkeypatch.setenv("SAMRS_FILL_RULE", "cv2_ge_452")
    assert transforms.resolve_fill_rule("auto") == "cv2_ge_452"
    monkeypatch.delenv("SAMRS_FILL_RULE")
    for ver, want in (("4.5.1", "cv2_le_451"), ("4.5.2", "cv2_ge_452"), ("4.10.0.84", "cv2_ge_452"), ("3.4.18", "cv2_le_451")):
        monkeypatch.setitem(sys.modules, "cv2", types.SimpleNamespace(__version__=ver))
        assert transforms.resolve_fill_rule("auto") == want, ver
    with pytest.raises(ValueError):
        transforms.resolve_fill_rule("cv2_5")


def test_outlier_rule_on_the_host_finds_the_planted_columns(tmp_path, capsys):
    """samrs_amd/outliers.py -- the host-side statement of the rule the engine applies on the device when it loads weights (the GPU suite
    asserts that the two pick the same columns): nothing on seeded-normal weights, exactly the planted channels on synth.heavy_tailed
    (LayerNorm gammas -> qkv / lin1 columns, hidden units -> lin2, v channels -> proj), each block outlier-dominated; and the CLI a user
    points at a checkpoint file."""
    from samrs_amd import outliers
    cfg = synth.CONFIGS["vit_tiny"]
    base = synth.make_state_dict(cfg, 0)
    assert all(len(idx) == 0 for idx, _ in outliers.outlier_columns(base, cfg).values())
    sd = synth.heavy_tailed(base, cfg, 0, hidden_scale=3e3, v_scale=3e3, gamma_scale=30.0)
    oc = outliers.outlier_columns(sd, cfg)
    gen = torch.Generator().manual_seed(424242)                   # synth.heavy_tailed's own draw order: hidden, v, gamma per block
    for i in range(cfg.depth):
        hid = torch.randperm(4 * cfg.embed_dim, generator=gen)[:4]
        vch = torch.randperm(cfg.embed_dim, generator=gen)[:4]
        gch = torch.randperm(cfg.embed_dim, generator=gen)[:4]
        assert oc[(i, "qkv")][0].tolist() == sorted(gch.tolist()) == oc[(i, "lin1")][0].tolist()
        assert oc[(i, "lin2")][0].tolist() == sorted(hid.tolist()) and oc[(i, "proj")][0].tolist() == sorted(vch.tolist())
        assert all(oc[(i, g)][1] > 0.5 for g in outliers.GEMMS)
    path = tmp_path / "heavy.pth"
    torch.save(sd, path)
    assert outliers.main([str(path), "--model", "vit_tiny"]) == 0
    out = capsys.readouterr().out
    assert "32 outlier columns in all; 2 of 2 blocks outlier-dominated" in out


def test_outlier_rule_caps_at_32_columns_per_gemm():
    """More candidates than one 64-wide K stage can carry (32 lo + 32 hi columns): the 32 largest scores are kept, ascending."""
    from samrs_amd import outliers
    cfg = synth.CONFIGS["vit_tiny"]
    sd = synth.heavy_tailed(synth.make_state_dict(cfg, 0), cfg, 0, gamma_scale=30.0, n_channels=40, blocks=[0])
    sc = outliers.block_scores(sd, cfg, 0)["qkv"]
    idx, share = outliers.pick(sc)
    assert len(idx) == 32 and idx.tolist() == sorted(idx.tolist()) and share > 0.5
    assert int((sc > 4 * sc.median()).sum()) == 40
    kept = set(idx.tolist())
    dropped = [c for c in torch.nonzero(sc > 4 * sc.median()).flatten().tolist() if c not in kept]
    assert len(dropped) == 8 and max(float(sc[c]) for c in dropped) <= min(float(sc[c]) for c in kept)
