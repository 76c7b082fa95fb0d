"""Pins the oracle (oracle/unetpp_oracle.py) to the reference: every committed golden vector was
produced by the reference NestedUNet itself (oracle/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from conftest import load_golden

SMALL = ["s_c3_32x32", "s_c3_64x64", "s_c7_48x80", "s_c3_128x96"]


def _inputs(syn, g):
    frames = syn.make_frames_u8(int(g["B"]), int(g["H"]), int(g["W"]), str(g["kind"]), int(g["fseed"]))
    assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g["frames_sha"])
    sd = syn.make_state_dict(int(g["num_classes"]), 3, bool(g["deep_supervision"]), int(g["wseed"]))
    return sd, syn.frames_to_chw_f32(frames)


@pytest.mark.parametrize("tag", SMALL)
def test_torch_restatement_matches_reference(tag, oracle, syn):
    g = load_golden(tag)
    sd, x = _inputs(syn, g)
    logits = oracle.torch_forward(sd, x)
    # same ops, same library: equal to a few ulp (thread count may differ from the generating run)
    np.testing.assert_allclose(logits, g["logits"], rtol=0, atol=2e-6)
    pred, cable, tape = oracle.masks_from_logits(logits)
    assert np.array_equal(pred, g["mask"])
    assert np.array_equal(cable, g["mask_cable"]) and np.array_equal(tape, g["mask_tape"])


@pytest.mark.parametrize("tag", SMALL[:3])
def test_numpy_restatement_matches_reference(tag, oracle, syn):
    g = load_golden(tag)
    sd, x = _inputs(syn, g)
    logits = oracle.numpy_forward(sd, x, dtype=np.float32)
    np.testing.assert_allclose(logits, g["logits"], rtol=0, atol=2e-5)
    pred, _, _ = oracle.masks_from_logits(logits)
    margin = oracle.top2_margin(g["logits"])
    differs = pred != g["mask"]
    assert not differs[margin > 1e-4].any()


def test_numpy_fp64_restatement(oracle, syn):
    g = load_golden("s_c3_32x32")
    sd, x = _inputs(syn, g)
    logits = oracle.numpy_forward(sd, x, dtype=np.float64)
    np.testing.assert_allclose(logits, g["logits"], rtol=0, atol=5e-6)


def test_intermediates_match_reference(oracle, syn):
    g = load_golden("s_c3_32x32")
    sd, x = _inputs(syn, g)
    _, t_np = oracle.numpy_forward(sd, x, return_intermediates=True)
    _, t_th = oracle.torch_forward(sd, x, return_intermediates=True)
    for name in ("x0_0", "x1_0", "x2_0", "x3_0", "x4_0", "x3_1", "x2_2", "x1_3", "x0_4"):
        ref = g["t_" + name]
        np.testing.assert_allclose(t_th[name], ref, rtol=0, atol=2e-6, err_msg=name)
        np.testing.assert_allclose(t_np[name], ref, rtol=0, atol=3e-5, err_msg=name)


def test_full_size_golden_512(oracle, syn):
    g = load_golden("b_c3_512x512")
    kinds = [str(k) for k in g["kinds"]]
    frames = np.stack([syn.make_frame_u8(512, 512, i, kinds[i % len(kinds)], int(g["fseed"])) for i in range(int(g["B"]))])
    assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g["frames_sha"])
    sd = syn.make_state_dict(3, 3, True, int(g["wseed"]))
    logits = oracle.torch_forward(sd, syn.frames_to_chw_f32(frames))
    np.testing.assert_allclose(logits[:, :, ::8, ::8], g["logits_sub8"], rtol=0, atol=5e-6)
    pred, _, _ = oracle.masks_from_logits(logits)
    diff = np.argwhere(pred != g["mask"])
    ties = {tuple(t) for t in g["tie_idx"].tolist()}
    assert all(tuple(d) in ties for d in diff.tolist())     # only near-tie pixels may differ
    assert len(diff) <= 2


def test_edge_cases(oracle, syn):
    sd = syn.make_state_dict(3, 3, True, 2)
    with pytest.raises(RuntimeError):                         # reference raises in torch.cat
        oracle.numpy_forward(sd, np.zeros((1, 3, 100, 100), np.float32))
    with pytest.raises(RuntimeError):
        oracle.torch_forward(sd, np.zeros((1, 3, 24, 32), np.float32))
    # batch-row invariance: frame i of a batch equals the same frame alone
    x = syn.frames_to_chw_f32(syn.make_frames_u8(3, 32, 48, "uniform", 5))
    full = oracle.torch_forward(sd, x)
    one = oracle.torch_forward(sd, x[1:2])
    np.testing.assert_allclose(full[1:2], one, rtol=0, atol=1e-5)
    # argmax tie rule: first maximal index (np.argmax, infer_two_stage_burr.py:300)
    lg = np.zeros((1, 3, 2, 2), np.float32)
    pred, cable, tape = oracle.masks_from_logits(lg)
    assert (pred == 0).all() and cable.sum() == 0 and tape.sum() == 0


def test_bilinear_tables(oracle):
    i0, i1, l0, l1 = oracle.bilinear_axis_tables(4, 8)
    assert i0[0] == 0 and l1[0] == 0 and i0[-1] == 3 and i1[-1] == 3     # corners align, last clamps
    assert np.allclose(l0 + l1, 1)


RULE_CASES = [("thr", "thresholded_argmax", dict(t_cable=0.45, t_tape=0.50, bg_margin=0.15)),
              ("thr_strict", "thresholded_argmax", dict(t_cable=0.60, t_tape=0.65, bg_margin=0.30)),
              ("bgcheck", "strict_bg_check", dict(t_cable=0.6, t_tape=0.65, bg_margin=0.4)),
              ("excl", "exclusive", dict(t_cable=0.55, t_tape=0.60, bg_margin=0.20, ct_margin=0.10)),
              ("excl_loose", "exclusive", dict(t_cable=0.34, t_tape=0.34, bg_margin=0.0, ct_margin=0.0))]


@pytest.mark.parametrize("tag", ["s_c3_64x64", "s_c3_128x96"])
def test_probability_rules_match_reference_scripts(tag, oracle):
    """softmax_np + the three rule families, pinned by masks the reference's own functions produced
    (infer_video_3class_best / _strict / _fixed / _robust, imported with a cv2 stub by make_golden.py)."""
    g = load_golden(tag)
    for key, rule, params in RULE_CASES:
        cable, tape, probs = oracle.rule_masks_from_logits(g["logits"], rule, **params)
        assert np.array_equal(cable, g[f"rule_{key}_cable"]), key
        assert np.array_equal(tape, g[f"rule_{key}_tape"]), key
        assert not (cable & tape).any()
    np.testing.assert_allclose(probs, g["probs_hwc"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["s_c3_64x64", "s_c3_128x96"])
def test_mask_statistics_match_reference_function(tag, oracle):
    """Per-row widths pinned by the reference's own _compute_width_per_row (src/utils/geometry_enhanced.py,
    imported with a cv2 stub, smooth=False) on the golden argmax masks."""
    g = load_golden(tag)
    counts, widths = oracle.mask_stats_np(g["mask"], 3)
    np.testing.assert_array_equal(widths[:, 1], g["rowwidth_cable"])
    np.testing.assert_array_equal(widths[:, 2], g["rowwidth_tape"])
    assert counts.sum() == g["mask"].size and counts[:, 1].sum() == int(g["mask_cable"].sum())


SIMPLE = ["su_c7_32x48", "su_c3_64x40", "su_c7_256x256"]


@pytest.mark.parametrize("tag", SIMPLE)
def test_simple_unet_restatements_match_reference(tag, oracle, syn):
    """SimpleUNet oracle (torch and NumPy) pinned by the reference class's own outputs."""
    g = load_golden(tag)
    B, H, W, C = int(g["B"]), int(g["H"]), int(g["W"]), int(g["num_classes"])
    frames = syn.make_frames_u8(B, H, W, str(g["kind"]), int(g["fseed"]))
    assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g["frames_sha"])
    sd = syn.make_simple_state_dict(C, 3, int(g["wseed"]))
    x = syn.frames_to_chw_f32(frames)
    lt, tt = oracle.simple_unet_torch_forward(sd, x, return_intermediates=True)
    if "logits" in g.files:
        np.testing.assert_allclose(lt, g["logits"], rtol=0, atol=5e-6)
        ln, tn = oracle.simple_unet_numpy_forward(sd, x, return_intermediates=True)
        np.testing.assert_allclose(ln, g["logits"], rtol=0, atol=5e-5)
        for k in [f for f in g.files if f.startswith("t_")]:
            np.testing.assert_allclose(tt[k[2:]], g[k], rtol=0, atol=5e-6, err_msg=k)
            np.testing.assert_allclose(tn[k[2:]], g[k], rtol=0, atol=5e-5, err_msg=k)
    else:
        np.testing.assert_allclose(lt[:, :, ::4, ::4], g["logits_sub4"], rtol=0, atol=1e-5)
    probs = oracle.softmax_np(lt, axis=1)
    ref_p = g["probs"] if "probs" in g.files else None
    if ref_p is not None:
        np.testing.assert_allclose(probs, ref_p, rtol=0, atol=2e-6)
    differs = np.argwhere(np.argmax(probs, axis=1) != g["mask"])
    assert len(differs) <= 2


def test_simple_unet_manifest_and_blob(syn):
    import json
    from conftest import ROOT
    from unet_amd import _lib, packing
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))["simple_c7"]
    assert [[k, list(s), d] for k, s, d in syn.simple_unet_manifest(7, 3)] == man
    sd = syn.make_simple_state_dict(7, 3, 0)
    blob = packing.build_simple_blob(sd, 7)
    assert blob.nbytes == _lib.load().unetpp_weights_blob_bytes_arch(1, 7, 3)
    assert blob[:32].view(np.uint32)[5] == 1
    with pytest.raises(RuntimeError, match="Missing key"):
        bad = dict(sd); bad.pop("up2.bias"); packing.check_simple_state_dict(bad, 7)


# ---- frame glue ------------------------------------------------------------------------
def test_roi_mapping_matches_reference_function(oracle):
    """map_roi_to_original (infer_two_stage_burr.py:37-47): oracle restatement and the host mirror against
    values the reference's own function returned (tests/golden/roi_map.json)."""
    import json
    from unet_amd import frame_loop
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "roi_map.json")))
    assert g["fixed_roi_512"] == frame_loop.FIXED_ROI_512
    assert tuple(g["fixed_roi_512"][k] for k in ("x1", "y1", "x2", "y2")) == oracle.FIXED_ROI_512
    for c in g["cases"]:
        assert list(oracle.map_roi_to_original_np(tuple(c["size"]), tuple(c["target"]))) == c["roi"]
        assert list(frame_loop.map_roi_to_original(tuple(c["size"]), tuple(c["target"]))) == c["roi"]


def test_cv2_linear_restatement_properties(oracle):
    """PARITY UNPINNED (cv2 absent): the INTER_LINEAR restatement is checked for the properties OpenCV's
    fixed-point algorithm has and against torch's float bilinear (same half-pixel geometry) to 1 LSB."""
    import torch
    rng = np.random.default_rng(5)
    for (sh, sw, c), (dw, dh) in [((108, 192, 3), (64, 64)), ((40, 30, 1), (90, 70)), ((33, 77, 3), (77, 33))]:
        img = rng.integers(0, 256, (sh, sw, c), dtype=np.uint8)
        out = oracle.cv2_resize_linear_u8_np(img, (dw, dh))
        assert out.shape == (dh, dw, c) and out.dtype == np.uint8
        t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(out.astype(np.float32) - ref).max() < 1.0
    img = rng.integers(0, 256, (21, 17, 3), dtype=np.uint8)
    assert np.array_equal(oracle.cv2_resize_linear_u8_np(img, (17, 21)), img)            # same size = identity
    flat = np.full((9, 11), 201, np.uint8)
    assert np.all(oracle.cv2_resize_linear_u8_np(flat, (40, 23)) == 201)                 # constants survive
    s0, s1, a0, a1 = oracle.cv2_linear_tables(1920, 512)
    assert np.all(a0 + a1 == 2048) and s0.min() >= 0 and s1.max() <= 1919 and np.all(s1 - s0 <= 1)
    # worked example, 2 -> 4 samples: fx = -0.25, 0.25, 0.75, 1.25 -> (s, a1) = (0, 0), (0, 512), (0, 1536), (1, 0)
    s0, s1, a0, a1 = oracle.cv2_linear_tables(2, 4)
    assert s0.tolist() == [0, 0, 0, 1] and a1.tolist() == [0, 512, 1536, 0]
    row = np.array([[0, 255]], np.uint8)
    assert oracle.cv2_resize_linear_u8_np(row, (4, 1)).tolist() == [[0, 64, 191, 255]]


def test_cv2_nearest_restatement_and_roi_clip(oracle):
    """INTER_NEAREST = floor(dst * src/dst) (torch documents its mode='nearest' as matching OpenCV's
    INTER_NEAREST); ROI clip = Python slice semantics (infer_two_stage_burr.py:311-314)."""
    import torch
    rng = np.random.default_rng(6)
    for (sh, sw), (dw, dh) in [((512, 512), (1920, 1080)), ((64, 96), (333, 127)), ((128, 128), (50, 30))]:
        m = rng.integers(0, 3, (sh, sw), dtype=np.uint8)
        out = oracle.cv2_resize_nearest_np(m, (dw, dh))
        ref = torch.nn.functional.interpolate(torch.from_numpy(m)[None, None].float(), size=(dh, dw), mode="nearest")
        assert np.array_equal(out, ref[0, 0].numpy().astype(np.uint8))
    full = np.ones((30, 40), np.uint8)
    c = oracle.clip_to_roi_np(full, (5, 2, 100, 7))
    assert c.sum() == 35 * 5 and c[2:7, 5:].all() and not c[:2].any()
    cable, tape = oracle.postprocess_masks_np(np.array([[0, 1], [2, 1]], np.uint8), (4, 4), (0, 0, 4, 2))
    assert cable.tolist() == [[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]]
    assert tape.tolist() == [[0] * 4] * 4


def test_config2_b16_and_config5_fixtures_pin_the_oracle(oracle, syn):
    """The two full-batch fixtures: the oracle reproduces the reference's subsampled logits, masks (up to listed
    near-ties) and sampled intermediates on a subset of frames (the whole 16 would take a minute here)."""
    g = load_golden("b_c3_512x512_b16")
    kinds = [str(k) for k in g["kinds"]]
    idx = [0, 1, 9]
    frames = np.stack([syn.make_frame_u8(512, 512, i, kinds[i % len(kinds)], int(g["fseed"])) for i in idx])
    sd = syn.make_state_dict(3, 3, True, int(g["wseed"]))
    logits, inter = oracle.torch_forward(sd, syn.frames_to_chw_f32(frames), return_intermediates=True)
    np.testing.assert_allclose(logits[:, :, ::8, ::8], g["logits_sub8"][idx], rtol=0, atol=5e-6)
    pred, _, _ = oracle.masks_from_logits(logits)
    ties = {tuple(t) for t in g["tie_idx"].tolist()}
    for j, i in enumerate(idx):
        assert all((i, y, x) in ties for y, x in np.argwhere(pred[j] != g["mask"][i]).tolist())
    for name in ("x0_0", "x1_0", "x2_0", "x3_0", "x4_0", "x3_1", "x2_2", "x1_3", "x0_4"):
        ys, xs = g["p_" + name][:, 0], g["p_" + name][:, 1]
        np.testing.assert_allclose(inter[name][:2][:, :, ys, xs], g["t_" + name], rtol=0, atol=5e-6, err_msg=name)
    assert g["mask"].shape == (16, 512, 512) and sorted(set(kinds)) == ["smooth", "uniform"]
