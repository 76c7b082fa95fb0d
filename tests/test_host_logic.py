"""CPU-side checks: ABI library loads and exports every declared symbol, BN folding, strict
state-dict checking, blob layout.  No compute call is made (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_builds_and_exports_all_symbols():
    from unet_amd import _lib
    path = _lib.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "unetpp.h")).read()
    declared = set(re.findall(r"\b(unetpp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.unetpp_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.unetpp_version()


def test_blob_size_matches_abi(syn):
    from unet_amd import _lib, packing
    lib = _lib.load()
    for C, ds in ((3, True), (7, False)):
        sd = syn.make_state_dict(C, 3, ds, 0)
        blob = packing.build_blob(sd, C)
        assert blob.nbytes == lib.unetpp_weights_blob_bytes(C, 3)
        hdr = blob[:32].view(np.uint32)
        assert hdr[0] == packing.BLOB_MAGIC and hdr[2] == C and hdr[4] == 19 and hdr[5] == 0
    # 7,846,723 folded conv weights+biases for the 3-class net
    assert (lib.unetpp_weights_blob_bytes(3, 3) - 32) // 4 == 7846723


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from unet_amd import _lib
    lib = _lib.load()
    cfg = _lib.Config(3, 3, 1, 32, 32, 0, 0, 0, 1, 0)
    h = ctypes.c_void_p()
    rc = lib.unetpp_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"no CPU fallback" in lib.unetpp_last_error(None) or b"HIP" in lib.unetpp_last_error(None)
    from unet_amd.nested_unet import NestedUNet
    with pytest.raises(RuntimeError):
        NestedUNet(3).to("cpu")


def test_bn_fold_equals_conv_bn(syn, oracle):
    from unet_amd import packing
    sd = syn.make_state_dict(3, 3, True, 5)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 32, 12, 10)).astype(np.float64)
    name = "conv1_0"
    ref = oracle.conv3x3_np(x, sd[f"{name}.conv1.weight"].astype(np.float64), sd[f"{name}.conv1.bias"].astype(np.float64))
    ref = oracle.batchnorm_eval_np(ref, sd[f"{name}.bn1.weight"], sd[f"{name}.bn1.bias"], sd[f"{name}.bn1.running_mean"], sd[f"{name}.bn1.running_var"])
    wf, bf = packing.fold_conv_bn(sd[f"{name}.conv1.weight"], sd[f"{name}.conv1.bias"], sd[f"{name}.bn1.weight"],
                                  sd[f"{name}.bn1.bias"], sd[f"{name}.bn1.running_mean"], sd[f"{name}.bn1.running_var"])
    got = oracle.conv3x3_np(x, wf.astype(np.float64), bf.astype(np.float64))
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_strict_state_dict_checks(syn):
    from unet_amd import packing
    sd = syn.make_state_dict(3, 3, True, 0)
    assert packing.check_state_dict(sd, 3, 3, True, strict=True) == ([], [])
    extra = dict(sd); extra["bogus.weight"] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        packing.check_state_dict(extra, 3, 3, True, strict=True)
    assert packing.check_state_dict(extra, 3, 3, True, strict=False) == ([], ["bogus.weight"])
    # a deep_supervision=True checkpoint loaded into a ds=False model: ds heads are unexpected (strict)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        packing.check_state_dict(sd, 3, 3, False, strict=True)
    assert packing.unwrap_checkpoint({"model": sd, "epoch": 3}) is sd
    assert packing.unwrap_checkpoint({"model_state_dict": sd}) is sd
    assert packing.infer_num_classes(sd) == 3


def test_manifest_matches_reference_fixture(syn):
    import json
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))
    for key, (C, ds) in (("c3_ds1", (3, True)), ("c7_ds0", (7, False))):
        mine = [[k, list(s), d] for k, s, d in syn.state_dict_manifest(C, 3, ds)]
        assert mine == man[key]
        assert len(mine) == (134 if ds else 128)


def test_frame_preprocess_matches_reference_semantics(syn):
    from unet_amd.frame_loop import preprocess_frames
    f = syn.make_frames_u8(2, 16, 32, "uniform", 9)
    x = preprocess_frames(f)
    assert x.shape == (2, 3, 16, 32) and x.dtype == np.float32
    assert x[1, 0, 3, 5] == np.float32(f[1, 3, 5, 2]) / np.float32(255.0)      # R plane comes from BGR index 2
    assert np.array_equal(x, syn.frames_to_chw_f32(f))


def _ac_rows(n_out):
    """float32 index math of the bilinear x2 upsample, align_corners=True, exactly as the kernels compute it
    (upsample2x_kernel / conv3x3_ws.h / tapmm_ws.h): src = dst * (in-1)/(out-1) in float32, i0 = int(src)."""
    n_in = n_out // 2
    s = np.float32(n_in - 1) / np.float32(n_out - 1) if n_in > 1 else np.float32(0)
    f = (s * np.arange(n_out, dtype=np.float32)).astype(np.float32)
    i0 = np.minimum(f.astype(np.int32), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    return s, i0, i1


@pytest.mark.parametrize("n_out", list(range(16, 1200, 16)) + [2048, 4096])
def test_fused_upsample_index_claims(n_out):
    """The geometry the fused-upsample loaders rely on (conv3x3_mfma.h UPF, conv3x3_ws.h, tapmm_ws.h upsum_kernel):
    for tile origins at multiples of 16 (rows) / 32 (columns) of an n_out-long axis,
      * an 8- / 16-row (32-column) tile with its 1-pixel halo touches at most 6 / 10 (18) low-res rows (columns) counted from
        floor(s * max(origin - 1, 0)), and the 16+2 rows of an upsum tile at most 11;
      * the two image rows (columns) of every 2x2 halo block — an odd one and the even one after it — share their low-res
        corner pair, so one producer lane can serve the block from four records."""
    s, i0, i1 = _ac_rows(n_out)
    for tile, lim in ((8, 6), (16, 10), (32, 18)):
        for o in range(0, n_out, tile):
            lo, hi = max(o - 1, 0), min(o + tile, n_out - 1)
            base = int(np.float32(s * np.float32(lo)))
            assert i0[lo:hi + 1].min() >= base
            assert i1[lo:hi + 1].max() - base + 1 <= lim
    for o in range(0, n_out, 16):                                   # upsum_kernel: rows o-1 .. o+16 of a 16-row tile
        lo, hi = max(o - 1, 0), min(o + 16, n_out - 1)
        assert i1[lo:hi + 1].max() - int(np.float32(s * np.float32(lo))) + 1 <= 11
    odd = np.arange(1, n_out - 1, 2)
    assert np.array_equal(i0[odd], i0[odd + 1]) and np.array_equal(i1[odd], i1[odd + 1])


def test_precision_codes_match_the_header():
    """include/unetpp.h's UNETPP_PREC_* enum and the Python binding's table are the same numbers."""
    import re
    from conftest import ROOT
    from unet_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "unetpp.h")).read()
    m = re.search(r"enum \{ UNETPP_PREC_EXACT = (\d+), UNETPP_PREC_FAST = (\d+), UNETPP_PREC_EXACT8 = (\d+) \};", hdr)
    assert m, "precision enum not found in include/unetpp.h"
    assert (_lib.PRECISIONS["exact"], _lib.PRECISIONS["fast"], _lib.PRECISIONS["exact8"]) == tuple(int(g) for g in m.groups())
    from unet_amd.nested_unet import NestedUNet
    import pytest
    with pytest.raises(ValueError):
        NestedUNet(3, precision="exact16")
    assert NestedUNet(3, precision="exact8").precision == "exact8"      # construction needs no device


def test_reciprocal_tile_decode_is_off_by_at_most_one():
    """conv3x3_ws.h decodes a workgroup's first tile number with one float reciprocal per radix digit and a single
    correction step (its `divmod`), valid below 2^22 tiles: whatever way v_rcp_f32 rounds (1 ulp), the truncated float
    quotient is within one of the true quotient, so the corrected (q, r) are exact."""
    rng = np.random.default_rng(7)
    u = np.concatenate([rng.integers(0, 1 << 22, 200_000), np.arange(0, 4096), (1 << 22) - 1 - np.arange(0, 4096)]).astype(np.int64)
    d = np.concatenate([rng.integers(1, 1 << 14, 200_000), rng.integers(1, 64, 8192)]).astype(np.int64)
    rcp = (np.float32(1.0) / d.astype(np.float32)).astype(np.float32)
    for bump in (-1, 0, 1):                                   # the hardware reciprocal: correctly rounded +- 1 ulp
        r32 = (rcp.view(np.int32) + bump).view(np.float32)
        q = (u.astype(np.float32) * r32).astype(np.float32).astype(np.int64)          # v_cvt_i32_f32 truncates
        r = u - q * d
        up, dn = (r >= d).astype(np.int64), (r < 0).astype(np.int64)
        q2, r2 = q + up - dn, r + (dn - up) * d
        assert np.array_equal(q2, u // d) and np.array_equal(r2, u % d)
