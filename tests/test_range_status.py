"""The engine stores activations as fp16 planes; the fp32 reference (src/models/unetpp.py:17-26,
src/models/simple_unet.py:94-128) has no 65504 ceiling and propagates NaN.  Values that do not fit are never
narrowed silently: the kernel sets a sticky flag (include/unetpp.h unetpp_status).  Also: BatchNorm statistics of
the kind only trained checkpoints have (tiny running_var, negative / zero gamma) must fold correctly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OVERFLOW, NAN = 1, 2


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    return torch


def test_trained_like_bn_statistics_match_oracle(torch_cuda, syn, oracle):
    """(c) running_var = 1e-8 on near-dead channels, gamma < 0, gamma = 0, large gamma: parity with the oracle, no flag."""
    torch = torch_cuda
    from unet_amd.nested_unet import NestedUNet
    sd = syn.make_trained_like_state_dict(3, 3, True, 2)
    x = syn.frames_to_chw_f32(syn.make_frames_u8(2, 64, 96, "smooth", 7))
    ref, inter = oracle.torch_forward(sd, x, return_intermediates=True)
    ref_mask, _, _ = oracle.masks_from_logits(ref)
    m = NestedUNet(3, max_batch=2, max_hw=(64, 96)).to("cuda:0")
    m.load_state_dict(sd, strict=True)
    m.debug_keep_intermediates(True)
    logits = m(torch.from_numpy(x).cuda()).cpu().numpy()
    for name in ("x0_0", "x2_0", "x4_0", "x2_2", "x0_4"):
        got = m.debug_activation(name, 2, 64, 96)
        scale = max(1.0, float(np.abs(inter[name]).max()))
        np.testing.assert_allclose(got, inter[name], rtol=0, atol=2e-5 * scale, err_msg=name)
    m.debug_keep_intermediates(False)
    mask = m.segment(torch.from_numpy(x).cuda()).cpu().numpy()
    err = float(np.abs(logits - ref).max())
    print(f"trained-like BN: max|dlogit|={err:.3e}, logits in [{ref.min():.2f},{ref.max():.2f}]")
    assert err < 1e-3 and err < 2e-5 * max(1.0, float(np.abs(ref).max()))
    flips = mask != ref_mask
    assert not (flips & (oracle.top2_margin(ref) > 2 * err + 1e-7)).any()
    assert m.status() == 0


def test_simple_unet_overflow_sets_flag(torch_cuda, syn, oracle):
    """(a) SimpleUNet has no BatchNorm: a large-gain checkpoint drives activations past the fp16 maximum."""
    torch = torch_cuda
    from unet_amd.nested_unet import SimpleUNet
    x = syn.frames_to_chw_f32(syn.make_frames_u8(1, 64, 64, "smooth", 7))
    xt = torch.from_numpy(x).cuda()

    def scaled(gain):
        sd = syn.make_simple_state_dict(7, 3, 0)
        for k in ("enc1.0.weight", "enc1.0.bias"):
            sd[k] = sd[k] * np.float32(gain)
        return sd

    # activations of a few thousand: representable, must match the oracle (relative to their size) and raise no flag
    sd = scaled(1000.0)
    ref, inter = oracle.simple_unet_torch_forward(sd, x, return_intermediates=True)
    assert 1e3 < max(float(np.abs(v).max()) for v in inter.values()) < 6e4
    m = SimpleUNet(7, 3, max_batch=1, max_hw=(64, 64)).to("cuda:0")
    m.load_state_dict(sd, strict=True)
    logits = m(xt).cpu().numpy()
    assert float(np.abs(logits - ref).max()) < 2e-5 * float(np.abs(ref).max())
    assert m.status() == 0

    # the same net 40x louder: enc1 exceeds 65504 -> clamped -> flagged (and kept flagged until cleared)
    sd = scaled(40000.0)
    _, inter = oracle.simple_unet_torch_forward(sd, x, return_intermediates=True)
    assert float(np.abs(inter["enc1"]).max()) > 65504
    m.load_state_dict(sd, strict=True)
    m(xt)
    assert m.status() & OVERFLOW
    assert m.status(clear=True) & OVERFLOW
    assert m.status() == 0
    strict = SimpleUNet(7, 3, max_batch=1, max_hw=(64, 64), check_range=True).to("cuda:0")
    strict.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError, match="value range"):
        strict(xt)


def test_nan_and_huge_input_set_flags(torch_cuda, syn):
    """(b) a NaN pixel / a float32 input beyond the fp16 range."""
    torch = torch_cuda
    from unet_amd.nested_unet import NestedUNet
    sd = syn.make_state_dict(3, 3, True, 2)
    m = NestedUNet(3, max_batch=1, max_hw=(32, 48)).to("cuda:0")
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(1, 32, 48, "smooth", 3))).cuda()
    m.segment(x)
    assert m.status() == 0
    xn = x.clone(); xn[0, 1, 7, 9] = float("nan")
    m.segment(xn)
    assert m.status(clear=True) & NAN
    xb = x.clone(); xb[0, 2, 20, 30] = 1.0e6
    m.segment(xb)
    assert m.status(clear=True) & OVERFLOW
    m.segment(x)                                   # sticky, not stuck: clean again after the clear
    assert m.status() == 0
    # uint8 frames cannot leave the range
    m.segment(torch.from_numpy(syn.make_frames_u8(1, 32, 48, "uniform", 4)).cuda())
    assert m.status() == 0


def test_non_finite_weights_set_flag_at_load(torch_cuda, syn):
    torch = torch_cuda
    from unet_amd.nested_unet import NestedUNet
    sd = syn.make_state_dict(3, 3, True, 2)
    sd["conv2_0.conv1.weight"] = sd["conv2_0.conv1.weight"].copy()
    sd["conv2_0.conv1.weight"][5, 7, 1, 1] = np.float32("nan")
    m = NestedUNet(3, max_batch=1, max_hw=(32, 32)).to("cuda:0")
    m.load_state_dict(sd, strict=True)
    m.segment(torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(1, 32, 32, "smooth", 3))).cuda())
    assert m.status(clear=True) & NAN
    sd = syn.make_state_dict(3, 3, True, 2)
    sd["conv0_4.bn2.bias"] = sd["conv0_4.bn2.bias"].copy()
    sd["conv0_4.bn2.bias"][3] = np.float32("inf")
    m.load_state_dict(sd, strict=True)
    assert m.status(clear=True) & NAN
    m.load_state_dict(syn.make_state_dict(3, 3, True, 2), strict=True)
    assert m.status() == 0
