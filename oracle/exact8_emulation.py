"""TEST INFRASTRUCTURE — CPU emulation (torch, float64) of the ARITHMETIC of the engine's precision 'exact8'
(include/unetpp.h UNETPP_PREC_EXACT8; unet-_amd/csrc/conv3x3_ws.h, conv3x3_mfma.h split_pack4_x8, aux_kernels.h
weight_pack_x8_kernel, tapmm_ws.h), layer by layer, following the engine's own execution plan for NestedUNet
(reference graph: src/models/unetpp.py:93-135).

This is NOT the parity oracle (that is unetpp_oracle.py: the fp32 reference restated; exact8 is judged against it at the
north-star bar of 1e-3).  It answers a different question: does the GPU compute what DESIGN.md §3 says it computes?  A wrong
block scale, a swapped byte order or a dropped cross term would still pass the 1e-3 gate on most inputs (the terms are 2^-11
of the result).

How it is used (tests/test_gpu_exact8.py): LAYER BY LAYER on the GPU's own stored inputs (`act_from_planes`), not end to
end.  The residual plane is a discontinuous function of the value: an fp32 summation-order difference of 1e-7 flips the e5m2
rounding of lo8 in ~0.2-3 % of the elements by one ulp (2^-14 .. 2^-12 of the value), and those flips spread -- after a few
layers two correct implementations of this arithmetic are as far from each other (1e-4 in the logits) as each is from the fp32
reference (measured: scripts/dev/x8_vs_emulation.py; it is also why two launch plans of the engine itself differ by 2e-4).
Within ONE layer the agreement is 50x closer than the layer's distance to the reference, and a dropped or mis-scaled term is
not.

Only tests/ may import this file.  The product path never does.

  stored activation   (h, l8, x8):  h = fp16(v);  l8 = e5m2(2^8 (v - h));  x8 = e5m2(2^-3 v);  read back as h + 2^-8 l8
  weights             ws = w 2^k (max |ws| of an output channel in [2^13, 2^14));  wh = fp16(ws);
                      wh8 = e4m3(2^-6 ws);  wl8 = e4m3(2^5 (ws - wh))
  product             acc += wh h + 2^6 2^-8 (wh8 l8 + wl8 x8);   v = relu(2^-k acc + bias)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
NB = (32, 64, 128, 256, 512)          # unetpp.py:49


def _t(a):
    return torch.from_numpy(np.asarray(a, np.float64))


def _f16(t):
    return t.to(torch.float16).to(torch.float64)


def _e5(t):
    return t.to(torch.float32).to(torch.float8_e5m2).to(torch.float64)


def _e4(t):
    return t.to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)


def _fold(sd, name):
    """BN folded into the conv in float64, then float32 (unet-_amd/packing.py fold_conv_bn: the canonical blob is fp32)."""
    w = _t(sd[f"{name}.weight"]); b = _t(sd[f"{name}.bias"])
    bn = name[:-5] + "bn" + name[-1]                          # 'conv0_0.conv1' -> 'conv0_0.bn1'
    s = _t(sd[f"{bn}.weight"]) / torch.sqrt(_t(sd[f"{bn}.running_var"]) + BN_EPS)
    wf = (w * s[:, None, None, None]).to(torch.float32).to(torch.float64)
    bf = ((b - _t(sd[f"{bn}.running_mean"])) * s + _t(sd[f"{bn}.bias"])).to(torch.float32).to(torch.float64)
    return wf, bf


def _scaled(w):
    """per-output-channel power of two (aux_kernels.h weight_scale_kernel): ws = w 2^k, max |ws| in [2^13, 2^14)"""
    m = w.abs().amax(dim=(1, 2, 3))
    k = torch.where(m > 0, 14 - (torch.floor(torch.log2(m.clamp_min(1e-300))) + 1), torch.zeros_like(m))
    return w * (2.0 ** k)[:, None, None, None], k


class Act:
    """an activation tensor as the engine stores it"""

    def __init__(self, v):
        v = v.to(torch.float32).to(torch.float64)            # the epilogue's fp32 value
        self.h = _f16(v)
        self.l8 = _e5((v - self.h) * 256.0)
        self.x8 = _e5(v / 8.0)

    @property
    def value(self):                                          # what a reader reconstructs: hi + 2^-8 lo8
        return self.h + self.l8 / 256.0


def act_from_planes(h, l8, x8) -> Act:
    """a tensor the engine stored, from its three planes as unetpp_debug_read returns them ("name#hi", "name#lo", "name#x8").
    (The reconstructed value h + 2^-8 l8 alone does not determine them: where e5m2 rounds the residual up to half an fp16 ulp,
    (h, +ulp/2) and (h + ulp, -ulp/2) read back the same -- 3 % of the elements -- and x8 is a rounding of the unsplit value.)"""
    a = Act.__new__(Act)
    a.h, a.l8, a.x8 = _t(h), _t(l8), _t(x8)
    return a


def conv_layer(inp: Act, sd, name):
    """fp32 output values of one conv3x3 + BN + ReLU in exact8 arithmetic (before they are split into planes)"""
    w, b = _fold(sd, name)
    return _conv(inp, w, b)


def decoder_conv1_layer(skip: Act, low: Act, sd, name, lowres_gemm: bool):
    w, b = _fold(sd, name)
    return _decoder_conv1(skip, low, w, b, lowres_gemm)


def first_block(x, sd):
    w1, b1 = _fold(sd, "conv0_0.conv1"); w2, b2 = _fold(sd, "conv0_0.conv2")
    return _conv(Act(_first_conv(_t(x).to(torch.float32).to(torch.float64), w1, b1)), w2, b2)


def stored(v):
    """what the engine reads back after storing fp32 values v"""
    return Act(v).value.numpy().astype(np.float32)


def pooled(v):
    return F.max_pool2d(v, 2)


def _cat(acts):
    a = Act.__new__(Act)
    a.h = torch.cat([t.h for t in acts], 1); a.l8 = torch.cat([t.l8 for t in acts], 1); a.x8 = torch.cat([t.x8 for t in acts], 1)
    return a


def _x8_product(act: Act, ws, pad):
    """sum over taps and channels of the exact8 product, in the accumulator domain (weights carry 2^k)"""
    wh = _f16(ws)
    c = lambda a, b: F.conv2d(a, b, padding=pad)
    return c(act.h, wh) + 0.25 * (c(act.l8, _e4(ws / 64.0)) + c(act.x8, _e4((ws - wh) * 32.0)))


def _finish(acc, k, b):
    return torch.relu(acc * (2.0 ** -k)[None, :, None, None] + b[None, :, None, None]).to(torch.float32).to(torch.float64)


def _conv(act: Act, w, b):
    ws, k = _scaled(w)
    return _finish(_x8_product(act, ws, 1), k, b)


def _up(v):
    return F.interpolate(v, scale_factor=2, mode="bilinear", align_corners=True)


def _first_conv(x, w, b):
    """conv0_0.conv1 in the loaders of the fused first block: three fp16 MFMAs on the raw input, both operands split into fp16
    hi + fp16 lo (conv3x3_ws.h C0F, aux_kernels.h conv0_pack_kernel); the lo * lo term is dropped"""
    ws, k = _scaled(w)
    xh = _f16(x); xl = _f16(x - xh)
    wh = _f16(ws); wl = _f16(ws - wh)
    c = lambda a, bb: F.conv2d(a, bb, padding=1)
    return _finish(c(xh, wh) + c(xl, wh) + c(xh, wl), k, b)


def _decoder_conv1(skip: Act, low: Act, w, b, lowres_gemm: bool):
    """conv3x3(cat([skip, up(low)])) as the engine runs it: levels 0-1 interpolate the stored low-res values in the loader
    and split the result into planes (UPF); levels 2-3 multiply the up channels at LOW resolution tap by tap, interpolate the
    fp32 products and add the taps that stay inside the image (tapmm_ws.h), then the skip channels' conv on top"""
    ws, k = _scaled(w)
    cs = skip.h.shape[1]
    if not lowres_gemm:
        return _finish(_x8_product(_cat([skip, Act(_up(low.value))]), ws, 1), k, b)
    acc = _x8_product(skip, ws[:, :cs], 1)
    H, W = skip.h.shape[2:]
    for dy in range(3):
        for dx in range(3):
            y = _x8_product(low, ws[:, cs:, dy:dy + 1, dx:dx + 1], 0)           # 1x1 GEMM of one tap at low resolution
            u = F.pad(_up(y), (1, 1, 1, 1))                                     # zero outside the high-res image
            acc = acc + u[:, :, dy:dy + H, dx:dx + W]
    return _finish(acc, k, b)


def exact8_forward(sd: dict, x: np.ndarray, tapmm_levels=(2, 3), return_nodes: bool = False):
    """logits [B, C, H, W] float32 (and, optionally, every node as the engine would read it back) for float32 input x [B,3,H,W]"""
    x = _t(x).to(torch.float32).to(torch.float64)
    nodes = {}
    with torch.no_grad():
        def block(inp_act, name, first=False):
            w1, b1 = _fold(sd, f"{name}.conv1"); w2, b2 = _fold(sd, f"{name}.conv2")
            v1 = _first_conv(inp_act, w1, b1) if first else _conv(inp_act, w1, b1)
            return _conv(Act(v1), w2, b2)                      # fp32 value of the block's output (before it is split)
        pool = lambda v: F.max_pool2d(v, 2)                   # on the epilogue's fp32 values (conv3x3_ws.h ws_epilogue, POOL)
        v = {}
        v["x0_0"] = block(x, "conv0_0", first=True)
        for l in range(1, 5):
            v[f"x{l}_0"] = block(Act(pool(v[f"x{l - 1}_0"])), f"conv{l}_0")
        low = Act(v["x4_0"])
        for l in (3, 2, 1, 0):
            name = f"conv{l}_{4 - l}"
            w1, b1 = _fold(sd, f"{name}.conv1"); w2, b2 = _fold(sd, f"{name}.conv2")
            v1 = _decoder_conv1(Act(v[f"x{l}_0"]), low, w1, b1, l in tapmm_levels)
            v[f"x{l}_{4 - l}"] = _conv(Act(v1), w2, b2)
            low = Act(v[f"x{l}_{4 - l}"])
        # the 1x1 head runs in fp32 on conv0_4.conv2's fp32 registers (nothing is split in between)
        wf = _t(sd["final.weight"]).to(torch.float32).to(torch.float64); bf = _t(sd["final.bias"]).to(torch.float32).to(torch.float64)
        logits = F.conv2d(v["x0_4"], wf) + bf[None, :, None, None]
        if return_nodes:
            nodes = {kk: Act(t).value.numpy().astype(np.float32) for kk, t in v.items() if kk != "x0_4"}
            nodes["x0_4"] = v["x0_4"].numpy().astype(np.float32)
    out = logits.numpy().astype(np.float32)
    return (out, nodes) if return_nodes else out
