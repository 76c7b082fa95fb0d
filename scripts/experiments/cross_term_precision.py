"""Numerical experiment (CPU, torch): how accurate is the network when the two small cross terms of the split
product  x*w = xh*wh + xl*wh + xh*wl (+ xl*wl, dropped)  are computed from 8-bit floating-point operands?

gfx950 has v_mfma_f32_32x32x64_f8f6f4 at twice (fp8) to four times (fp6/fp4) the fp16 MFMA rate; the exact mode
spends two of its three MFMAs on terms that are 2^-11 of the result.  This script emulates, layer by layer,
    y = conv(xh, wh) + conv(q(xl), q(w)) + conv(q(x), q(wl))
with q = round to e4m3 (power-of-two tensor scale) and compares the logits with the fp32 reference path.
Nothing here touches the GPU or the product code.   usage: python scripts/experiments/cross_term_precision.py [size]
"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from unet_amd import synthetic, packing            # noqa: E402

torch.set_num_threads(8)


def split16(t):
    h = t.to(torch.float16).to(torch.float64)
    l = (t - h).to(torch.float16).to(torch.float64)
    return h, l


def q_e4m3(t, mant_bits=3):
    """round to an 8-bit float with `mant_bits` explicit mantissa bits after a power-of-two scale that puts the
    largest magnitude just under 256 (e4m3fn max 448)"""
    m = float(t.abs().max())
    if m == 0:
        return t
    s = 2.0 ** np.floor(np.log2(256.0 / m))
    if mant_bits == 3:
        return (t * s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) / s
    if mant_bits == 2:
        return (t * s / 4).to(torch.float32).to(torch.float8_e5m2).to(torch.float64) * 4 / s
    raise ValueError


def conv(x, w, mode):
    """x float64 [B,C,H,W] (already hi+lo representable), w float64 [Co,Ci,3,3]"""
    c = lambda a, b: F.conv2d(a, b, padding=w.shape[-1] // 2)
    if mode == "ref":
        return c(x, w)
    xh, xl = split16(x)
    wh, wl = split16(w)
    if mode == "exact3":
        return c(xh, wh) + c(xl, wh) + c(xh, wl)
    if mode == "fast":
        return c(xh, wh)
    if mode == "w16":      # weights rounded to fp16, activations hi+lo
        return c(xh, wh) + c(xl, wh)
    if mode == "x16":      # activations rounded to fp16, weights hi+lo
        return c(xh, wh) + c(xh, wl)
    if mode == "fp8":
        return c(xh, wh) + c(q_e4m3(xl), q_e4m3(w)) + c(q_e4m3(x), q_e4m3(wl))
    if mode.startswith("hw"):
        # the scheme as it would run: per-output-channel power-of-two weight scale 2^k (max |w 2^k| in [2^13, 2^14)),
        # xl8 = e4m3(xl 2^A), w8 = e4m3(w 2^(k-A)), x8 = e4m3(x), wl8 = e4m3(residual of the scaled weight); saturating
        A = int(mode[2:])
        e4 = lambda t: t.clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        mx = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
        k = 14 - torch.floor(torch.log2(mx)) - 1
        ws = w * 2.0 ** k
        wsh = ws.to(torch.float16).to(torch.float64)
        wsl = ws - wsh
        acc = c(xh, wsh) + c(e4(xl * 2.0 ** A), e4(ws * 2.0 ** -A)) + c(e4(x), e4(wsl))
        return acc * (2.0 ** -k).reshape(1, -1, 1, 1)
    if mode in ("mix", "mixw5", "mix6"):
        # the EXACT8 scheme as built (csrc/conv3x3_ws.h, P8): activations' 8-bit planes are e5m2 (fp16's exponent range:
        # no scale, no calibration), weights' are e4m3 with a per-output-channel power-of-two block scale
        e5 = lambda t: t.clamp(-57344, 57344).to(torch.float32).to(torch.float8_e5m2).to(torch.float64)
        e4 = lambda t: t.clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        mx = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
        k = 14 - torch.floor(torch.log2(mx)) - 1
        ws = w * 2.0 ** k
        wsh = ws.to(torch.float16).to(torch.float64)
        wsl = (ws - wsh).to(torch.float16).to(torch.float64)
        def qw(t):
            if mode == "mixw5":
                return e5(t)
            m = t.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
            s = 2.0 ** (8 - torch.floor(torch.log2(m)) - 1)      # max |t s| in [128, 256)
            return e4(t * s) / s
        acc = c(xh, wsh) + c(e5(xl), qw(wsh)) + c(e5(x), qw(wsl))
        return acc * (2.0 ** -k).reshape(1, -1, 1, 1)
    if mode == "x8":
        # precision 'exact8' as built (csrc/conv3x3_mfma.h split_pack4_x8, aux_kernels.h weight_pack_x8_kernel):
        #   lo8 = e5m2(2^8 xl), x8 = e5m2(2^-3 x), wh8 = e4m3(2^-6 ws), wl8 = e4m3(2^5 (ws - fp16(ws))), block scales 2^6 * 2^-8
        e5 = lambda t: t.to(torch.float32).to(torch.float8_e5m2).to(torch.float64)
        e4 = lambda t: t.to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        mx = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
        k = 14 - torch.floor(torch.log2(mx)) - 1
        ws = w * 2.0 ** k
        wsh = ws.to(torch.float16).to(torch.float64)
        acc = c(xh, wsh) + (c(e5(xl * 256.0), e4(ws / 64.0)) + c(e5(x / 8.0), e4((ws - wsh) * 32.0))) * 0.25
        return acc * (2.0 ** -k).reshape(1, -1, 1, 1)
    if mode.startswith("y8"):
        # candidates for a more accurate exact8: activation planes in e4m3 (one more mantissa bit than e5m2, eight binades less
        # range) with fixed scales 2^LS (residual) and 2^XS (value), saturating.  mode = y8_<LS>_<XS>[_w5]
        parts = mode.split("_")
        LS, XS = int(parts[1]), int(parts[2])
        e4 = lambda t: t.clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        mx = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
        k = 14 - torch.floor(torch.log2(mx)) - 1
        ws = w * 2.0 ** k
        wsh = ws.to(torch.float16).to(torch.float64)
        acc = c(xh, wsh) + c(e4(xl * 2.0 ** LS), e4(ws / 64.0)) * 2.0 ** (6 - LS) + c(e4(x * 2.0 ** XS), e4((ws - wsh) * 32.0)) * 2.0 ** (-5 - XS)
        return acc * (2.0 ** -k).reshape(1, -1, 1, 1)
    if mode == "bf8":
        return c(xh, wh) + c(q_e4m3(xl, 2), q_e4m3(w, 2)) + c(q_e4m3(x, 2), q_e4m3(wl, 2))
    raise ValueError(mode)


def forward(layers, head, x, mode):
    def block(t, name):
        for j in (1, 2):
            w, b = layers[f"{name}.conv{j}"]
            # per-output-channel power-of-two scaling as the engine does (keeps fp16 weights in range)
            t = conv(t, w, mode) + b[None, :, None, None]
            t = torch.relu(t)
            if mode.startswith("y8"):
                LS = int(mode.split("_")[1])
                h = t.to(torch.float16).to(torch.float64)
                t = h + ((t - h) * 2.0 ** LS).clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) / 2.0 ** LS
            elif mode == "x8":       # stored as fp16 hi + e5m2(2^8 lo): what the next layer (conv, upsample, pool) reads back
                h = t.to(torch.float16).to(torch.float64)
                t = h + ((t - h) * 256.0).to(torch.float32).to(torch.float8_e5m2).to(torch.float64) / 256.0
            elif mode != "ref":    # activations are stored as fp16 hi + lo
                h, l = split16(t)
                t = h + l
        return t
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    pool = lambda t: F.max_pool2d(t, 2)
    x00 = block(x, "conv0_0"); x10 = block(pool(x00), "conv1_0"); x20 = block(pool(x10), "conv2_0")
    x30 = block(pool(x20), "conv3_0"); x40 = block(pool(x30), "conv4_0")
    x31 = block(torch.cat([x30, up(x40)], 1), "conv3_1")
    x22 = block(torch.cat([x20, up(x31)], 1), "conv2_2")
    x13 = block(torch.cat([x10, up(x22)], 1), "conv1_3")
    x04 = block(torch.cat([x00, up(x13)], 1), "conv0_4")
    wf, bf = head
    return F.conv2d(x04, wf) + bf[None, :, None, None]


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for kind in ("trained_like", "random"):
        sd = synthetic.make_trained_like_state_dict(3) if kind == "trained_like" else synthetic.make_state_dict(3)
        layers = {}
        for name, w, b in packing.folded_layers(sd):
            layers[name] = (torch.from_numpy(np.asarray(w, np.float64)), torch.from_numpy(np.asarray(b, np.float64)))
        head = (torch.from_numpy(np.asarray(sd["final.weight"], np.float64)), torch.from_numpy(np.asarray(sd["final.bias"], np.float64)))
        frames = synthetic.frames_to_chw_f32(synthetic.make_frames_u8(2, size, size))
        x = torch.from_numpy(frames.astype(np.float64))
        with torch.no_grad():
            ref = forward(layers, head, x, "ref")
            srt = torch.sort(ref, dim=1).values
            margin = (srt[:, -1] - srt[:, -2])
            print(f"== {kind} weights, {size}x{size}, 2 frames; |logit| max {float(ref.abs().max()):.2f}")
            for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("exact3", "fp8", "bf8", "x16", "w16", "fast")):
                out = forward(layers, head, x, mode)
                err = (out - ref).abs()
                flips = (out.argmax(1) != ref.argmax(1))
                nf = int(flips.sum())
                worst_margin = float(margin[flips].max()) if nf else 0.0
                print(f"  {mode:7s} max |dlogit| {float(err.max()):.3e}   mean {float(err.mean()):.3e}   mask flips {nf} of {flips.numel()}"
                      f"   largest reference margin at a flip {worst_margin:.2e}")


if __name__ == "__main__":
    main()
