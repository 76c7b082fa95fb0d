"""The drop-in boundary is a C ABI: include/unetpp.h must be valid C99 and usable from a host with no Python,
torch or C++ in it.  tests/c_abi/abi_host.c is such a host; it is compiled with gcc here (CPU) and run against
the oracle on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c_abi", "abi_host.c")


def compile_host(out_dir):
    from unet_amd import _lib
    lib = _lib.build()
    exe = os.path.join(str(out_dir), "abi_host")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-Werror", "-pedantic",
           "-I", os.path.join(ROOT, "include"), "-o", exe, SRC,
           "-L", os.path.dirname(lib), "-lunetpp_hip", "-ldl", "-Wl,-rpath," + os.path.dirname(lib)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_c99_and_c_host_links(tmp_path):
    exe = compile_host(tmp_path)
    # no GPU here: the program must stop at its usage check, not at the dynamic loader
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("C,B,H,W", [(3, 2, 64, 96), (7, 1, 48, 80)])
def test_c_host_matches_oracle(C, B, H, W, tmp_path, syn, oracle):
    from unet_amd import packing
    exe = compile_host(tmp_path)
    sd = syn.make_state_dict(C, 3, C == 3, 2)
    frames = syn.make_frames_u8(B, H, W, "smooth", 77)
    blob = packing.build_blob(sd, C)
    blob.tofile(tmp_path / "blob.bin")
    frames.tofile(tmp_path / "frames.bin")
    r = subprocess.run([exe, str(tmp_path / "blob.bin"), str(tmp_path / "frames.bin"), str(tmp_path / "out.bin"),
                        str(C), str(B), str(H), str(W), "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "gfx950" in r.stdout
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
    px = B * H * W
    logits = raw[:px * C * 4].view(np.float32).reshape(B, C, H, W)
    mask, cable, tape = (raw[px * C * 4 + i * px: px * C * 4 + (i + 1) * px].reshape(B, H, W) for i in range(3))
    ref = oracle.torch_forward(sd, syn.frames_to_chw_f32(frames))
    ref_mask, ref_cable, ref_tape = oracle.masks_from_logits(ref)
    err = float(np.abs(logits - ref).max())
    assert err < 2e-5
    flips = mask != ref_mask
    assert not (flips & (oracle.top2_margin(ref) > 2 * err + 1e-7)).any()
    assert np.array_equal(cable, (mask == 1).astype(np.uint8)) and np.array_equal(tape, (mask == 2).astype(np.uint8))
    assert np.array_equal(cable[~flips], ref_cable[~flips]) and np.array_equal(tape[~flips], ref_tape[~flips])
