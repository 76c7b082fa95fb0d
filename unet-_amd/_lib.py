"""ctypes binding of the C ABI declared in include/unetpp.h, plus the in-tree build of the HIP library.

The product path has no CPU fallback: if libunetpp_hip.so is missing or cannot be loaded, or no HIP
device is present, every entry point raises."""
from __future__ import annotations

import ctypes
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunetpp_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["unetpp_abi.hip"]
HEADERS = ["conv3x3_mfma.h", "conv3x3_ws.h", "tapmm_ws.h", "convt2x2_mfma.h", "aux_kernels.h", os.path.join("..", "..", "include", "unetpp.h")]

# every symbol include/unetpp.h declares
ABI_SYMBOLS = [
    "unetpp_create", "unetpp_destroy", "unetpp_last_error", "unetpp_version", "unetpp_weights_blob_bytes",
    "unetpp_weights_blob_bytes_arch",
    "unetpp_load_weights", "unetpp_load_weights_device", "unetpp_forward", "unetpp_forward_ex", "unetpp_mask_stats", "unetpp_workspace_bytes",
    "unetpp_resize_linear_u8", "unetpp_resize_nearest_roi_u8", "unetpp_status",
    "unetpp_profile_enable", "unetpp_profile_count", "unetpp_profile_read", "unetpp_profile_name",
    "unetpp_profile_work", "unetpp_debug_read", "unetpp_debug_keep_intermediates",
]

STATUS_OVERFLOW, STATUS_NAN = 1, 2
PREC_EXACT, PREC_FAST, PREC_EXACT8 = 0, 1, 2
PRECISIONS = {"exact": PREC_EXACT, "fast": PREC_FAST, "exact8": PREC_EXACT8}
ARCH_NESTED, ARCH_SIMPLE = 0, 1
IN_F32_NCHW, IN_U8_NHWC_BGR = 0, 1


class Config(ctypes.Structure):
    _fields_ = [("num_classes", ctypes.c_int), ("in_channels", ctypes.c_int), ("max_batch", ctypes.c_int),
                ("max_h", ctypes.c_int), ("max_w", ctypes.c_int), ("precision", ctypes.c_int),
                ("device", ctypes.c_int), ("micro_batch", ctypes.c_int), ("streams", ctypes.c_int), ("arch", ctypes.c_int)]


RULES = {"argmax": 0, "thresholded_argmax": 1, "strict_bg_check": 2, "exclusive": 3}


class Outputs(ctypes.Structure):
    _fields_ = [("dev_logits", ctypes.c_void_p), ("dev_probs", ctypes.c_void_p), ("dev_mask", ctypes.c_void_p),
                ("dev_cable", ctypes.c_void_p), ("dev_tape", ctypes.c_void_p), ("rule", ctypes.c_int),
                ("t_cable", ctypes.c_float), ("t_tape", ctypes.c_float), ("bg_margin", ctypes.c_float),
                ("ct_margin", ctypes.c_float)]


# -fno-slp-vectorize: the SLP vectoriser turns pairs of float operations into v_pk_mul_f32 / v_pk_fma_f32, and a packed
# fp32 instruction issued beside another wave's MFMA stream on the same SIMD takes 26-58 cycles instead of 5-7
# (scripts/microbench/valu_beside_mfma.hip) -- the loader waves of the wave-specialised kernels run exactly there.
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"]


def source_hash() -> str:
    """Digest of every file the library is compiled from and of the compiler flags; baked into the binary
    (unetpp_version() ends in 'src:<hash>') so that a .so built from other sources is recognised wherever it travels."""
    h = hashlib.sha256()
    h.update(" ".join(CXXFLAGS).encode() + b"\0")
    for rel in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def built_hash(path: str = LIB_PATH):
    """The 'src:' tag of a built library, read from the file (no dlopen), or None."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(b"(gfx950) src:")
    if i < 0:
        return None
    tag = blob[i + 13:i + 13 + 16]
    if blob[i + 29:i + 36] == b" +wsdbg" and not os.environ.get("UNETPP_ALLOW_DBG_LIB"):
        return "wsdbg-build"                 # a measurement build never counts as current
    return tag.decode("ascii", "replace")


def _stale() -> bool:
    return built_hash() != source_hash()


def _hipcc():
    cand = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return cand if os.path.exists(cand) else shutil.which("hipcc")


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> unet-_amd/libunetpp_hip.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    if not hipcc:
        raise RuntimeError("hipcc not found: cannot build libunetpp_hip.so")
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"      # several ranks may build at once: write aside, then rename atomically
    cmd = [hipcc] + CXXFLAGS + ["-shared", "-fPIC", f'-DUNETPP_SRC_HASH="{source_hash()}"',
           "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd).replace(tmp, LIB_PATH), flush=True)
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


_lib = None


def share_torch_hip_runtime() -> None:
    """Import torch (when it is installed) BEFORE the library is dlopen'ed.  The torch wheel ships its own
    libamdhip64 / libhsa-runtime64; loaded first, they satisfy this library's DT_NEEDED entries and the process has
    one HIP runtime.  The other way round the library pulls in /opt/rocm's copies, torch adds its own, and whichever
    runtime initialises second finds no device ("no HIP device available" from unetpp_create after a torch CUDA call)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    share_torch_hip_runtime()
    if _stale():
        # missing, or built from other sources than the tree holds (the .so is git-ignored but travels to the GPU
        # box): rebuild when a compiler is at hand, otherwise refuse -- never run kernels that do not match the tree
        what = "is missing" if not os.path.exists(LIB_PATH) else f"was built from other sources (src:{built_hash()}, tree {source_hash()})"
        if not build_if_missing or not _hipcc():
            raise RuntimeError(f"{LIB_PATH} {what}: build it with __graft_entry__.build(); there is no CPU fallback")
        build()
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.unetpp_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]; lib.unetpp_create.restype = ci
    lib.unetpp_destroy.argtypes = [vp]; lib.unetpp_destroy.restype = None
    lib.unetpp_last_error.argtypes = [vp]; lib.unetpp_last_error.restype = ctypes.c_char_p
    lib.unetpp_version.argtypes = []; lib.unetpp_version.restype = ctypes.c_char_p
    lib.unetpp_weights_blob_bytes.argtypes = [ci, ci]; lib.unetpp_weights_blob_bytes.restype = cs
    lib.unetpp_weights_blob_bytes_arch.argtypes = [ci, ci, ci]; lib.unetpp_weights_blob_bytes_arch.restype = cs
    lib.unetpp_load_weights.argtypes = [vp, vp, cs]; lib.unetpp_load_weights.restype = ci
    lib.unetpp_load_weights_device.argtypes = [vp, vp, cs, vp]; lib.unetpp_load_weights_device.restype = ci
    lib.unetpp_forward.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]; lib.unetpp_forward.restype = ci
    lib.unetpp_forward_ex.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.POINTER(Outputs), vp]; lib.unetpp_forward_ex.restype = ci
    lib.unetpp_mask_stats.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp]; lib.unetpp_mask_stats.restype = ci
    lib.unetpp_resize_linear_u8.argtypes = [vp, vp, ci, ci, ci, ci, vp, ci, ci, vp]; lib.unetpp_resize_linear_u8.restype = ci
    lib.unetpp_resize_nearest_roi_u8.argtypes = [vp, vp, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.unetpp_resize_nearest_roi_u8.restype = ci
    lib.unetpp_workspace_bytes.argtypes = [vp]; lib.unetpp_workspace_bytes.restype = cs
    lib.unetpp_status.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ci]; lib.unetpp_status.restype = ci
    lib.unetpp_profile_enable.argtypes = [vp, ci]; lib.unetpp_profile_enable.restype = ci
    lib.unetpp_profile_count.argtypes = [vp]; lib.unetpp_profile_count.restype = ci
    lib.unetpp_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ci]; lib.unetpp_profile_read.restype = ci
    lib.unetpp_profile_name.argtypes = [vp, ci]; lib.unetpp_profile_name.restype = ctypes.c_char_p
    lib.unetpp_profile_work.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.unetpp_profile_work.restype = ci
    lib.unetpp_debug_read.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), cs]
    lib.unetpp_debug_read.restype = ctypes.c_longlong
    lib.unetpp_debug_keep_intermediates.argtypes = [vp, ci]; lib.unetpp_debug_keep_intermediates.restype = ci
    ver = lib.unetpp_version().decode()
    if ver.endswith(" +wsdbg") and os.environ.get("UNETPP_ALLOW_DBG_LIB"):
        ver = ver[:-len(" +wsdbg")]            # measurement build with phase ablations (scripts/ws_ablate.sh)
    if not ver.endswith("src:" + source_hash()):
        raise RuntimeError(f"{LIB_PATH} reports {lib.unetpp_version().decode()!r}, tree is src:{source_hash()}")
    _lib = lib
    return lib
