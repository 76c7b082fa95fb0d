#!/bin/bash
# One wall-clock timeline of an un-profiled forward (measurement build): entry of the first workgroup and end of the last
# stamped wave of every wave-specialised launch on the chip's 100 MHz clock -- what lies inside the launches and between them.
# usage (GPU box, repo root): [PREC=exact8] [BATCH=1] scripts/dev/timeline.sh
set -e
cd "$(dirname "$0")/../.."
H=$(python -c 'from unet_amd import _lib; print(_lib.source_hash())')
FLAGS=$(python -c 'from unet_amd import _lib; print(" ".join(_lib.CXXFLAGS))')
(cd unet-_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DUNETPP_WS_DBG=1 -DUNETPP_SRC_HASH=\"$H\" -o ../libunetpp_hip.so unetpp_abi.hip)
UNETPP_ALLOW_DBG_LIB=1 UNETPP_WS_STAMPS=all timeout -k 10 120 python - <<PY 2>&1 | grep "timeline\|wall" || true
import sys, time, torch
sys.path.insert(0, ".")
from unet_amd import synthetic as syn
from unet_amd.nested_unet import NestedUNet
B = ${BATCH:-1}
m = NestedUNet(3, deep_supervision=True, precision="${PREC:-exact}", max_batch=B, max_hw=(512, 512)).to("cuda:0")
m.load_state_dict(syn.make_state_dict(3, 3, True, 2))
x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(B, 512, 512, "smooth", 1234))).cuda()
for _ in range(10):
    m.segment(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    m.segment(x)
torch.cuda.synchronize(); print(f"wall {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per forward")
PY
UNETPP_FORCE_BUILD=1 python __graft_entry__.py > /dev/null
