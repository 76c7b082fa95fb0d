#!/bin/bash
# In-kernel phase times of conv3x3_ws_kernel for the named layers (measurement build -DUNETPP_WS_DBG, cycle sums of one
# consumer and one producer wave per workgroup; see WS_STAMP in csrc/conv3x3_ws.h), then rebuilds the product library.
# usage (on the GPU box, from the repo root): [PREC=exact8] [BATCH=1] [UNETPP_WS_DBG=bits] scripts/ws_stamps.sh conv0_4.conv1 conv0_4.conv2 ...
set -e
cd "$(dirname "$0")/.."
H=$(python -c 'from unet_amd import _lib; print(_lib.source_hash())')
FLAGS=$(python -c 'from unet_amd import _lib; print(" ".join(_lib.CXXFLAGS))')
(cd unet-_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DUNETPP_WS_DBG=1 -DUNETPP_SRC_HASH=\"$H\" -o ../libunetpp_hip.so unetpp_abi.hip)
for l in "$@"; do
  UNETPP_ALLOW_DBG_LIB=1 UNETPP_WS_STAMPS=$l timeout -k 10 120 python scripts/layer_profile.py ${PREC:-exact} ${BATCH:-16} 2>&1 | grep -A12 "stamps\|^$l" | grep -v "^--" || true
done
UNETPP_FORCE_BUILD=1 python __graft_entry__.py > /dev/null
