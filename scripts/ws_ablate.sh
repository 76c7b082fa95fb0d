#!/bin/bash
# Phase ablations of conv3x3_ws_kernel: builds the library with -DUNETPP_WS_DBG, runs the per-layer
# profile once per UNETPP_WS_DBG value given on the command line, then rebuilds the product library.
#   bits: 1 no interpolation, 2 no halo / low-res DMA, 4 no MFMAs, 8 no slab DMA, 16 no consumer work at all,
#         32 corner reads of the interpolation in the old (2-way conflicting) order, 64 interpolation without split + stores,
#         128 no DMA wait, 256 no epilogue, 512 interpolation without corner reads; bits 10-11 producer s_setprio,
#         12-13 consumer s_setprio; 16384 halo DMA with a pixel's units on neighbouring lanes; 262144 epilogue without its global stores; 32768 stamps inside the
#         interpolation (scripts/ws_stamps.sh).  Results are garbage -- and so are the operands of later layers: the
#         chip's clock depends on the data, compare a layer only with itself.
# usage (on the GPU box, from the repo root): [PREC=exact8] [BATCH=1] [LAYERS='conv1_3\|conv0_4'] scripts/ws_ablate.sh 0 1 4 256 ...
set -e
cd "$(dirname "$0")/.."
H=$(python -c 'from unet_amd import _lib; print(_lib.source_hash())')
FLAGS=$(python -c 'from unet_amd import _lib; print(" ".join(_lib.CXXFLAGS))')
(cd unet-_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DUNETPP_WS_DBG=1 -DUNETPP_SRC_HASH=\"$H\" -o ../libunetpp_hip.so unetpp_abi.hip)
for d in "$@"; do
  echo "UNETPP_WS_DBG=$d"
  UNETPP_ALLOW_DBG_LIB=1 UNETPP_WS_DBG=$d timeout -k 10 120 python scripts/layer_profile.py ${PREC:-exact} ${BATCH:-16} 2>&1 | grep "${LAYERS:-conv0_0\|conv0_4}" || true
done
UNETPP_FORCE_BUILD=1 python __graft_entry__.py > /dev/null
