#!/bin/bash
# A/B of one compile-time switch on one box: layer tables (batch 1 and 16) of the product build, then of a build with the
# given -D flag(s), then the product build is restored.   usage: scripts/dev/ab_flag.sh -DUNETPP_WT_STORES [exact|exact8]
set -e
cd "$(dirname "$0")/../.."
PREC=${2:-exact}
H=$(python -c 'from unet_amd import _lib; print(_lib.source_hash())')
FLAGS=$(python -c 'from unet_amd import _lib; print(" ".join(_lib.CXXFLAGS))')
run() {
  for b in 1 16; do
    echo "== $1 batch $b"
    python scripts/layer_profile.py $PREC $b 512 512 3 50 2>&1 | grep " us \|^wall\|^sum" | awk '{ if ($1 == "wall" || $1 == "sum") print; else print $1, $(NF-5), $(NF-4) }' | sed 's/|conv3x3_ws_kernel<2,/ /'
  done
}
run base
(cd unet-_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -shared -fPIC $1 -DUNETPP_SRC_HASH=\"$H\" -o ../libunetpp_hip.so unetpp_abi.hip)
run "$1"
UNETPP_FORCE_BUILD=1 python __graft_entry__.py > /dev/null
