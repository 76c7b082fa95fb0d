#!/bin/bash
# Ablation bits applied to ONE launch at a time (its inputs stay genuine): for every named layer, its time without and with
# each bit set.   usage (GPU box, repo root): [PREC=exact8] [BATCH=16] scripts/dev/ablate_one.sh "262144 256" conv1_0.conv1 conv4_0.conv2 ...
set -e
cd "$(dirname "$0")/../.."
BITS=$1; shift
H=$(python -c 'from unet_amd import _lib; print(_lib.source_hash())')
FLAGS=$(python -c 'from unet_amd import _lib; print(" ".join(_lib.CXXFLAGS))')
(cd unet-_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DUNETPP_WS_DBG=1 -DUNETPP_SRC_HASH=\"$H\" -o ../libunetpp_hip.so unetpp_abi.hip)
for l in "$@"; do
  for d in 0 $BITS; do
    printf "%s dbg=%s: " $l $d
    UNETPP_ALLOW_DBG_LIB=1 UNETPP_WS_DBG=$d UNETPP_WS_DBG_ONLY=$l timeout -k 10 120 python scripts/layer_profile.py ${PREC:-exact} ${BATCH:-16} 2>&1 | grep "$l|\|$l+" | awk '{print $(NF-5), $(NF-4)}' | tr '\n' ' '
    echo
  done
done
UNETPP_FORCE_BUILD=1 python __graft_entry__.py > /dev/null
