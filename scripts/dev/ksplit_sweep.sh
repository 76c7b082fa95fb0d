mkdir -p gpurun_out/r3g
for k in 1 4,2 8,2 8,4 16,2 16,4; do
  echo "== UNETPP_KSPLIT=$k"
  UNETPP_KSPLIT=$k python scripts/layer_profile.py exact 1 512 512 3 50 2>&1 | grep "conv2_0\|conv3_\|conv4_\|conv2_2\|wall" | awk '{print $1, $(NF-5), $(NF-4)}' | tr '\n' ';'
  echo
done
