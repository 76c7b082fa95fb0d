# batch-1 layer times over split-K plans: UNETPP_KSPLIT=<max>,<min chunks per share>,<gate: split when tiles * gate <= CUs>
for k in 1 16,4,4 16,4,2 16,2,2 16,2,1; do
  echo "== UNETPP_KSPLIT=$k"
  UNETPP_KSPLIT=$k python scripts/layer_profile.py ${PREC:-exact} 1 512 512 3 50 2>&1 | grep "conv1_0\|conv2_0\|conv3_\|conv4_\|conv2_2\|conv1_3.conv2\|wall" | awk '{print $1, $(NF-5), $(NF-4)}' | sed 's/|conv3x3_ws_kernel<2,//' | tr '\n' ';'
  echo
done
