#!/bin/bash
# as ab_head.sh, over the workloads of BASELINE.json: config 2, batch 1, config 4, config 5, SimpleUNet, and the fast mode
cd "$(dirname "$0")/../.."
one() {
  for tree in . build/ab_old; do
    (cd $tree && python bench.py "$@" --no-e2e-leg --no-fast-leg --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$tree', '$*', ['%.4f' % t for t in d['ms_per_step_samples']])")
  done
}
for round in 1 2; do
  one --precision exact; one --precision exact8; one --precision fast
  one --batch 1 --precision exact; one --batch 1 --precision exact8
  one --arch simple --precision exact; one --arch simple --precision exact8; one --arch simple --precision fast
done
one --classes 7 --height 448 --width 800 --batch 32 --precision exact; one --height 1024 --width 1024 --batch 8 --precision exact
