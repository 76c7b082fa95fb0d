#!/bin/bash
# A/B on one box: the working tree against the last commit (a worktree under build/ab_old, built there), same bench
# arguments, alternating.   usage (from the repo root; the worktree is made HERE, before gpurun ships the tree):
#   git worktree add -f build/ab_old HEAD && (cd build/ab_old && python -c "from unet_amd import _lib; _lib.build()")
#   gpurun -- 'bash scripts/dev/ab_head.sh --batch 1'
cd "$(dirname "$0")/../.."
for round in 1 2; do
  for tree in . build/ab_old; do
    for prec in exact exact8; do
      (cd $tree && python bench.py "$@" --precision $prec --no-e2e-leg --no-fast-leg --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$tree', '$prec', ['%.4f' % t for t in d['ms_per_step_samples']])")
    done
  done
done
