#!/bin/bash
# The other workloads of BASELINE.json / SURVEY 8(d), two passes each (un-profiled bench line, rocprofv3 --kernel-trace --stats of
# the same command): config 4 (7-class 448x800 batch 32), config 5 (3-class 1024x1024 batch 8), SimpleUNet (7-class 256x256
# batch 16), and the reference frame loop's batch of 1.  Output: profiles/<TAG>_<name>_bench.json, _kernel_stats.csv,
# _bench_under_rocprof.json.   usage (GPU box, repo root): scripts/profile_configs.sh TAG
set -e
TAG=${1:-r03}
ROOT=$(pwd)
run() {
  name=$1; shift
  OUT=$ROOT/gpurun_out/${TAG}_$name
  mkdir -p $OUT
  cd $ROOT
  python3 bench.py --steps 20 --warmup 5 "$@" > profiles/${TAG}_${name}_bench.json 2> $OUT/bench.err
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-fast-leg --no-e2e-leg "$@" > $ROOT/profiles/${TAG}_${name}_bench_under_rocprof.json 2> $OUT/stats.err
  cd $ROOT
  cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}_${name}_kernel_stats.csv
  echo "$name done" >> $ROOT/gpurun_out/${TAG}_configs_progress.txt
}
run c4 --classes 7 --height 448 --width 800 --batch 32 --cpu-frames 4
run c5 --height 1024 --width 1024 --batch 8 --cpu-frames 4
run simple --arch simple
run b1 --batch 1
mkdir -p $ROOT/gpurun_out/${TAG}_profiles && cp profiles/${TAG}_* $ROOT/gpurun_out/${TAG}_profiles/
