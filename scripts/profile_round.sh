#!/bin/bash
# The five measurement passes behind profiles/<TAG>_* (run on the GPU box from the repo root; output under gpurun_out/<TAG>):
#   1. bench.py un-profiled (the JSON line)            2. rocprofv3 --kernel-trace --stats of the same command
#   3./4. --pmc FETCH_SIZE / --pmc WRITE_SIZE passes    5. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
# then the summaries (kernel stats, per-kernel / per-launch HBM traffic tagged with the build's source hash, MFMA busy).
# rocprofv3 gets the python interpreter itself after `--` (no env/bash hop: the profiler has initialised the GPU).
# usage: scripts/profile_round.sh TAG [WORKLOAD_KEY [bench.py arguments ...]]   e.g.  r03_v1_x8 nested-c3-512x512-b16-exact8 --precision exact8
set -e
TAG=${1:-r02}
WORKLOAD=${2:-nested-c3-512x512-b16-exact}
shift; shift || true
ARGS="$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" > $OUT/progress.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-fast-leg --no-e2e-leg $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done" >> $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-fast-leg --no-e2e-leg $ARGS > /dev/null 2> $OUT/fetch.err
echo "fetch done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-fast-leg --no-e2e-leg $ARGS > /dev/null 2> $OUT/write.err
echo "write done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-fast-leg --no-e2e-leg $ARGS > /dev/null 2> $OUT/mfma.err
echo "mfma done" >> $OUT/progress.txt
cd $ROOT
STATS=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
FETCH=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
WRITE=$(find $OUT/write -name "*counter_collection.csv" | head -1)
MFMA=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
python3 scripts/summarize_profile.py $TAG $STATS $FETCH $WRITE $WORKLOAD > $OUT/summary.txt
python3 scripts/pmc_layers.py $FETCH $WRITE > profiles/${TAG}_pmc_per_layer.txt
python3 scripts/pmc_mfma.py $MFMA $STATS > profiles/${TAG}_pmc_mfma_busy.txt
cp $OUT/bench.json profiles/${TAG}_bench.json
cp $OUT/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
# the un-profiled bench line once more, now that this build's traffic file exists (so that the line carries `traffic`)
python3 bench.py --steps 20 --warmup 5 $ARGS > profiles/${TAG}_bench.json 2> $OUT/bench2.err
mkdir -p $ROOT/gpurun_out/${TAG}_profiles && cp profiles/${TAG}_* $ROOT/gpurun_out/${TAG}_profiles/
echo "summaries done" >> $OUT/progress.txt
tail -3 profiles/${TAG}_pmc_per_layer.txt
