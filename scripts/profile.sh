#!/bin/bash
# usage: bash scripts/profile.sh <tag> <commit> [config]   (run on the GPU box through gpurun; the snapshot has no .git,
# so the commit the binary was built from is passed in: bash scripts/profile.sh r03 $(git rev-parse --short HEAD))
TAG=${1:-r01}; COMMIT=${2:-unknown}; CFG=${3:-resnet50_tt}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-per-layer > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-per-layer --no-forward > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-per-layer --no-forward > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-per-layer --no-forward > /dev/null 2> $OUT/pmc_mfma.err
cd $R && python3 scripts/summarize_profile.py $OUT $COMMIT $CFG > $OUT/summary.md 2> $OUT/summary.err; cat $OUT/summary.md | head -60
# keep the merge small: drop the raw per-dispatch traces, keep stats + summary
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
