#!/bin/bash
# usage: bash scripts/profile.sh <tag>    (run on the GPU box through gpurun)
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-per-layer > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-per-layer --no-forward > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-per-layer --no-forward > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-per-layer --no-forward > /dev/null 2> $OUT/pmc_mfma.err
cd $R && python3 scripts/summarize_profile.py $OUT > $OUT/summary.md 2> $OUT/summary.err; cat $OUT/summary.md | head -60
# keep the merge small: drop the raw per-dispatch traces, keep stats + summary
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
