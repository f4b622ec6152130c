#!/bin/bash
# usage (GPU box): bash scripts/chain_trace.sh <config> <layer-substring> <tag>
R=$GRAFT_REPO_ROOT
CFG=${1:-resnet50_tt}; LAYER=${2:-layer4.1.conv2}; TAG=${3:-chain}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/scripts/chain_trace.py run $CFG $LAYER > $OUT/run.log 2> $OUT/err
python3 $R/scripts/chain_trace.py post $OUT/t > $OUT/seq.txt 2>> $OUT/err
find $OUT -name "*kernel_trace.csv" -size +4M -delete
head -30 $OUT/seq.txt
