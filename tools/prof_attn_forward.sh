#!/bin/bash
# Dev recipe: per-kernel times of the attention U-Net forward (tools/one_attn_forward.py, 30 forwards at B = 64, N = 2048) under rocprofv3 --kernel-trace --stats.
# Run ON THE GPU BOX from the repo root:   bash tools/prof_attn_forward.sh   -> gpurun_out/attn_fwd/
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/attn_fwd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o r --output-format csv -- python3 $ROOT/tools/one_attn_forward.py 1 > $OUT/run.log 2>&1
cd $ROOT
cut -c1-160 $OUT/r_kernel_stats.csv | head -24
