#!/bin/bash
# Dev recipe: per-layer table of VAE3DLarge decode + encode at B = 32 (tools/one_vae_decode.py under rocprofv3 --kernel-trace, table by tools/vae_layers.py).
# Run ON THE GPU BOX from the repo root:   bash tools/prof_vae_layers.sh   -> gpurun_out/vae_layers/table.txt
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/vae_layers
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o r --output-format rocpd -- python3 $ROOT/tools/one_vae_decode.py > $OUT/run.log 2>&1
cd $ROOT
DB=$(find $OUT -name "*.db" | head -1)
python3 tools/vae_layers.py $DB > $OUT/table.txt
cat $OUT/table.txt
