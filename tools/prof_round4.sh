#!/bin/bash
# Round-4 evidence, run ON THE GPU BOX from the repo root: bash tools/prof_round3.sh
# (1) bench.py (default cfg2 line) under rocprofv3 --kernel-trace --stats; (2) bench.py --config cfg4 the same way (the persistent
# latent kernel's duration = the 1000-step loop); (3) the dominant GEMM's fabric traffic (two --pmc passes, tools/pmc_gf3.sh).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/bench -o b --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/cfg4 -o c --output-format csv -- python3 $ROOT/bench.py --config cfg4 --no-cpu-baseline > $OUT/cfg4_under_rocprof.json 2> $OUT/cfg4.err
cd $ROOT
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_gf3.sh > $OUT/pmc_gf3.log 2>&1
find $OUT -name "*_kernel_stats.csv" | head
# (4) the default line un-profiled, and the two launch modes in the other order (--graph: the replay leg runs first, the eager leg second)
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --graph --no-cpu-baseline --no-attention --no-other-configs > $OUT/bench_graph_first.json 2> $OUT/bench_graph_first.err
