#!/bin/bash
# Round-5 evidence, run ON THE GPU BOX from the repo root:   GIT_HEAD=<hash> bash tools/prof_round5.sh
# (1) bench.py (default cfg2 line) under rocprofv3 --kernel-trace --stats; (2) bench.py --config cfg4 the same way; (3) the PMC recipes whose json files bench.py
# reads (fabric traffic of the dominant GEMM: tools/pmc_gf3.sh; SQ counters of the three headline kernels: tools/pmc_sq.sh); (4) the default line un-profiled and
# the graph-first order.  Copy gpurun_out/r05/* of interest into profiles/ (r05_f ...), gf3_pmc_latest.json and sq_pmc_latest.json as they are.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/bench -o b --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/cfg4 -o c --output-format csv -- python3 $ROOT/bench.py --config cfg4 --no-cpu-baseline > $OUT/cfg4_under_rocprof.json 2> $OUT/cfg4.err
cd $ROOT
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_gf3.sh > $OUT/pmc_gf3.log 2>&1
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_sq.sh > $OUT/pmc_sq.log 2>&1
find $OUT -name "*_kernel_stats.csv" | head
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --graph --no-cpu-baseline --no-attention --no-other-configs > $OUT/bench_graph_first.json 2> $OUT/bench_graph_first.err
