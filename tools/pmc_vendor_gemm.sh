#!/bin/bash
# Dev recipe (yardstick only): fabric traffic of the vendor library's GEMM kernel on the global_feat.3 shape (M = 131072, K = 2048, N = 4096, fp16),
# the same two rocprofv3 --pmc passes and corrections as tools/pmc_gf3.sh, over tools/one_vendor_gemm.py.  Run ON THE GPU BOX from the repo root.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_vendor
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python3 $ROOT/tools/one_vendor_gemm.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/write -o w --output-format csv -- python3 $ROOT/tools/one_vendor_gemm.py > /dev/null 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
def rows(root, counter):
    f = glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True)[0]
    return [(r["Kernel_Name"][:60], int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f)) if "Cijk" in r["Kernel_Name"] and r["Counter_Name"] == counter]
f = rows("$OUT/fetch", "FETCH_SIZE"); w = rows("$OUT/write", "WRITE_SIZE"); h = rows("$OUT/write", "TCC_HIT_sum"); m = rows("$OUT/write", "TCC_MISS_sum")
# one_vendor_gemm.py launches 5 x (2048, 4096), 5 x (1024, 2048), 5 x (1024, 1024), 5 x (512, 512) in that order
names = ["K=2048 N=4096", "K=1024 N=2048", "K=1024 N=1024", "K= 512 N= 512"]
for i, nm in enumerate(names):
    sl = slice(5 * i + 2, 5 * i + 5)
    fk = sum(v for _, _, v in f[sl]) / 3; wk = sum(v for _, _, v in w[sl]) / 3
    hh = sum(v for _, _, v in h[sl]) / 3; mm = sum(v for _, _, v in m[sl]) / 3
    print(f"{nm}: {f[5 * i + 2][0]:60s} fabric read {fk * 2048 / 1e6:8.1f} MB (FETCH_SIZE x 2 KiB)  write {wk * 1024 / 1e6:8.1f} MB  L2 hit rate {hh / (hh + mm):.3f}")
PY
