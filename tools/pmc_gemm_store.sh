#!/bin/bash
# The widest store GEMM under PMC counters, tile-boundary kernel (cfg 7) against the generic one (cfg 5): L2 hit rate, fabric bytes, MFMA busy.
# Run ON THE GPU BOX from the repo root: bash tools/pmc_gemm_store.sh ; prints one line per (kernel, counter).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_store
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in 7 5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f$cfg -o f --output-format csv -- python3 $ROOT/tools/one_gemm_store.py $cfg 10 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/w$cfg -o w --output-format csv -- python3 $ROOT/tools/one_gemm_store.py $cfg 10 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/m$cfg -o m --output-format csv -- python3 $ROOT/tools/one_gemm_store.py $cfg 10 > /dev/null 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob
def mean(root, counter):
    f = glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    v = v[len(v) // 3:]
    return sum(v) / max(1, len(v))
def dur(root):
    f = glob.glob(f"{root}/**/*_kernel_trace.csv", recursive=True)[0]
    v = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"]]
    v = v[len(v) // 3:]
    return sum(v) / len(v) / 1e3
for cfg, name in ((7, "gemm_xp_kernel<F16>"), (5, "gemm_f16_kernel<256,256,..,F16>")):
    o = "$OUT"
    fetch, write = mean(f"{o}/f{cfg}", "FETCH_SIZE"), mean(f"{o}/w{cfg}", "WRITE_SIZE")
    hit, miss = mean(f"{o}/w{cfg}", "TCC_HIT_sum"), mean(f"{o}/w{cfg}", "TCC_MISS_sum")
    mfma, gui = mean(f"{o}/m{cfg}", "SQ_VALU_MFMA_BUSY_CYCLES"), mean(f"{o}/m{cfg}", "GRBM_GUI_ACTIVE")
    print(f"{name}: {dur(f'{o}/m{cfg}'):7.1f} us under the profiler | fabric read {fetch * 2048 / 1e6:7.1f} MB (FETCH_SIZE x 2 KiB) write {write * 1024 / 1e6:6.1f} MB | "
          f"L2 hit rate {hit / (hit + miss):.3f} | MFMA busy cycles {mfma:.3e} | GRBM_GUI_ACTIVE {gui:.3e}")
PY
