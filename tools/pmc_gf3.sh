#!/bin/bash
# Recipe for bench.py's roofline.traffic: HBM/fabric bytes per launch of the dominant GEMM (global_feat.3 + max).
# Two separate rocprofv3 --pmc passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2) over tools/one_gemm.py,
# corrected as MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads).
# Run ON THE GPU BOX from the repo root:   GIT_HEAD=<hash> bash tools/pmc_gf3.sh
# Writes gpurun_out/pmc_gf3/* and profiles/gf3_pmc_latest.json (copy the latter back into the repo).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_gf3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python3 $ROOT/tools/one_gemm.py -1 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/write -o w --output-format csv -- python3 $ROOT/tools/one_gemm.py -1 10 > /dev/null 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, hashlib, json, os
def mean(root, counter):
    f = glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if ("gemm_xw_kernel" in r["Kernel_Name"] or "gemm_xp_kernel" in r["Kernel_Name"] or "gemm_f16_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter]
    v = v[len(v) // 3:]            # drop warm-up launches
    return sum(v) / len(v)
fetch_kb, write_kb = mean("$OUT/fetch", "FETCH_SIZE"), mean("$OUT/write", "WRITE_SIZE")
hit, miss = mean("$OUT/write", "TCC_HIT_sum"), mean("$OUT/write", "TCC_MISS_sum")
rec = {"kernel": "gemm_xw_kernel (256x256 tile, 2x4 waves, persistent + XCD patch mapping; activation panel through LDS, fragment-order weights straight from global memory; tools/one_gemm.py)",
       "recipe": "tools/pmc_gf3.sh", "git_head": os.environ.get("GIT_HEAD", "unknown"),
       "kernel_source_sha16": hashlib.sha256(open("$ROOT/3d-shape-generation_amd/csrc/gemm_f16.hip", "rb").read()).hexdigest()[:16],
       "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "l2_hit_rate": hit / (hit + miss),
       "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
       "note": "FETCH_SIZE x 1024 x 2 (gfx950 half-count correction) + WRITE_SIZE x 1024; fabric-side counters, Infinity-Cache hits included"}
json.dump(rec, open("$OUT/gf3_pmc_latest.json", "w"), indent=1)
json.dump(rec, open("$ROOT/gpurun_out/gf3_pmc_latest.json", "w"), indent=1)
print(json.dumps(rec))
PY
