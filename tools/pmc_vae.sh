#!/bin/bash
# Dev recipe: PMC counters of the VAE convolution kernels (tools/one_vae_decode.py), two passes.
# Run ON THE GPU BOX from the repo root: bash tools/pmc_vae.sh ; output under gpurun_out/pmc_vae/
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_vae
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o a --output-format csv -- python3 $ROOT/tools/one_vae_decode.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/b -o b --output-format csv -- python3 $ROOT/tools/one_vae_decode.py > $OUT/b.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*_counter_collection.csv" % tag, recursive=True)
    if not f: print("no csv for", tag); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv3d" not in k: continue
        key = (k[:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in agg.items():
        print(key)
        for c, v in cs.items(): print(f"     {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
