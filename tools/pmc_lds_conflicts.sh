#!/bin/bash
# Dev recipe: LDS bank conflicts of every kernel of the hot paths (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel name).
# Run ON THE GPU BOX from the repo root: bash tools/pmc_lds_conflicts.sh ; output gpurun_out/pmc_lds/summary.txt
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/bench -o r --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/latent -o r --output-format csv -- python3 $ROOT/tools/one_latent_steps.py > $OUT/latent.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/eval -o r --output-format csv -- python3 $ROOT/tools/one_eval.py > $OUT/eval.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/attn -o r --output-format csv -- python3 $ROOT/bench.py --backbone attention --steps 2 --warmup 1 --no-cpu-baseline > $OUT/attn.log 2>&1
cd $ROOT
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for tag in ("bench", "latent", "eval", "attn"):
    f = glob.glob("$OUT/%s/**/*_counter_collection.csv" % tag, recursive=True)
    if not f: print("no csv for", tag); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", tag)
    for k, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
        a, c = cs.get("SQ_LDS_IDX_ACTIVE", 0), cs.get("SQ_LDS_BANK_CONFLICT", 0)
        if a > 0: print(f"  {k:90s} lds_active {a:12.4g} conflict {c:12.4g} = {100 * c / a:5.1f} %")
PY
cat $OUT/summary.txt
