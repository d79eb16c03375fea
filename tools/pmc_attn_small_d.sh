#!/bin/bash
# Dev recipe: SQ counters of the set-attention kernel at d = 32 / 16 (tools/one_attn.py 128 / 64).  Two --pmc passes per width, --kernel-trace only.
# Run ON THE GPU BOX from the repo root:   bash tools/pmc_attn_small_d.sh   -> gpurun_out/pmc_attn_small_d/summary.txt
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_attn_small_d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in 128 64; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a$C -o r --output-format csv -- python3 $ROOT/tools/one_attn.py $C > $OUT/a$C.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace -d $OUT/b$C -o r --output-format csv -- python3 $ROOT/tools/one_attn.py $C > $OUT/b$C.log 2>&1
done
cd $ROOT
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for C in (128, 64):
    for tag in ("a", "b"):
        f = glob.glob("$OUT/%s%d/**/*_counter_collection.csv" % (tag, C), recursive=True)
        if not f:
            print("no counters for", tag, C); continue
        a = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            a[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in a.items():
            if "set_attention" in k:
                print(C, k, {n: round(sum(v[2:]) / max(1, len(v[2:]))) for n, v in cs.items()})
PY
cat $OUT/summary.txt
