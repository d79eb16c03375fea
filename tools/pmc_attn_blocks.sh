#!/bin/bash
# Dev recipe: LDS bank conflicts and matrix-pipe occupancy of the attention U-Net's kernels (the round-4 head / tail / LayerNorm-in-Linear launches among them)
# over tools/one_attn_forward.py.  Two --pmc passes (no other trace domains beside --kernel-trace).  Run ON THE GPU BOX from the repo root:
#   bash tools/pmc_attn_blocks.sh ; output gpurun_out/pmc_attn/summary.txt
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/lds -o r --output-format csv -- python3 $ROOT/tools/one_attn_forward.py 1 > $OUT/lds.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $OUT/mfma -o r --output-format csv -- python3 $ROOT/tools/one_attn_forward.py 1 > $OUT/mfma.log 2>&1
cd $ROOT
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
def agg(tag):
    f = glob.glob("$OUT/%s/**/*_counter_collection.csv" % tag, recursive=True)
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    if f:
        for r in csv.DictReader(open(f[0])):
            a[r["Kernel_Name"][:80]][r["Counter_Name"]] += float(r["Counter_Value"])
    return a
lds, mf = agg("lds"), agg("mfma")
print("kernel | LDS bank conflict cycles / LDS active cycles | MFMA busy / SQ busy cycles (x4: per-SIMD pipes v. one busy counter per SE slice, relative numbers only) | waves waiting / wave cycles")
for k, cs in sorted(lds.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a, c = cs.get("SQ_LDS_IDX_ACTIVE", 0), cs.get("SQ_LDS_BANK_CONFLICT", 0)
    m = mf.get(k, {})
    mb, sb, wc, wa = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_INST_ANY", 0)
    if a > 0:
        print(f"  {k:80s} conflict {100 * c / a:5.1f} % | mfma/busy {mb / sb if sb else 0:6.3f} | wait {100 * wa / wc if wc else 0:5.1f} %")
PY
cat $OUT/summary.txt
