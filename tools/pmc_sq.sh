#!/bin/bash
# SQ counters of the kernels the headline numbers rest on: the dominant GEMM (gemm_xw_kernel), the widest store GEMM
# (gemm_xs_kernel), the d = 64 set-attention kernel (set_attention_sp_kernel) and the d = 32 / 16 ones (set_attention_spn_kernel).  One rocprofv3 --pmc pass each (7 SQ counters + GRBM_GUI_ACTIVE;
# --kernel-trace only, the program itself after `--`).  MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), at the
# clocks the chip holds UNDER the profiler (1.89-1.95 GHz; MI355X_MICROARCH.md DVFS item 2).
# Run ON THE GPU BOX from the repo root:   GIT_HEAD=<hash> bash tools/pmc_sq.sh     -> gpurun_out/sq_pmc_latest.json (copy into profiles/)
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/gf3 -o s --output-format csv -- python3 $ROOT/tools/one_gemm.py -1 10 > /dev/null 2>&1
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/store -o s --output-format csv -- python3 $ROOT/tools/one_gemm_store_wfrag.py 10 > /dev/null 2>&1
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/attn -o s --output-format csv -- python3 $ROOT/tools/one_attn.py 256 > /dev/null 2>&1
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/attn32 -o s --output-format csv -- python3 $ROOT/tools/one_attn.py 128 > /dev/null 2>&1
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/attn16 -o s --output-format csv -- python3 $ROOT/tools/one_attn.py 64 > /dev/null 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, hashlib, json, os
def sha16(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
def counters(root, pat):
    f = glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for k, v in agg.items()}      # drop warm-up launches
def dur(root, pat):
    f = glob.glob(f"{root}/**/*_kernel_trace.csv", recursive=True)[0]
    v = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    v = v[len(v) // 3:]
    return sum(v) / len(v) / 1e3
rec = {"recipe": "tools/pmc_sq.sh", "git_head": os.environ.get("GIT_HEAD", "unknown"), "kernels": {},
       "note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); wait_* / active as fractions of SQ_WAVE_CYCLES; "
               "lds_active = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8); profiled clocks"}
src = {"gemm_xw_kernel": "gemm_f16.hip", "gemm_xs_kernel": "gemm_f16.hip", "set_attention_sp_kernel": "attention.hip",
       "set_attention_spn_kernel<32>": "attention.hip", "set_attention_spn_kernel<16>": "attention.hip"}
for key, sub, pat in (("gemm_xw_kernel", "gf3", "gemm_xw_kernel"), ("gemm_xs_kernel", "store", "gemm_xs_kernel"), ("set_attention_sp_kernel", "attn", "set_attention_sp_kernel"),
                      ("set_attention_spn_kernel<32>", "attn32", "set_attention_spn_kernelILi32"), ("set_attention_spn_kernel<16>", "attn16", "set_attention_spn_kernelILi16")):
    c = counters("$OUT/" + sub, pat)
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    rec["kernels"][key] = {
        "kernel_source_sha16": sha16("$ROOT/3d-shape-generation_amd/csrc/" + src[key]),
        "profiled_us": dur("$OUT/" + sub, pat), "kernel_cycles": cyc,
        "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
        "wait_any": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "wait_inst_any": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "active_inst_any": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "lds_active": c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), "raw": c}
json.dump(rec, open("$ROOT/gpurun_out/sq_pmc_latest.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "raw"} for k, v in rec["kernels"].items()}, indent=1))
PY
