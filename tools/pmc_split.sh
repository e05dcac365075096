#!/bin/bash
# SQ/GRBM counters for the split-bf16 GEMM kernel (separate --pmc passes); run on the GPU box from the repo root
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_s1 -o g -- python3 $R/tools/kernel_bench.py --gemm --rounds 3 > $R/gpurun_out/pmc_s1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/pmc_s2 -o g -- python3 $R/tools/kernel_bench.py --gemm --rounds 3 > $R/gpurun_out/pmc_s2.log 2>&1
cd $R
python3 - <<PY
import csv,collections,glob
for d in ("pmc_s1","pmc_s2"):
    fs=glob.glob(f"gpurun_out/{d}/*counter_collection.csv")
    if not fs: print(d,"no output"); print(open(f"gpurun_out/{d}.log").read()[-600:]); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "gemm_split" in r["Kernel_Name"]:
            k=r["Kernel_Name"].split("(")[0][-28:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in acc.items():
        print(d,k,"dur_us=%.0f"%(sum(dur[k])/len(dur[k])), {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()})
PY
