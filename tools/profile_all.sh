#!/bin/bash
# The rocprofv3 evidence of a round, in one GPU call (run from the repo root on the GPU box):
#   bash tools/profile_all.sh <tag>            e.g. r04a
# For the SAGE headline, GATConv and GCNConv at the C4 shape, GATConv at the C5 size and the uniform-source control:
#   kernel trace + stats in one run; FETCH_SIZE and WRITE_SIZE each in a run of their own (never combined with tracing), as
#   /opt/skills/guides/MI355X_MICROARCH.md prescribes (tools/profile_bench.sh, tools/profile_control.sh)
#   -> profiles/<tag>[_gat|_c5|_gcn|_bf16]_kernel_stats.csv, <tag>*_pmc_summary.json, <tag>_control_*  (tools/rocprof_summary.py)
#   -> profiles/pmc_traffic.json: HBM-side bytes per aggregation launch + the sha of the kernel sources they were measured on
#   -> profiles/<tag>_mfma_util.json: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the two projection kernels (tools/gemm_pair.py)
# then the bench itself: the driver's command -> profiles/<tag>_bench_line.json (the compact line; bench.py reads
# pmc_traffic.json: frac_traffic is current again) and `bench.py --extras` -> profiles/<tag>_bench.json (the full record).
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
export TMPDIR=/tmp
T=${1:-r04a}
X="--no-configs --no-control --virtual-world 0 --no-parity --no-live-pmc"
mkdir -p gpurun_out
BENCH_ARGS="$X" STEPS=5 bash tools/profile_bench.sh ${T} > gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T} ${T} 1.992 sage > /dev/null
BENCH_ARGS="$X --conv gat" STEPS=5 bash tools/profile_bench.sh ${T}_gat >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_gat ${T}_gat 1.992 gat > /dev/null
BENCH_ARGS="$X --conv gat --nodes 4000000 --edges 100000000 --graph-seed 2" STEPS=3 bash tools/profile_bench.sh ${T}_c5 >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_c5 ${T}_c5 1.992 c5gat > /dev/null
BENCH_ARGS="$X --conv gcn" STEPS=5 bash tools/profile_bench.sh ${T}_gcn >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_gcn ${T}_gcn 1.992 gcn > /dev/null
BENCH_ARGS="$X --storage bf16" STEPS=5 bash tools/profile_bench.sh ${T}_bf16 >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_bf16 ${T}_bf16 1.992 bf16 > /dev/null
bash tools/profile_control.sh ${T} >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_control_summary.py gpurun_out/prof_${T}_control ${T} 1.992 > /dev/null
# matrix-pipe utilisation of the two projection kernels by counter (a PMC pass of its own, with --kernel-trace only)
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/prof_${T}_mfma" -o m -- python3 "$OLDPWD/tools/gemm_pair.py" >> "$OLDPWD/gpurun_out/${T}_prof.log" 2>&1)
python tools/mfma_util_summary.py "$(find gpurun_out/prof_${T}_mfma -name '*counter_collection.csv' | head -1)" profiles/${T}_mfma_util.json > /dev/null
rm -rf gpurun_out/prof_${T}_mfma
cat profiles/pmc_traffic.json
rm -rf gpurun_out/prof_${T} gpurun_out/prof_${T}_c5 gpurun_out/prof_${T}_gat gpurun_out/prof_${T}_gcn gpurun_out/prof_${T}_bf16 gpurun_out/prof_${T}_control
# the driver's command first (the compact line as the driver will see it), then the lab harness (--extras: the full record)
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench_line.err; echo "bench rc=$?"
cp gpurun_out/${T}_bench_line.json profiles/${T}_bench_line.json
timeout 1700 python bench.py --extras > gpurun_out/${T}_bench_extras_line.json 2> gpurun_out/${T}_bench.err; echo "bench --extras rc=$?"
cp bench_detail.json gpurun_out/${T}_bench.json; cp bench_detail.json profiles/${T}_bench.json
mkdir -p gpurun_out/profiles_${T}; cp profiles/${T}* profiles/pmc_traffic.json gpurun_out/profiles_${T}/
tail -c 400 gpurun_out/${T}_bench.err
