#!/bin/bash
# round 3: bench line + rocprofv3 passes (kernel stats; FETCH_SIZE / WRITE_SIZE each in its own run) for the SAGE headline,
# the uniform control, GATConv and GCNConv at the C4 shape -> profiles/r03<x>_*
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r03h}
X="--no-configs --no-control --virtual-world 0"
BENCH_ARGS="$X" STEPS=5 bash tools/profile_bench.sh ${T} > gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T} ${T} 1.992 sage > /dev/null
BENCH_ARGS="$X --conv gat" STEPS=5 bash tools/profile_bench.sh ${T}_gat >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_gat ${T}_gat 1.992 gat > /dev/null
BENCH_ARGS="$X --conv gat --nodes 4000000 --edges 100000000 --graph-seed 2" STEPS=3 bash tools/profile_bench.sh ${T}_c5 >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_c5 ${T}_c5 1.992 c5gat > /dev/null
BENCH_ARGS="$X --conv gcn" STEPS=5 bash tools/profile_bench.sh ${T}_gcn >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_gcn ${T}_gcn 1.992 gcn > /dev/null
bash tools/profile_control.sh ${T} >> gpurun_out/${T}_prof.log 2>&1
python tools/rocprof_control_summary.py gpurun_out/prof_${T}_control ${T} 1.992 > /dev/null
cat profiles/pmc_traffic.json
mkdir -p gpurun_out/profiles_${T}; cp profiles/${T}* profiles/pmc_traffic.json gpurun_out/profiles_${T}/
rm -rf gpurun_out/prof_${T} gpurun_out/prof_${T}_c5 gpurun_out/prof_${T}_gat gpurun_out/prof_${T}_gcn gpurun_out/prof_${T}_control
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/${T}_bench.err
