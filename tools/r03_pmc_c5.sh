#!/bin/bash
# round 3: the GATConv aggregation kernels at the C5 size (N = 4M, E = 100M, the graph of configs.C5_1gpu): kernel stats and
# FETCH_SIZE / WRITE_SIZE, each in its own rocprofv3 run -> profiles/<tag>_c5_* and the c5_* constants of pmc_traffic.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r03u}
X="--no-configs --no-control --virtual-world 0"
BENCH_ARGS="$X --conv gat --nodes 4000000 --edges 100000000 --graph-seed 2" STEPS=3 bash tools/profile_bench.sh ${T}_c5 > gpurun_out/${T}_c5_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${T}_c5 ${T}_c5 1.992 c5gat > /dev/null
cat profiles/pmc_traffic.json
mkdir -p gpurun_out/profiles_${T}; cp profiles/${T}_c5* profiles/pmc_traffic.json gpurun_out/profiles_${T}/
tail -n 3 gpurun_out/prof_${T}_c5/stats.log; tail -n 3 gpurun_out/prof_${T}_c5/pmc_fetch.log
rm -rf gpurun_out/prof_${T}_c5
