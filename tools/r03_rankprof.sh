#!/bin/bash
# round 3: kernels of ONE virtual rank's step (hubs cut, 8 ranks), SAGE and GAT -> gpurun_out/rankprof_*.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for conv in sage gat; do
  python tools/virtual_rank_probe.py --conv $conv --steps 30 2>&1 | tail -3 > gpurun_out/rankprof_${conv}.txt
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$conv -o rp -- python3 "$GRAFT_REPO_ROOT/tools/virtual_rank_probe.py" --conv $conv --steps 30 > /tmp/rp_$conv.log 2>&1 )
  python tools/kstats.py /tmp/rp_$conv/rp_kernel_stats.csv 2>/dev/null | head -40 >> gpurun_out/rankprof_${conv}.txt
done
cat gpurun_out/rankprof_sage.txt
