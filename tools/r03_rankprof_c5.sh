#!/bin/bash
# round 3: kernels of ONE virtual rank's GATConv layer step at the C5 size (hubs cut, 8 ranks) -> gpurun_out/rankprof_c5_gat.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
A="--conv gat --steps 10 --nodes 4000000 --edges 100000000"
python tools/virtual_rank_probe.py $A 2>&1 | tail -1 > gpurun_out/rankprof_c5_gat.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_c5 -o rp -- python3 "$GRAFT_REPO_ROOT/tools/virtual_rank_probe.py" $A > /tmp/rp_c5.log 2>&1 )
python tools/kstats.py /tmp/rp_c5/rp_kernel_stats.csv 2>/dev/null | head -45 >> gpurun_out/rankprof_c5_gat.txt
cut -c1-160 gpurun_out/rankprof_c5_gat.txt
