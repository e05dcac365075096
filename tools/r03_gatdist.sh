#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_dist_rccl.py -m gpu -x -q > gpurun_out/r03j_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r03j_tests.log
python tools/virtual_rank_probe.py --conv gat > gpurun_out/r03j_probe.log 2>&1
NPI_GAT_DIRECT=0 python tools/virtual_rank_probe.py --conv gat >> gpurun_out/r03j_probe.log 2>&1
python tools/virtual_rank_probe.py --conv gat --world 1 >> gpurun_out/r03j_probe.log 2>&1
python tools/virtual_rank_probe.py --conv sage >> gpurun_out/r03j_probe.log 2>&1
grep world gpurun_out/r03j_probe.log
bash tools/r03_probe.sh r03j --conv gat | tail -24
