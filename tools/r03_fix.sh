#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r03o_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03o_tests.log
for i in 1 2; do timeout 300 python bench.py --no-configs --no-control --virtual-world 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', round(d['ms_per_step'],3), [round(x,3) for x in d['ms_per_step_repeats']], 'seg avg', round(d['roofline']['avg_launch_ms'],3))"; done
timeout 300 python bench.py --conv gat --no-configs --no-control --virtual-world 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gat', round(d['ms_per_step'],3))"
python tools/virtual_rank_probe.py 2>&1 | grep world
