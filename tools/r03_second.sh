#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r03d_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03d_tests.log
for i in 1 2; do
timeout 300 python bench.py --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03d_plain$i.json 2>/dev/null
timeout 300 python bench.py --force-sharded --partition hubs --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03d_w1_hubs$i.json 2> gpurun_out/r03d_w1_hubs.err
done
NPI_DIRECT_HUB_ROWS=0 timeout 300 python bench.py --force-sharded --partition hubs --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03d_w1_hubs_classic.json 2> /dev/null
timeout 300 python bench.py --force-sharded --conv gcn --partition hubs --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03d_w1_gcn.json 2> /dev/null
python tools/virtual_rank_probe.py > gpurun_out/r03d_probe.log 2>&1
python tools/virtual_rank_probe.py --rank 3 >> gpurun_out/r03d_probe.log 2>&1
python tools/virtual_rank_probe.py --conv gcn >> gpurun_out/r03d_probe.log 2>&1
grep world gpurun_out/r03d_probe.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03d_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['ms_per_step'],3), d.get('parity_max_err'))
    except Exception as e: print(f, 'ERR', e)
PY
