#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r03f}
timeout 1200 python -m pytest tests/test_gpu_gat.py -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/${TAG}_tests.log
timeout 300 python bench.py --conv gat --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/${TAG}_gat.json 2>/dev/null
python - $TAG <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/{sys.argv[1]}_gat.json').read().strip().splitlines()[-1]); print('gat ms', round(d['ms_per_step'],3), d.get('ms_per_step_repeats'))
PY
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o gp -- python3 $R/bench.py --conv gat --no-configs --no-control --virtual-world 0 --no-cpu-baseline --steps 10 > /tmp/gp.log 2>&1
cd $R
f=$(find /tmp/gp -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${TAG}_gat_kernel_stats.csv
python - $TAG <<'PY'
import csv,sys
for r in list(csv.DictReader(open(f'gpurun_out/{sys.argv[1]}_gat_kernel_stats.csv')))[:20]:
    print(f"{r['Name'][:95]:95s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
