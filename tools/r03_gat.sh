#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_gat.py tests/test_gpu_fullsize.py tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r03e_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r03e_tests.log
for it in 1 0; do
NPI_GAT_ITEMS=$it timeout 300 python bench.py --conv gat --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03e_gat_items$it.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03e_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3))
    except Exception as e: print(f,'ERR',e)
PY
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o gp -- python3 $R/bench.py --conv gat --no-configs --no-control --virtual-world 0 --no-cpu-baseline --steps 10 > /tmp/gp.log 2>&1
cd $R
f=$(find /tmp/gp -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r03e_gat_kernel_stats.csv
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r03e_gat_kernel_stats.csv')))[:24]:
    print(f"{r['Name'][:95]:95s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
