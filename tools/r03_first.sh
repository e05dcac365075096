#!/bin/bash
# round 3, first GPU session: sharded-path tests, the bench line with the virtual world, W=1 sharded overhead A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_dist_rccl.py -m gpu -x -q > gpurun_out/r03a_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03a_tests.log
timeout 600 python bench.py --skip-c5 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench rc=$?"
for ps in 1 0; do
  NPI_PARTIAL_STREAM=$ps timeout 300 python bench.py --force-sharded --partition hubs --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03a_w1_hubs_ps$ps.json 2> gpurun_out/r03a_w1_hubs_ps$ps.err; echo "w1 ps=$ps rc=$?"
done
timeout 300 python bench.py --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03a_plain.json 2>/dev/null
timeout 300 python bench.py --force-sharded --partition edges --no-configs --no-control --virtual-world 0 --no-cpu-baseline > gpurun_out/r03a_w1_edges.json 2> gpurun_out/r03a_w1_edges.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['ms_per_step'],3), d.get('parity_max_err'))
    except Exception as e: print(f, 'ERR', e)
PY
