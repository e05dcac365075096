#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
X="--no-configs --no-control --virtual-world 0 --no-cpu-baseline"
for i in 1 2; do
timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', round(d['ms_per_step'],3), [round(x,3) for x in d['ms_per_step_repeats']])"
timeout 300 python bench.py $X --force-sharded --partition hubs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w1 hubs', round(d['ms_per_step'],3), [round(x,3) for x in d['ms_per_step_repeats']], d['parity_max_err'])"
done
timeout 300 python bench.py $X --force-sharded --partition hubs --conv gat 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w1 hubs gat', round(d['ms_per_step'],3), d['parity_max_err'])"
timeout 300 python bench.py $X --force-sharded --partition hubs --conv gcn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w1 hubs gcn', round(d['ms_per_step'],3), d['parity_max_err'])"
timeout 300 python bench.py $X --force-sharded --partition rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w1 rows', round(d['ms_per_step'],3), d['parity_max_err'])"
timeout 300 python bench.py $X --force-sharded --partition edges 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w1 edges', round(d['ms_per_step'],3), d['parity_max_err'])"
