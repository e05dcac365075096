#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for h in 64 128 256 512; do
  timeout 300 python bench.py --hidden $h --no-configs --no-control --virtual-world 0 --no-cpu-baseline 2>/dev/null > /tmp/h.json
  python - $h <<'PY'
import sys,json
d=json.loads(open('/tmp/h.json').read().strip().splitlines()[-1])
print("hidden", sys.argv[1], round(d["ms_per_step"],3), "ms", round(d["value"]/1e9,2), "G edges/s; seg avg", round(d["roofline"]["avg_launch_ms"],3), "alg frac", round(d["roofline"]["frac_algorithmic"],3), {k:round(v,3) for k,v in d["projection"]["per_gemm_ms"].items()})
PY
done
