#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gat.py tests/test_dist_gpu.py -m gpu -x -q -k "not item_parallel" > gpurun_out/r03l_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r03l_tests.log
timeout 900 python bench.py > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03l_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ms_per_step_repeats"])
print({k:round(v["ms_per_step"],3) for k,v in d["configs"].items() if "ms_per_step" in v})
vw=d["configs"]["C4_w8_virtual"]
for k,v in vw.items():
    if isinstance(v,dict): print(k, "ceil", round(v["compute_ceiling"],2), "max", round(max(v["per_rank_ms"]),3), "min", round(min(v["per_rank_ms"]),3))
PY
