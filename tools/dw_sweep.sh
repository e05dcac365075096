#!/bin/bash
# step time at C4 by dW grid (workgroups of the co-resident dW GEMM) and without the overlap; run on the GPU box
for c in 32 64 96 128 192; do
  echo "NPI_DW_CTAS=$c $(NPI_DW_CTAS=$c timeout -k 5 300 python3 bench.py --steps 10 --warmup 3 --no-configs --no-control --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["projection"]["per_gemm_ms"].items()}, round(d["roofline"]["avg_launch_ms"],3))')"
done
echo "no overlap $(NPI_OVERLAP_STREAMS=0 timeout -k 5 300 python3 bench.py --steps 10 --warmup 3 --no-configs --no-control --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["projection"]["per_gemm_ms"].items()}, round(d["roofline"]["avg_launch_ms"],3))')"
echo "exact dW (NPI_GEMM_SPLIT=0 for dW only is not selectable from the env; whole-step exact): $(NPI_GEMM_SPLIT=0 timeout -k 5 300 python3 bench.py --steps 10 --warmup 3 --no-configs --no-control --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3))')"
