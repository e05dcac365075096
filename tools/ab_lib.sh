#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <variant .so> [rounds]   (B = the in-tree library)
# GEMM micro-benchmarks and the C4 step, alternating
V=$1; R=${2:-3}
for i in $(seq $R); do
  echo "== A ($V)"; NPI_GNN_LIB=$V timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -3
  echo "== B (in-tree)"; timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -3
done
for i in $(seq $R); do
  echo "== A step"; NPI_GNN_LIB=$V timeout -k 5 300 python3 bench.py --no-cpu-baseline --no-control --no-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_repeats'))"
  echo "== B step"; timeout -k 5 300 python3 bench.py --no-cpu-baseline --no-control --no-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_repeats'))"
done
