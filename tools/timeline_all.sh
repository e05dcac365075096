#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace) of: one virtual rank's SAGE / GAT step, the single-GPU GAT layer at C4.
# usage (GPU box, repo root): bash tools/timeline_all.sh <tag> [vsage,vgat,gat1]  -> gpurun_out/<tag>_tl_*.txt
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
ROOT="$(pwd)"; T=${1:-tl}; export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # name n_kernels cmd...
  local name=$1 n=$2; shift 2
  local O="$ROOT/gpurun_out/tl_${T}_${name}"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- "$@" > "$O.log" 2>&1)
  local f=$(find "$O" -name '*kernel_trace.csv' | head -1)
  python3 tools/step_timeline.py "$f" $n > gpurun_out/${T}_tl_${name}.txt
  tail -2 "$O.log"
  rm -rf "$O"
}
W=${2:-vsage,vgat,gat1}
[[ $W == *vsage* ]] && run vsage 60 python3 "$ROOT/tools/virtual_rank_probe.py" --conv sage --steps 6
[[ $W == *vgat* ]] && run vgat 260 python3 "$ROOT/tools/virtual_rank_probe.py" --conv gat --steps 6
[[ $W == *gat1* ]] && run gat1 60 python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-control --no-live-pmc --virtual-world 0 --conv gat
[[ $W == *c5* ]] && run c5 60 python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-control --no-live-pmc --virtual-world 0 --conv gat --nodes 4000000 --edges 100000000 --graph-seed 2
[[ $W == *c5s* ]] && run c5s 130 python3 "$ROOT/tools/c5_stack_probe.py" 2
[[ $W == *net1* ]] && run net1 100 python3 "$ROOT/tools/net1_step_probe.py" 20
[[ $W == *c3g* ]] && run c3g 60 python3 "$ROOT/tools/c13_probe.py" c3
[[ $W == *c1g* ]] && run c1g 50 python3 "$ROOT/tools/c13_probe.py" c1
[[ $W == *gath8* ]] && run gath8 40 python3 "$ROOT/tools/gat_heads_probe.py" 8 3
[[ $W == *gath2* ]] && run gath2 40 python3 "$ROOT/tools/gat_heads_probe.py" 2 3
[[ $W == *vsw4* ]] && run vsw4 70 python3 "$ROOT/tools/virtual_rank_probe.py" --conv sage --steps 6 --wire-gbps 400
true
