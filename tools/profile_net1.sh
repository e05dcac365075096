#!/bin/bash
# kernel trace of the reference's real workload (NPInter2 fold 0, net1.fit) -- eager and with captured steps.
# Run on the GPU box from the repo root; copy <out>/stats_kernel_stats.csv into profiles/.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_${1:-r02}_net1"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/eager" -o stats -- python3 "$ROOT/examples/train_npinter2.py" --epochs 3 --no-train-eval > "$OUT/eager.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/capture" -o stats -- python3 "$ROOT/examples/train_npinter2.py" --epochs 4 --no-train-eval --capture > "$OUT/capture.log" 2>&1
cd "$ROOT"
du -sh "$OUT"
