#!/bin/bash
# round 4, second GPU call: the whole -m gpu suite on the in-launch chain resolution, the bench line, a rank's kernel timeline
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/r04b_tests.log"
python bench.py --steps 20 --warmup 3 > "$OUT/r04b_bench.json" 2> "$OUT/r04b_bench.err"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/r04b_vrank" -o vr -- python3 "$ROOT/tools/virtual_rank_probe.py" --steps 10 > "$OUT/r04b_vrank.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/r04b_vrank" -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$f" 60 > "$OUT/r04b_vrank_timeline.txt" 2>&1
rm -rf "$OUT/r04b_vrank"
tail -5 "$OUT/r04b_tests.log"; tail -c 300 "$OUT/r04b_bench.err"; tail -2 "$OUT/r04b_vrank.log"
