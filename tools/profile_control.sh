#!/bin/bash
# rocprofv3 passes for roofline.control_uniform (bench.py --control-only): kernel trace + FETCH_SIZE / WRITE_SIZE in
# separate --pmc runs.  Run on the GPU box from the repo root; tools/rocprof_control_summary.py condenses the output.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_${1:-r02}_control"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --control-only > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --control-only > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --control-only > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
du -sh "$OUT"
