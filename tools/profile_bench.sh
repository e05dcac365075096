#!/bin/bash
# rocprofv3 passes for the bench (run on the GPU box from the repo root).
# kernel trace + stats in one pass; PMC counters each in their own pass (never combined with
# other tracing), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_${1:-r01}"
mkdir -p "$OUT"
export TMPDIR=/tmp
# (--no-live-pmc: a bench that is itself under rocprofv3 must not start nested --pmc child passes; bench.py also detects it)
ARGS="--steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-live-pmc ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc ${BENCH_ARGS:-} > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc ${BENCH_ARGS:-} > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
find "$OUT" -type f | head -50
du -sh "$OUT"
