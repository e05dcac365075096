#!/bin/bash
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
python -m pytest tests/test_gpu_fullsize.py -q -k "c5_eight" 2>&1 | tail -8 > "$OUT/r04d_c5test.log"
export TMPDIR=/tmp; cd /tmp
for m in bf16 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r04d_c2_$m" -o c2 -- python3 "$ROOT/tools/c2_probe.py" $m > "$OUT/r04d_c2_$m.log" 2>&1
  f=$(find "$OUT/r04d_c2_$m" -name "*kernel_stats.csv" | head -1)
  head -40 "$f" | cut -c1-160 > "$OUT/r04d_c2_${m}_kernel_stats.csv"
  rm -rf "$OUT/r04d_c2_$m"
done
cd "$ROOT"
cat "$OUT/r04d_c5test.log"; grep "ms per step" "$OUT"/r04d_c2_*.log
