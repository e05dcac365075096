#!/bin/bash
# HBM-side bytes of one virtual rank's SAGE step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (per kernel, summed
# over the last N steps).  usage (GPU box, repo root): bash tools/vrank_pmc.sh [sage|gat]
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
ROOT="$(pwd)"; export TMPDIR=/tmp; C=${1:-sage}
for ctr in FETCH_SIZE WRITE_SIZE; do
  O="$ROOT/gpurun_out/vpmc_${C}_$ctr"
  (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d "$O" -o p -- python3 "$ROOT/tools/virtual_rank_probe.py" --conv $C --steps 4 > "$O.log" 2>&1)
  f=$(find "$O" -name '*counter_collection.csv' | head -1)
  python3 - "$f" $ctr <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    tot[r["Kernel_Name"][:70]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"][:70]] += 1
print(sys.argv[2], "KB per launch (launches):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("  %12.0f  x%-4d %s" % (v / cnt[k], cnt[k], k))
PY
  rm -rf "$O"
done
