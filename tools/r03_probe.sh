#!/bin/bash
# kernel trace of ONE virtual rank's steps (tools/virtual_rank_probe.py) -> gpurun_out/<tag>_vrank_*
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r03b}; shift
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vr -o vr -- python3 $R/tools/virtual_rank_probe.py --steps 20 "$@" > $R/gpurun_out/${TAG}_vrank.log 2>&1
cd $R
tail -2 gpurun_out/${TAG}_vrank.log
f=$(find /tmp/vr -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${TAG}_vrank_kernel_stats.csv
f=$(find /tmp/vr -name '*kernel_trace.csv' | head -1); python - "$f" "$TAG" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=open(f'gpurun_out/{sys.argv[2]}_vrank_trace_tail.txt','w')
t0=None
for r in rows[-90:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    out.write(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:120]}\n")
PY
python - gpurun_out/${TAG}_vrank_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:28]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
