#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/c5.py <<'PY'
import sys, torch, time
sys.path.insert(0, sys.argv[1])
import npi_gnn_amd as npi
from npi_gnn_amd.synth import bipartite_edge_index
dev=torch.device('cuda:0')
N5,E5,F5=4_000_000,100_000_000,256
ei=bipartite_edge_index(N5,E5,seed=2).to(dev)
g=npi.CSRGraph(ei,N5); _=g.by_src; del ei
gen=torch.Generator().manual_seed(11)
conv=npi.GATConv(F5,F5,heads=1).to(dev)
x=torch.randn(N5,F5,generator=gen).to(dev).requires_grad_(True)
go=torch.randn(N5,F5,generator=gen).to(dev)
def step():
    for p in conv.parameters(): p.grad=None
    x.grad=None
    conv(x,g).backward(go)
for _ in range(2): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); print('C5 one GAT layer ms', (time.perf_counter()-t0)/3*1e3)
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -o c5 -- python3 /tmp/c5.py $R > /tmp/c5.log 2>&1
cd $R; tail -2 /tmp/c5.log
f=$(find /tmp/c5p -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r03k_c5_layer_kernel_stats.csv
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r03k_c5_layer_kernel_stats.csv')))[:22]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:8.3f} pct {r['Percentage']}")
PY
