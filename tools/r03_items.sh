#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for thr in 1048576 4194304 16777216; do
  echo "== threshold $thr"
  NPI_SMALL_GRAPH_ENTRIES=$thr python tools/virtual_rank_probe.py 2>&1 | grep world
  NPI_SMALL_GRAPH_ENTRIES=$thr python tools/virtual_rank_probe.py --conv gat 2>&1 | grep world
  NPI_SMALL_GRAPH_ENTRIES=$thr python tools/virtual_rank_probe.py --world 4 2>&1 | grep world
  NPI_SMALL_GRAPH_ENTRIES=$thr python tools/virtual_rank_probe.py --world 2 2>&1 | grep world
done
