#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in "NPI_OVERLAP_STREAMS=0" "NPI_PARTIAL_STREAM=0" ""; do
  echo "== $v capture"
  env $v timeout 120 python tools/virtual_rank_probe.py --capture 2>&1 | tail -3
done
