for S in "overlap_min_rows=0,gat_rank2_min_rows=0" "overlap_min_rows=0,gat_rank2_min_rows=0,gat_rank2_epilogue=False" "overlap_min_rows=0,gat_rank2_min_rows=0,partial_stream=False" "overlap_min_rows=0,gat_rank2_min_rows=0,overlap_streams=False"; do
  SCHED="$S" python tools/vrank_stack_probe.py 100000 1000000 2>&1 | grep -v amdgpu.ids | cut -c1-420
done
for S in "gat_rank2_epilogue=False" "partial_stream=False" "overlap_streams=False"; do
  SCHED="$S" python tools/vrank_stack_probe.py 1000000 5000000 2>&1 | grep -v amdgpu.ids | cut -c1-420
done
