#!/bin/bash
# Per-kernel split of the MSDeformAttn backward (C4, N = 16): rocprofv3 --kernel-trace --stats of tools/prof_msda_bwd.py on the
# binned path (default) and on the cell-sorted path (MSS_MSDA_BWD_ROWS=1). Run from the repo root on the GPU box; writes
# gpurun_out/$1/{rows,binned}_kernel_stats.csv
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05rows}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in rows binned; do
  if [ $mode = rows ]; then export MSS_MSDA_BWD_ROWS=1; else unset MSS_MSDA_BWD_ROWS; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -- python3 $GRAFT_REPO_ROOT/tools/prof_msda_bwd.py > /dev/null 2>&1
  f=$(ls -t $O/prof_$mode/*/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $O/${mode}_kernel_stats.csv; echo "== $mode"; head -12 "$f" | cut -d, -f1-5; fi
done
