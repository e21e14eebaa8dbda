#!/bin/bash
# Per-kernel split of the MSDeformAttn backward (C4, N = 16): rocprofv3 --kernel-trace --stats of tools/prof_msda_bwd.py on the
# binned path (the default). Run from the repo root on the GPU box; writes gpurun_out/$1/binned_kernel_stats.csv
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06msda}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_binned -- python3 $GRAFT_REPO_ROOT/tools/prof_msda_bwd.py > /dev/null 2>&1
f=$(ls -t $O/prof_binned/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/binned_kernel_stats.csv; head -12 "$f" | cut -d, -f1-5; fi
