#!/bin/bash
# rocprofv3 kernel statistics of the MSDeformAttn backward at C4 N=16, C4 N=1 and C5 N=1 (run through gpurun from the repo root)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "16 c4" "1 c4" "1 c5"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/msda_bwd_stats_$2_n$1 -- python3 $R/tools/prof_msda_bwd.py $1 $2 > /dev/null 2>&1
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -delete
for cfg in c4_n16 c4_n1 c5_n1; do python tools/kstats.py "gpurun_out/msda_bwd_stats_$cfg/**/*kernel_stats.csv" 8 | grep -v "at::native"; done
