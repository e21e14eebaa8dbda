#!/bin/bash
# rocprofv3 kernel statistics of the eval forward (OOD-score path) at 1x1024x2048 (run through gpurun from the repo root)
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/eval_stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/eval_stats -- python3 $R/tools/prof_eval.py > /dev/null 2>&1
cd $R
find gpurun_out/eval_stats -name "*kernel_trace.csv" -delete
python tools/kstats.py "gpurun_out/eval_stats/**/*kernel_stats.csv" 24
