#!/bin/bash
# rocprofv3 kernel statistics of the pixel decoder's forward + backward at C4 (16 x 704^2); run through gpurun from the repo root
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/decoder_stats -- python3 $R/tools/prof_decoder.py "$@" > /dev/null 2>&1
cd $R
find gpurun_out/decoder_stats -name "*kernel_trace.csv" -delete
python tools/kstats.py "gpurun_out/decoder_stats/**/*kernel_stats.csv" 45
