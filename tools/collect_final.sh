#!/bin/bash
# Everything profiles/rNN quotes for the final tree, in one gpurun call from the repo root:  tools/collect_final.sh <dir under gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
bash $R/tools/profile_round.sh $1/round > $O/profile_round.log 2>&1
cd $R
python tools/layer_table.py 2>&1 | grep -v amdgpu.ids > $O/layer_table.txt
python tools/layer_table_eval.py 1 2>&1 | grep -v amdgpu.ids > $O/layer_table_eval.txt
python tools/prof_decoder.py 16 704 704 2 layers 2>&1 | grep -v amdgpu.ids > $O/decoder_layer_table.txt
bash tools/prof_decoder.sh 16 704 704 4 > $O/decoder_kernel_stats.txt 2>&1
cp gpurun_out/decoder_stats/*/*kernel_stats.csv $O/decoder_kernel_stats.csv 2>/dev/null
python tools/bench_wino_in.py MSS_WINO_IN_ORDER 0 1 2>&1 | grep -v amdgpu.ids > $O/bench_wino_in_order.jsonl
bash tools/prof_eval.sh > $O/prof_eval.log 2>&1
ls -la $O
