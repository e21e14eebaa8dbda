#!/bin/bash
# Everything profiles/rNN quotes for the final tree, in one gpurun call from the repo root:
#   gpurun -- 'MSS_TREE=<commit> tools/collect_round.sh <dir under gpurun_out>'   (the box has no .git: MSS_TREE labels the traffic profile)
#  * round/            : tools/profile_round.sh on the NATIVE route (kernel statistics of the step, FETCH_SIZE / WRITE_SIZE passes)
#  * stats_split/      : rocprofv3 --kernel-trace --stats of the same step on the split-bf16 route (MSS_GEMM_SPLIT=1)
#  * pmc_split_nt|tn/  : SQ counters of gemm_nt_bf16x3_kernel / gemm_tn_bf16x3_kernel (separate --pmc passes)
#  * layer tables of both routes, the decoder's, bench lines, kernel A/B tables, the GPU test log
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
bash tools/profile_round.sh $1/round > $O/profile_round.log 2>&1
(cd /tmp && export TMPDIR=/tmp && MSS_GEMM_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_split -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ood --no-split --no-m2f --no-parity > $O/bench_split_under_rocprof.json 2> /dev/null)
find $O/stats_split -name "*kernel_trace.csv" -delete
bash tools/pmc_split.sh $1/pmc_split_nt > $O/pmc_split_nt.log 2>&1
PMC_SPLIT_TN=1 bash tools/pmc_split.sh $1/pmc_split_tn 64,2304,4096,256 1,162624,1024,256 64,29412,256,256 > $O/pmc_split_tn.log 2>&1
python tools/layer_table.py 2>&1 | grep -v amdgpu.ids > $O/layer_table_native.txt
MSS_GEMM_SPLIT=1 python tools/layer_table.py 2>&1 | grep -v amdgpu.ids > $O/layer_table_bf16x3.txt
python tools/bench_gemm_split.py 2>&1 | grep -v amdgpu.ids > $O/bench_gemm_split.jsonl
python tools/bench_gemm_split.py --affine 2>&1 | grep -v amdgpu.ids > $O/bench_gemm_split_affine.jsonl
python tools/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids > $O/bench_wgrad_split.jsonl
python tools/prof_decoder.py 16 704 704 2 layers 2>&1 | grep -v amdgpu.ids > $O/decoder_layer_table_native.txt
MSS_GEMM_SPLIT=1 python tools/prof_decoder.py 16 704 704 2 layers 2>&1 | grep -v amdgpu.ids > $O/decoder_layer_table_bf16x3.txt
python -c "
import sys, json; sys.path.insert(0, 'tools')
import peaks; print(json.dumps({k: round(v, 1) for k, v in peaks.measure().items()}))" 2>/dev/null > $O/measured_peaks.json
python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
python bench.py --workload c2_700 --no-cpu-baseline --no-m2f --no-ood > $O/bench_c2_16x700.json 2> /dev/null
ls -la $O
