#!/bin/bash
# SQ counters of gemm_nt_bf16x3_kernel (the split-bf16 GEMM route) on products of the step. Separate --pmc passes, never combined
# with tracing; summarised by tools/pmc_split_summary.py.   usage (through gpurun): tools/pmc_split.sh <out_dir under gpurun_out> [cases]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T="$GRAFT_REPO_ROOT/tools/pmc_split.py"
CASES="${*:-1,65536,2048,4096 64,2112,1024,2048 36,16384,256,256}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/p1" -- python3 "$T" $CASES > "$OUT/bench_p1.jsonl" 2> /dev/null
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d "$OUT/p2" -- python3 "$T" $CASES > "$OUT/bench_p2.jsonl" 2> /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p3" -- python3 "$T" $CASES > "$OUT/bench_p3.jsonl" 2> /dev/null
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_split.py --launches 20 $CASES > "$OUT/bench_unprofiled.jsonl" 2> /dev/null
python3 tools/pmc_split.py --native --launches 20 $CASES > "$OUT/bench_unprofiled_native.jsonl" 2> /dev/null
python3 tools/pmc_split_summary.py "$OUT" > "$OUT/pmc_split.md"
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
cat "$OUT/pmc_split.md"
