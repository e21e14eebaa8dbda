#!/bin/bash
# SQ counters of gemm_nt_kernel (the dominant kernel) on two of the step's batched products, the variant the tree ships.
# Separate --pmc passes (8 SQ slots per pass, never combined with tracing); summarised by tools/pmc_gemm_summary.py.
#   usage (through gpurun, from the repo root): tools/pmc_gemm.sh <out_dir under gpurun_out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T="$GRAFT_REPO_ROOT/tools/bench_bgemm.py"
CASES="64,1936,512,1024 64,7396,256,256 64,2304,4096,256"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/p1" -- python3 "$T" $CASES > "$OUT/bench_p1.jsonl" 2> /dev/null
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d "$OUT/p2" -- python3 "$T" $CASES > "$OUT/bench_p2.jsonl" 2> /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p3" -- python3 "$T" $CASES > "$OUT/bench_p3.jsonl" 2> /dev/null
cd "$GRAFT_REPO_ROOT"
python3 tools/bench_bgemm.py $CASES > "$OUT/bench_unprofiled.jsonl" 2> /dev/null
python3 tools/pmc_gemm_summary.py "$OUT" > "$OUT/pmc_gemm.md"
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
cat "$OUT/pmc_gemm.md"
