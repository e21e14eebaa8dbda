#!/bin/bash
# SQ counters of every MFMA kernel of the training step in ONE --pmc pass over `bench.py --steps 2` (never combined with tracing),
# and FETCH_SIZE / WRITE_SIZE in their own passes; summarised per kernel family by tools/pmc_step_mfma.py.
#   usage (through gpurun, from the repo root): tools/pmc_step_mfma.sh <out_dir under gpurun_out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-ood --no-split --no-m2f --no-parity"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 "$B" $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$B" $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$B" $ARGS > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/traffic.json" > /dev/null
python3 tools/pmc_step_mfma.py "$OUT" > "$OUT/pmc_step_mfma.md"
find "$OUT" -name "*counter_collection.csv" -delete
cat "$OUT/pmc_step_mfma.md"
