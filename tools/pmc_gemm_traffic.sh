#!/bin/bash
# Per-launch HBM-side traffic of gemm_nt_kernel in one training step (tools/pmc_gemm_traffic.py): two separate --pmc passes over
# tools/layer_table.py.   usage (through gpurun, repo root): tools/pmc_gemm_traffic.sh <out_dir under gpurun_out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 "$GRAFT_REPO_ROOT/tools/layer_table.py" > "$OUT/table.txt" 2> /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 "$GRAFT_REPO_ROOT/tools/layer_table.py" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_gemm_traffic.py "$OUT" > "$OUT/gemm_traffic_by_launch.txt" 2>&1
find "$OUT" -name "*counter_collection.csv" -delete
tail -3 "$OUT/gemm_traffic_by_launch.txt"
