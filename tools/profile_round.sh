#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): kernel-trace statistics of the benchmark step,
# FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md: they do not fit one pass; never combined with
# tracing domains), the same two counter passes over the HBM-bound microbenchmarks, and the un-profiled numbers.
#   usage: tools/profile_round.sh <out_dir under gpurun_out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ood --no-split --no-m2f --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$B" $BARGS > "$OUT/bench_under_rocprof.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-ood --no-split --no-m2f --no-parity > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-ood --no-split --no-m2f --no-parity > /dev/null 2>&1
export MSS_BENCH_SKIP_WINO=1
M="$GRAFT_REPO_ROOT/tools/microbench_hbm.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hbm_stats" -- python3 "$M" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/hbm_pmc_fetch" -- python3 "$M" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/hbm_pmc_write" -- python3 "$M" > /dev/null 2>&1
unset MSS_BENCH_SKIP_WINO
cd "$GRAFT_REPO_ROOT"
python3 tools/microbench_hbm.py > "$OUT/microbench_hbm.jsonl" 2> /dev/null
python3 tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/conv_traffic.json"
python3 tools/pmc_traffic.py "$OUT/hbm_pmc_fetch" "$OUT/hbm_pmc_write" "$OUT/hbm_traffic.json" || true
# keep the summaries, drop the per-dispatch traces (tens of MB)
find "$OUT" -name "*kernel_trace.csv" -delete
for d in pmc_fetch pmc_write hbm_pmc_fetch hbm_pmc_write; do find "$OUT/$d" -name "*counter_collection.csv" -delete; done
du -sh "$OUT"
