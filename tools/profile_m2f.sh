#!/bin/bash
# rocprofv3 artefacts for the Mask2Former-side kernels (MSDA op / encoder / pixel decoder / M2F score / OOD metrics):
# kernel statistics + FETCH_SIZE / WRITE_SIZE of the MSDA forward in separate --pmc passes.   usage: tools/profile_m2f.sh <out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/msda_stats" -- python3 "$R/tools/bench_msda.py" > "$OUT/bench_msda_under_rocprof.jsonl" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/msda_pmc_fetch" -- python3 "$R/tools/bench_msda.py" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/msda_pmc_write" -- python3 "$R/tools/bench_msda.py" > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/encoder_stats" -- python3 "$R/tools/bench_encoder.py" 16 > "$OUT/bench_encoder16_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/metric_stats" -- python3 "$R/tools/bench_metric.py" > "$OUT/bench_metric_under_rocprof.jsonl" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/m2f_stats" -- python3 "$R/tools/bench_m2f.py" > "$OUT/bench_m2f_under_rocprof.jsonl" 2> /dev/null
cd "$R"
python3 tools/bench_msda.py > "$OUT/bench_msda.jsonl" 2> /dev/null
python3 tools/bench_encoder.py 1 > "$OUT/bench_encoder.jsonl" 2> /dev/null
python3 tools/bench_encoder.py 16 >> "$OUT/bench_encoder.jsonl" 2> /dev/null
python3 tools/bench_metric.py > "$OUT/bench_metric.jsonl" 2> /dev/null
python3 tools/bench_m2f.py > "$OUT/bench_m2f.jsonl" 2> /dev/null
python3 tools/bench_decoder.py > "$OUT/bench_decoder.jsonl" 2> /dev/null
python3 - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
out = sys.argv[1]
res = {}
for counter, d in (("FETCH_SIZE", "msda_pmc_fetch"), ("WRITE_SIZE", "msda_pmc_write")):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"msda_\w+", row["Kernel_Name"])
            if not m:
                continue
            k = m.group(0)
            e = res.setdefault(k, {}).setdefault(counter, [])
            e.append(float(row["Counter_Value"]))
summ = {k: {c: {"launches": len(v), "min_KB": min(v), "max_KB": max(v)} for c, v in d.items()} for k, d in res.items()}
json.dump({"note": "per-dispatch FETCH_SIZE / WRITE_SIZE (KiB) of the MSDA kernels over tools/bench_msda.py: the three shapes (C4 N=1, C4 N=16, C5 N=1) "
                   "give the min .. max; FETCH_SIZE x2 is the gfx950 correction for 16-B/lane streaming reads (MI355X_MICROARCH.md), uncalibrated for 128-B row gathers",
           "kernels": summ}, open(os.path.join(out, "msda_traffic.json"), "w"), indent=1)
print(json.dumps(summ)[:1500])
PY
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
du -sh "$OUT"
