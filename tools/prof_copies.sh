#!/bin/bash
# rocprofv3 memory-copy trace of a few training steps: which small copies does a step issue? (run through gpurun from the repo root)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --memory-copy-trace --output-format csv -d $R/gpurun_out/copies -- python3 $R/bench.py --no-m2f --no-parity --no-cpu-baseline --no-ood --no-split --steps 3 --warmup 1 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/copies/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else "")
    c = collections.Counter()
    for r in rows:
        c[(r.get("Direction"), r.get("Size") or r.get("Bytes"))] += 1
    for k, v in c.most_common(30):
        print(k, v)
PY
