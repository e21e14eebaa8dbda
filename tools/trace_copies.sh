#!/bin/bash
# Where do the small device-to-device copies of a training step come from? Kernel trace of 2 steps; prints, for every
# __amd_rocclr_copyBuffer / fillBuffer dispatch of the LAST step, the kernels launched just before and after it.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_copies -- python3 $R/bench.py --no-m2f --no-parity --no-cpu-baseline --no-ood --no-split --steps 1 --warmup 1 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/trace_copies/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:70] for r in rows]
n = len(names)
idx = [i for i, k in enumerate(names) if "copyBuffer" in k or "fillBuffer" in k]
print(len(rows), "dispatches,", len(idx), "copy/fill")
half = [i for i in idx if i > n * 0.55]
for i in half:
    dur = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    print(f"{names[i-1][:48]:50s} -> {names[i][:28]:30s} {dur:6.1f} us -> {names[i+1][:48] if i + 1 < n else ''}")
PY
find gpurun_out/trace_copies -name "*.csv" -delete
