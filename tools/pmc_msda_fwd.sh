#!/bin/bash
# SQ counters of the two MSDeformAttn forward kernels at C4 N = 16 (one --pmc pass, never combined with tracing).
#   usage (through gpurun, from the repo root): tools/pmc_msda_fwd.sh <out_dir under gpurun_out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_msda_fwd.py" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$OUT" <<'PY' | tee "$OUT/pmc_msda_fwd.md"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        n = r["Kernel_Name"]
        if "msda_fwd" not in n:
            continue
        key = ("record kernel (r04)" if "msda_fwd_rec" in n else "round-2/3 kernel") + (", fused" if "true" in n.split("<")[1].split(">")[0].split(",")[1] else ", op")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# MSDeformAttn forward, C4 N = 16 (15.6 M samples): SQ counters per launch in millions, rocprofv3 --pmc (one pass)\n")
print("| kernel | launches | SQ_INSTS_VALU | SQ_INSTS_SALU | SQ_INSTS_LDS | SQ_INSTS_VMEM_RD | SQ_WAVES |")
print("|---|---|---|---|---|---|---|")
tot = {}
for k in sorted(acc):
    a = {c: sum(v) / len(v) for c, v in acc[k].items()}
    tot[k] = a
    print(f"| {k} | {len(acc[k]['SQ_INSTS_VALU'])} | {a['SQ_INSTS_VALU'] / 1e6:.2f} | {a['SQ_INSTS_SALU'] / 1e6:.2f} | {a['SQ_INSTS_LDS'] / 1e6:.2f} | "
          f"{a['SQ_INSTS_VMEM_RD'] / 1e6:.2f} | {a['SQ_WAVES'] / 1e6:.3f} |")
for form in ("op", "fused"):
    new, old = tot.get(f"record kernel (r04), {form}"), tot.get(f"round-2/3 kernel, {form}")
    if new and old:
        print(f"\n{form} form: VALU instructions {new['SQ_INSTS_VALU'] / old['SQ_INSTS_VALU']:.2f}x, SALU {new['SQ_INSTS_SALU'] / old['SQ_INSTS_SALU']:.2f}x, "
              f"buffer loads {new['SQ_INSTS_VMEM_RD'] / old['SQ_INSTS_VMEM_RD']:.2f}x of the round-2/3 kernel's")
PY
find "$OUT" -name "*counter_collection.csv" -delete
