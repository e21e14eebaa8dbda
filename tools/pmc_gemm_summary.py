"""Summarise the --pmc passes of tools/pmc_gemm.sh: per gemm_nt_kernel launch shape (grouped by grid size), the mean of every
counter over its launches, and the derived shares (matrix-pipe busy share of SIMD cycles from the MFMA count)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))        # (kernel, grid) -> counter -> values
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if "gemm_nt_kernel" not in name:
                continue
            key = (name.split("(")[0][-60:], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
bench = []
p = os.path.join(out, "bench_unprofiled.jsonl")
if os.path.exists(p):
    bench = [json.loads(l) for l in open(p) if l.strip().startswith("{")]
print("# gemm_nt_kernel -- SQ counters, rocprofv3 --pmc, three separate passes (tools/pmc_gemm.sh); means per launch, in millions\n")
print("un-profiled timings of the same products:", json.dumps(bench), "\n")
for key, ctr in sorted(rows.items(), key=lambda kv: kv[0][1]):
    print(f"## {key[0]} grid {key[1]} x {key[2]}\n")
    print("| counter | mean per launch (M) | launches |")
    print("|---|---|---|")
    for c in sorted(ctr):
        v = ctr[c]
        print(f"| {c} | {sum(v) / len(v) / 1e6:.3f} | {len(v)} |")
    m = lambda c: (sum(ctr[c]) / len(ctr[c])) if c in ctr else None
    wc, wa, wi, ai = m("SQ_WAVE_CYCLES"), m("SQ_WAIT_ANY"), m("SQ_WAIT_INST_ANY"), m("SQ_ACTIVE_INST_ANY")
    if wc:
        print(f"\nwave cycles: parked (s_waitcnt / barrier) {100 * wa / wc:.1f} %, issue-stalled {100 * wi / wc:.1f} %, issuing {100 * ai / wc:.1f} %")
    mf, busy = m("SQ_INSTS_MFMA"), m("SQ_BUSY_CYCLES")
    if mf and m("SQ_INSTS_VALU"):
        print(f"instructions per MFMA: VALU {m('SQ_INSTS_VALU') / mf:.2f}, LDS {m('SQ_INSTS_LDS') / mf:.2f}, SALU {m('SQ_INSTS_SALU') / mf:.2f}, "
              f"VMEM rd {m('SQ_INSTS_VMEM_RD') / mf:.3f}, wr {m('SQ_INSTS_VMEM_WR') / mf:.3f}")
    print()
