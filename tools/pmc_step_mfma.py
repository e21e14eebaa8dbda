"""Summarise tools/pmc_step_mfma.sh: per MFMA kernel family of the training step, launches, matrix-pipe busy share of the launch's
shader cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)), instructions per MFMA, and memory-side traffic per
launch (FETCH_SIZE x2 + WRITE_SIZE, separate passes, from traffic.json)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
FAMS = ["gemm_nt_kernel", "gemm_tn_direct_kernel", "gemm_tn_narrow_kernel", "gemm_tn_wgrad_kernel", "conv_igemm_kernel", "conv_wgrad_kernel",
        "stem_conv_pool_kernel"]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            fam = next((k for k in FAMS if k in r["Kernel_Name"]), None)
            if fam:
                acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[fam][r["Counter_Name"]] += 1
traffic = {}
tp = os.path.join(out, "traffic.json")
if os.path.exists(tp):
    traffic = json.load(open(tp)).get("families", {})
print("# MFMA kernels of the 2 x 1024 x 2048 training step -- rocprofv3 --pmc over `bench.py --steps 2 --warmup 1` (SQ counters in one pass,\n"
      "# FETCH_SIZE and WRITE_SIZE in their own passes); sums over all launches of a family\n")
print("| kernel family | launches | matrix pipe busy, % of shader cycles | VALU (non-MFMA) / MFMA | LDS / MFMA | VMEM rd / MFMA | MB per launch, memory side |")
print("|---|---|---|---|---|---|---|")
for fam in FAMS:
    a = acc.get(fam)
    if not a or not a.get("SQ_INSTS_MFMA"):
        continue
    n = cnt[fam]["SQ_INSTS_MFMA"]
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    share = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    mf = a["SQ_INSTS_MFMA"]
    t = traffic.get(fam, {}).get("hbm_bytes_per_launch")
    print(f"| `{fam}` | {n} | {100 * share:.1f} | {(a['SQ_INSTS_VALU'] - mf) / mf:.2f} | {a['SQ_INSTS_LDS'] / mf:.2f} | {a['SQ_INSTS_VMEM_RD'] / mf:.3f} | "
          f"{t / 1e6:.0f}" + " |" if t else f"| `{fam}` | {n} | {100 * share:.1f} | {(a['SQ_INSTS_VALU'] - mf) / mf:.2f} | {a['SQ_INSTS_LDS'] / mf:.2f} | "
          f"{a['SQ_INSTS_VMEM_RD'] / mf:.3f} | - |")
print("\n(SQ_INSTS_VALU counts the MFMA instructions too; the column subtracts them. GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
      "Profiled launches run slower than un-profiled ones; the shares are of the cycles they took.)")
