"""Summarise the --pmc passes of tools/pmc_split.sh: per gemm_nt_bf16x3_kernel launch shape (grouped by grid size), the mean of every
counter over its launches, and the derived shares: matrix-pipe busy share of the SIMD cycles, effective clock, instructions per MFMA."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))        # (kernel, grid) -> counter -> values
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if "gemm_nt_bf16x3_kernel" not in name:
                continue
            key = (name.split("(")[0][-70:], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))


def load(name):
    p = os.path.join(out, name)
    return [json.loads(l) for l in open(p) if l.strip().startswith("{")] if os.path.exists(p) else []


bench, native = load("bench_unprofiled.jsonl"), load("bench_unprofiled_native.jsonl")
print("# gemm_nt_bf16x3_kernel -- SQ counters, rocprofv3 --pmc, three separate passes (tools/pmc_split.sh); means per launch, in millions\n")
print("un-profiled timings of the same products, split route:", json.dumps(bench), "\n")
print("native fp32 MFMA route:", json.dumps(native), "\n")
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
    mf = m("SQ_INSTS_MFMA")
    if mf and m("GRBM_GUI_ACTIVE") and m("SQ_VALU_MFMA_BUSY_CYCLES"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; one v_mfma_f32_32x32x16_bf16 holds its SIMD's matrix pipe for 32 cycles and
        # SQ_VALU_MFMA_BUSY_CYCLES counts exactly that; 256 CUs x 4 SIMDs. fp32-equivalent FLOPs: 32768 per MFMA / 6 products
        cyc = m("GRBM_GUI_ACTIVE") / 8.0
        share = m("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0)
        flops = mf * 32768.0 / 6.0
        match = [b for b in bench if abs(2.0 * b["P"] * b["T"] * b["C"] * b["K"] / flops - 1) < 0.07]
        line = f"\n**matrix pipe busy {100 * share:.1f} % of the launch's {cyc / 1e6:.3f} M shader cycles** (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)); busy cycles per MFMA {m('SQ_VALU_MFMA_BUSY_CYCLES') / mf:.1f}"
        if match:
            b = match[0]
            clk = cyc / (b["ms"] * 1e-3) / 1e9
            peak = 2500.0 / 6.0
            line += (f"; product {b['P']} x {b['T']} x {b['C']} -> {b['K']}: {b['ms']} ms un-profiled = {b['tflops']} TFLOP/s fp32-equivalent = "
                     f"{b['tflops'] / peak:.3f} of bf16 dense peak / 6 = {peak:.0f} (2.4 GHz); cycles / time = {clk:.2f} GHz effective clock, i.e. "
                     f"{b['tflops'] / (peak * clk / 2.4):.3f} of the peak AT THAT CLOCK")
        print(line)
    if mf and m("SQ_INSTS_VALU"):
        print(f"instructions per MFMA: VALU (incl. MFMA) {m('SQ_INSTS_VALU') / mf:.2f}, LDS {m('SQ_INSTS_LDS') / mf:.2f}, SALU {m('SQ_INSTS_SALU') / mf:.2f}, "
              f"VMEM rd {m('SQ_INSTS_VMEM_RD') / mf:.3f}, wr {m('SQ_INSTS_VMEM_WR') / mf:.3f}")
    print()
