"""Summarise the --pmc passes of tools/pmc_split.sh: per product (the launches of gemm_nt_bf16x3_kernel in dispatch order, 3 warm-up +
5 measured per product given to tools/pmc_split.py), the mean of every counter over its launches and the derived shares: matrix-pipe
busy share of the SIMD cycles, effective clock, instructions per MFMA. Also writes <out>/split_pmc.json (the headline product's
figures; copied to profiles/split_pmc_latest.json, which bench.py attaches to the co-headline's roofline object)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = "gemm_tn_bf16x3_kernel" if os.environ.get("PMC_SPLIT_TN") == "1" else "gemm_nt_bf16x3_kernel"
PER_CASE = 8          # launches of the kernel per product: 3 warm-up + 5 timed (tools/pmc_split.py default)


def load(name):
    p = os.path.join(out, name)
    return [json.loads(l) for l in open(p) if l.strip().startswith("{")] if os.path.exists(p) else []


bench, native = load("bench_unprofiled.jsonl"), load("bench_unprofiled_native.jsonl")
cases = defaultdict(lambda: defaultdict(list))       # case index -> counter -> values
meta = {}
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(dict)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if KERNEL not in r.get("Kernel_Name", ""):
                continue
            d = int(r["Dispatch_Id"])
            per_dispatch[d][r["Counter_Name"]] = float(r["Counter_Value"])
            per_dispatch[d]["_meta"] = (r["Kernel_Name"].split("(")[0][-48:], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    for i, d in enumerate(sorted(per_dispatch)):
        ci = i // PER_CASE
        if i % PER_CASE < 3:
            continue                                    # warm-up launches
        meta[ci] = per_dispatch[d]["_meta"]
        for c, v in per_dispatch[d].items():
            if c != "_meta":
                cases[ci][c].append(v)
print(f"# {KERNEL} -- SQ counters, rocprofv3 --pmc, three separate passes (tools/pmc_split.sh); means per launch, in millions\n")
print("un-profiled timings of the same products, split route:", json.dumps(bench), "\n")
print("native fp32 MFMA route:", json.dumps(native), "\n")
summary = []
for ci in sorted(cases):
    ctr = cases[ci]
    b = bench[ci] if ci < len(bench) else None
    nat = native[ci] if ci < len(native) else None
    km = meta[ci]
    title = f"{b['P']} x {b['T']} x {b['C']} -> {b['K']}" if b else f"case {ci}"
    print(f"## {title}   ({km[0]}, grid {km[1]} x {km[2]}, {km[3]} VGPR + {km[4]} AGPR, LDS {km[5]} B, scratch {km[6]} B)\n")
    print("| counter | mean per launch (M) | launches |")
    print("|---|---|---|")
    for c in sorted(ctr):
        v = ctr[c]
        print(f"| {c} | {sum(v) / len(v) / 1e6:.3f} | {len(v)} |")
    m = lambda c: (sum(ctr[c]) / len(ctr[c])) if c in ctr else None
    wc, wa, wi, ai = m("SQ_WAVE_CYCLES"), m("SQ_WAIT_ANY"), m("SQ_WAIT_INST_ANY"), m("SQ_ACTIVE_INST_ANY")
    rec = {"product": title}
    if wc:
        print(f"\nwave cycles: parked (s_waitcnt / barrier) {100 * wa / wc:.1f} %, issue-stalled {100 * wi / wc:.1f} %, issuing {100 * ai / wc:.1f} %")
        rec.update(wave_parked_pct=round(100 * wa / wc, 1), wave_issue_stalled_pct=round(100 * wi / wc, 1))
    mf = m("SQ_INSTS_MFMA")
    if mf and m("GRBM_GUI_ACTIVE") and m("SQ_VALU_MFMA_BUSY_CYCLES"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; one v_mfma_f32_32x32x16_bf16 holds its SIMD's matrix pipe for 32 cycles and
        # SQ_VALU_MFMA_BUSY_CYCLES counts exactly that; 256 CUs x 4 SIMDs. fp32-equivalent FLOPs: 32768 per MFMA / 6 products
        cyc = m("GRBM_GUI_ACTIVE") / 8.0
        share = m("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0)
        line = (f"\n**matrix pipe busy {100 * share:.1f} % of the launch's {cyc / 1e6:.3f} M shader cycles** (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)); "
                f"busy cycles per MFMA {m('SQ_VALU_MFMA_BUSY_CYCLES') / mf:.1f}")
        rec.update(mfma_pipe_busy_pct=round(100 * share, 1))
        if b:
            clk = cyc / (b["ms"] * 1e-3) / 1e9
            peak = 2500.0 / 6.0
            line += (f"; {b['ms']} ms un-profiled = {b['tflops']} TFLOP/s fp32-equivalent = {b['tflops'] / peak:.3f} of bf16 dense peak / 6 = {peak:.0f} (2.4 GHz); "
                     f"cycles / time = {clk:.2f} GHz effective clock, i.e. {b['tflops'] / (peak * clk / 2.4):.3f} of the peak AT THAT CLOCK")
            rec.update(ms=b["ms"], tflops_fp32_equivalent=b["tflops"], frac_of_bf16_peak_over_6=round(b["tflops"] / peak, 4), effective_clock_GHz=round(clk, 2),
                       frac_of_peak_at_that_clock=round(b["tflops"] / (peak * clk / 2.4), 3))
            if nat:
                line += f"; native fp32 MFMA kernel on the same operands: {nat['ms']} ms = {nat['tflops']} TFLOP/s (x{nat['ms'] / b['ms']:.2f})"
                rec.update(native_ms=nat["ms"], native_tflops=nat["tflops"])
        print(line)
    if mf and m("SQ_INSTS_VALU"):
        print(f"instructions per MFMA: VALU (other than MFMA) {m('SQ_INSTS_VALU') / mf - 1:.2f}, LDS {m('SQ_INSTS_LDS') / mf:.2f}, SALU {m('SQ_INSTS_SALU') / mf:.2f}, "
              f"VMEM rd {m('SQ_INSTS_VMEM_RD') / mf:.3f}, wr {m('SQ_INSTS_VMEM_WR') / mf:.3f}; LDS bank-conflict cycles {m('SQ_LDS_BANK_CONFLICT') or 0:.0f}")
        rec.update(valu_per_mfma=round(m("SQ_INSTS_VALU") / mf - 1, 2), lds_per_mfma=round(m("SQ_INSTS_LDS") / mf, 2))
    summary.append(rec)
    print()
json.dump({"source": "rocprofv3 --pmc, three separate passes, tools/pmc_split.sh", "products": summary}, open(os.path.join(out, "split_pmc.json"), "w"), indent=1)
