"""Per-launch HBM-side traffic of gemm_nt_kernel in one training step, joined with the launch's shape: which products re-read.

    (gpurun)  cd /tmp; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/layer_table.py > $OUT/table.txt
              rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/layer_table.py > /dev/null
              python tools/pmc_gemm_traffic.py $OUT

tools/layer_table.py runs two steps (warm-up + measured) and prints the measured one's launches in order; the k-th gemm_nt row of
the table is the k-th gemm_nt dispatch of the second step. Traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction for
16-byte-per-lane streaming reads, MI355X_MICROARCH.md HBM section), algorithmic = 4 x (rows x (C + K) + weights)."""
import ast, csv, glob, os, re, sys

out = sys.argv[1]


def per_dispatch(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "gemm_nt_kernel" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"].split("(")[0][-28:]))
    rows.sort()
    return rows


fe, wr = per_dispatch(os.path.join(out, "f"), "FETCH_SIZE"), per_dispatch(os.path.join(out, "w"), "WRITE_SIZE")
shapes = []
for ln in open(os.path.join(out, "table.txt")):
    m = re.match(r"gemm_nt\s+(\(.*?\))\s+([\d.]+)\s+([\d.]+)", ln)
    if m:
        shapes.append((ast.literal_eval(m.group(1)), float(m.group(2)), float(m.group(3))))
n = len(shapes)
assert len(fe) >= n and len(wr) >= n and len(fe) == len(wr), (len(fe), len(wr), n)
fe, wr = fe[-n:], wr[-n:]
print(f"{'P/N,H,W,C,K':40s} {'ms':>7s} {'TF/s':>6s} {'alg MB':>8s} {'read MB':>8s} {'write MB':>8s} {'traffic/alg':>11s} {'reads/alg reads':>15s}  kernel")
tot_a = tot_t = 0.0
for (shp, ms, tf), (_, f, name), (_, w, _) in zip(shapes, fe, wr):
    P, H, Wd, C, K = shp[0], shp[1], shp[2], shp[3], shp[4]
    rows = P * H * Wd if (H == 1 or P > 2) and shp[5] == 1 and False else P * H * Wd
    batch = P if H == 1 else 1
    alg_r = 4.0 * (rows * C + batch * K * C)
    alg_w = 4.0 * rows * K
    rd, wt = 2 * f * 1024, w * 1024
    tot_a += alg_r + alg_w
    tot_t += rd + wt
    print(f"{str(shp[:5]):40s} {ms:7.3f} {tf:6.1f} {(alg_r + alg_w) / 1e6:8.1f} {rd / 1e6:8.1f} {wt / 1e6:8.1f} {(rd + wt) / (alg_r + alg_w):11.2f} {rd / alg_r:15.2f}  {name}")
print(f"\nstep: {n} launches, traffic {tot_t / 1e9:.2f} GB, algorithmic {tot_a / 1e9:.2f} GB, ratio {tot_t / tot_a:.3f}")
