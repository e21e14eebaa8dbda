"""Summarise two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md HBM section)
over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ood` into profiles/conv_traffic_latest.json.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]

Per kernel family: launches, average FETCH_SIZE x2 (gfx950 counts a 128-B read request as 64 B for 16-B/lane
streaming reads) and WRITE_SIZE, in KiB as the counters report them; `hbm_bytes_per_launch` for conv_igemm is what
bench.py prints as roofline.traffic."""
import csv, glob, json, os, sys
from collections import defaultdict

FAMILIES = ["gemm_nt_kernel", "conv_igemm_kernel", "conv_wgrad_kernel", "gemm_tn_direct_kernel", "gemm_tn_narrow_kernel", "gemm_tn_wgrad_kernel",
            "wino_input_transform_aspp", "wino_input_transform_lds", "wino_input_transform_kernel",
            "wino_output_transform", "wino_grad_output_transform", "bn_stats_kernel", "bn_relu_bwd_reduce", "bn_relu_bwd_apply",
            "ood_score_v4", "ood_score_bwd_tiled", "rcl_pass1_v4", "rcl_pass2_v4", "upsample_ac_kernel", "upsample_ac_bwd_fast",
            "maxpool3s2", "colsum_kernel", "im2col3x3_c3", "m2f_score_kernel", "adam_kernel"]


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                fam = next((k for k in FAMILIES if k in row["Kernel_Name"]), None)
                if fam:
                    acc[fam][0] += 1
                    acc[fam][1] += float(row["Counter_Value"])
    return acc


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                    "profiles", "conv_traffic_latest.json")
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    fams = {}
    for k in FAMILIES:
        if fe[k][0] and wr[k][0]:
            f_kb, w_kb = fe[k][1] / fe[k][0], wr[k][1] / wr[k][0]
            fams[k] = {"launches": fe[k][0], "fetch_size_avg_KB": f_kb, "write_size_avg_KB": w_kb,
                       "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
    cands = [k for k in ("gemm_nt_kernel", "conv_igemm_kernel") if k in fams] or list(fams)
    dom = max(cands, key=lambda k: fams[k]["launches"] * fams[k]["hbm_bytes_per_launch"])
    c = fams[dom]
    out = {"kernel": dom, "git_commit": os.environ.get("MSS_TREE", "unknown"),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-ood`, summarised by tools/pmc_traffic.py",
           "launches": c["launches"], "fetch_size_avg_KB": c["fetch_size_avg_KB"], "write_size_avg_KB": c["write_size_avg_KB"],
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B for 16-B/lane streaming reads, "
                         "MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; units KiB",
           "hbm_bytes_per_launch": c["hbm_bytes_per_launch"], "families": fams}
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: {"launches": v["launches"], "MB_per_launch": round(v["hbm_bytes_per_launch"] / 1e6, 1)} for k, v in fams.items()}))


if __name__ == "__main__":
    main()
