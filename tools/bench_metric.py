"""OOD-metric sweep on the device (8f-1): Mpix/s of update (count + compact) and compute (2 sorts + rank pass) at
BASELINE sizes, with the reference-style CPU path (numpy oracle = sklearn's algorithm) timed on a bounded sample."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import metric as M
from oracle import metric as ometric


def run(images, h=1024, w=2048):
    g = torch.Generator(device="cuda").manual_seed(images)
    batches = []
    for _ in range(images):
        lab = (torch.rand(1, h, w, device="cuda", generator=g) < 0.03).long()
        lab[torch.rand(1, h, w, device="cuda", generator=g) < 0.05] = 255
        batches.append((torch.randn(1, h, w, device="cuda", generator=g) + 1.2 * (lab == 1), lab))
    for rep in range(2):                      # first repetition warms the allocator / kernels
        meter = M.OODMeter()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s, l in batches:
            meter.update(s, l)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res = meter.compute()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    px = images * h * w
    return dict(images=images, pixels=px, update_ms=round(1e3 * (t1 - t0), 2), compute_ms=round(1e3 * (t2 - t1), 2),
                total_mpix_s=round(px / (t2 - t0) / 1e6, 1), update_GBs=round(px * 12 / (t1 - t0) / 1e9, 1), measures=res)


if __name__ == "__main__":
    for images in (1, 8, 64):
        print(json.dumps(run(images)), flush=True)
    # CPU baseline on one 1024x2048 map (the reference would also pay the D2H copies)
    rng = np.random.default_rng(0)
    lab = (rng.random((1, 1024, 2048)) < 0.03).astype(np.int64)
    sc = (rng.standard_normal((1, 1024, 2048)) + 1.2 * lab).astype(np.float32)
    t0 = time.perf_counter(); r = ometric.eval_ood_measure(sc, lab); t = time.perf_counter() - t0
    print(json.dumps(dict(cpu_oracle_pixels=sc.size, seconds=round(t, 3), mpix_s=round(sc.size / t / 1e6, 2), measures=r)))
