#!/usr/bin/env python3
"""Headline benchmark: DeepLabV3+/WRN38 training step (+ OOD-score throughput) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch: trunk forward (train-mode BN +
Dropout2d on the frozen trunk), decoder/heads forward, fused RelContrastiveLoss, backward
(dgrad/wgrad/BN backward of the stage's trainable set), gradient all-reduce (N > 1), Adam step.
Default workload = BASELINE.json's metric configuration, per GPU: one (original, augmented) pair of
1024x2048 images (2 images), stage-2 trainable set (aspp, bot_fine, bot_aspp, ood_head). Every rank
has the same amount of work (weak scaling); `value` is whole-job images/s. Inputs are generated on
the device before the timed region; weights are synthetic (multishiftseg_amd.synth), fp32 throughout.

One JSON line on rank 0 carries the step metric, the fp32-MFMA roofline of the dominant kernel
(gemm_nt / conv_igemm: HIP events on the launch stream around every launch, in a SECOND pass of K steps right after the
timed region so that the ~110 event records per step are not charged to `value`), `parity` (the default route against the
reference's own outputs at this very configuration, argmax flips over all pixels), the OOD-score Mpix/s of the eval path
(eager, score-only, hipGraph), at N = 1 the Mask2Former legs (`m2f`: MSDeformAttn op, pixel decoder, fused score, metric
sweep) and a CPU baseline (stock-torch restatements on the host's cores, bounded samples), at N > 1 `comm`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md, dense fp32 matrix
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md, dense bf16 matrix (never the 2:1-sparsity figure)
FWD_GMAC_1024x2048 = 5827.2            # conv MACs of one forward at 1024x2048
STAGE2_STEP_OVER_FWD = 1.287           # stage-2 step FLOPs / forward FLOPs


def _host_info():
    import subprocess
    info = {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads()}
    try:
        for ln in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            key, _, val = ln.partition(":")
            if key.strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core"):
                info["lscpu_" + key.strip().lower().replace(" ", "_").replace("(s)", "s")] = val.strip()
    except Exception as exc:
        info["lscpu_error"] = str(exc)
    return info


def cpu_baseline(bench_hw):
    """SURVEY 8(d) "CPU baseline beside it", on this host's cores, in this process, before the GPU timings:
      * C1 exactly (1x3x512x1024 eval forward + per-pixel OOD score) with the graph composed from STOCK torch CPU ops
        (oracle/deepv3_torch.py: F.conv2d / F.batch_norm / F.interpolate(align_corners=True) / logsumexp -- a port, not
        the reference's files): 1 warm-up + 3 timed runs, median;
      * the numpy restatement (oracle/deepv3.py) on the same C1 image, once;
      * one stock-torch forward of a single image at the benchmark resolution, from which the train-step figure is
        derived by the stage-2 step/forward FLOP ratio (a 2-image CPU train step would take minutes)."""
    import statistics
    from multishiftseg_amd import synth
    from oracle import deepv3 as odeepv3, deepv3_torch
    params = synth.deepwv3plus_params(0)
    pt = deepv3_torch.to_torch(params)
    img = torch.from_numpy(synth.synth_image(3, 1, 512, 1024))
    runs = []
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            score, logit = deepv3_torch.forward_t(pt, img)
            dt = time.perf_counter() - t0
            if i:
                runs.append(dt)
        H, W = bench_hw
        big = torch.from_numpy(synth.synth_image(3, 1, H, W))
        t0 = time.perf_counter()
        deepv3_torch.forward_t(pt, big)
        t_big = time.perf_counter() - t0
    t0 = time.perf_counter()
    ns, nl = odeepv3.forward(params, img.numpy())
    t_np = time.perf_counter() - t0
    agree = float(np.abs(nl - logit.numpy()).max())
    return dict(c1_runs_s=[round(r, 3) for r in runs], c1_median_s=statistics.median(runs), c1_numpy_s=t_np,
                c1_numpy_vs_torch_max_abs_logit_diff=agree, bench_fwd_s=t_big, bench_hw=[H, W], host=_host_info())


def cpu_baseline_msda():
    """SURVEY 8(d) "MSDA: grid_sample-based restatement at C4": the forward of the op at N = 1, Lq = S = 10 164 (levels
    22^2 / 44^2 / 88^2, 8 heads x 32 channels, 4 points) composed from stock torch CPU ops the way the reference's own CPU path
    composes it (oracle/msda.py:forward_sampled <- ops/functions/ms_deform_attn_func.py:52-72); 1 warm-up + 3 runs, median."""
    import statistics
    from oracle import msda as omsda
    rng = np.random.default_rng(0)
    shapes = np.array([(22, 22), (44, 44), (88, 88)], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(shapes.prod(1))[:-1]]).astype(np.int64)
    S = int(shapes.prod(1).sum())
    value = rng.standard_normal((1, S, 8, 32), dtype=np.float32)
    loc = rng.random((1, S, 8, 3, 4, 2), dtype=np.float32)
    attn = rng.random((1, S, 8, 3, 4), dtype=np.float32)
    attn /= attn.sum((-1, -2), keepdims=True)
    runs = []
    for i in range(4):
        t0 = time.perf_counter()
        omsda.forward_sampled(value, shapes, starts, loc, attn)
        if i:
            runs.append(time.perf_counter() - t0)
    return {"forward_c4_n1_s": round(statistics.median(runs), 4), "runs_s": [round(r, 4) for r in runs], "tokens": S,
            "kind": "port", "threads": torch.get_num_threads(),
            "sample": "MSDeformAttn forward, N=1, 10 164 queries x 8 heads x 12 samples, stock torch grid_sample composition"}


def parity_check(model, H, W, pairs):
    """The default route of THIS process against the reference's own outputs at THIS configuration (fixture generated by
    importing the reference: tools/gen_golden.py train_c3 -> tests/golden/deepwv3plus_train_step_2x1024x2048.npz; data only):
    train-mode forward with the fixture's Dropout2d masks; max |error| on the stored strided logit / score slices and argmax
    flips against the reference's full label map over ALL pixels. Running statistics are restored afterwards."""
    path = os.path.join(ROOT, "tests", "golden", f"deepwv3plus_train_step_{2 * pairs}x{H}x{W}.npz")
    if not os.path.exists(path):
        return None
    from multishiftseg_amd import synth
    g = np.load(path)
    pre = "stage2_"
    dev = next(model.parameters()).device
    saved = {k: v.detach().clone() for k, v in model.state_dict().items()}
    was = model.training
    try:
        model.train()
        model.dropout_masks = {"mod6": torch.from_numpy(g[pre + "drop_mod6"]), "mod7": torch.from_numpy(g[pre + "drop_mod7"])}
        img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), 2 * pairs, H, W)).to(dev)
        with torch.no_grad():
            score, logit = model(img)
        ss, ls = int(g["score_stride"]), int(g["logit_stride"])
        flip = logit.argmax(1).to(torch.uint8) != torch.from_numpy(g[pre + "label"]).to(dev)
        clear = torch.from_numpy(np.unpackbits(g[pre + "clear_bits_1e3"])[:flip.numel()].reshape(tuple(flip.shape)).astype(bool)).to(dev)
        return {"against": "reference outputs (tests/golden/" + os.path.basename(path) + ", train-mode forward, reference Dropout2d masks)",
                "max_abs_logit_err": float((logit[:, :, ::ls, ::ls] - torch.from_numpy(g[pre + "logit_sub"]).to(dev)).abs().max()),
                "max_abs_score_err": float((score[:, ::ss, ::ss] - torch.from_numpy(g[pre + "score"]).to(dev)).abs().max()),
                "compared": f"every {ls}th logit / {ss}th score in each direction", "tolerance": 1e-3,
                "argmax_flips_all_pixels": int(flip.sum()), "pixels": int(flip.numel()),
                "argmax_flips_where_reference_margin_gt_1e-3": int((flip & clear).sum()),
                "wino_accuracy": os.environ.get("MSS_WINO_ACCURACY", "strict")}
    finally:
        model.dropout_masks = None
        model.load_state_dict(saved)
        model.train(was)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes through torch.distributed.run as a
    CHILD process (never exec: this image forbids replacing a process image once a GPU runtime is loaded, and the
    parent must not touch the GPU at all -- it has not: nothing before this point makes a HIP call), relay what
    rank 0's ONE JSON line on stdout (anything else the ranks or their libraries write to stdout -- e.g. gloo's connection
    banner -- goes to stderr, so that stdout stays one line) and exit with the launcher's status."""
    import subprocess
    # --standalone: the launcher's own c10d rendezvous on a port IT picks and binds (no bind-close-rebind race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    relayed = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
            relayed += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and relayed != 1:
        sys.stderr.write(f"bench.py: the ranks exited 0 but rank 0 printed {relayed} result lines\n")
        return 1
    if rc != 0:
        sys.stderr.write(f"bench.py: torch.distributed.run exited with status {rc}; no result line is valid\n")
    return rc


def launch_check(args):
    """--workload launchcheck: no model. Every rank pushes K steps of stage-2-sized synthetic gradients through
    ddp.GradAllReduce (the bucketed side-stream all-reduce the training step uses) and checks the averaged values.
    Exercises launcher + process group + collective path on whatever backend the box offers (nccl = RCCL on GPUs, gloo on
    CPU), so the RCCL branch can be proven without the 137 M-parameter model."""
    from multishiftseg_amd import ddp
    rank, world, local_rank, device = ddp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    sizes = [("g0", 4864), ("g1", 1 << 20), ("g2", 9 << 20), ("g3", 9 << 20)] if device.type == "cuda" else \
        [("g0", 4864), ("g1", 1 << 16), ("g2", 1 << 18)]
    named = [(n, torch.zeros(sz, device=device)) for n, sz in sizes]
    sync = ddp.GradAllReduce(named, bucket_bytes=(48 << 20) if device.type == "cuda" else (1 << 19))

    def one_step(k):
        grads = [(n, sync(n, torch.full((sz,), float(rank + 1 + k), device=device))) for n, sz in sizes]
        sync.backward_done()
        return [(n, g if g is not None else torch.full((sz,), float(rank + 1 + k), device=device)) for (n, g), (_, sz) in zip(grads, sizes)]

    def sync_dev():
        if device.type == "cuda":
            torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    if world > 1:
        dist.barrier()
    sync_dev()
    t0 = time.perf_counter()
    for k in range(args.steps):
        grads = one_step(k)
    sync_dev()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    want = sum(r + 1 + (args.steps - 1) for r in range(world)) / world
    ok = all(bool(torch.allclose(g, torch.full_like(g, want))) for _, g in grads)
    if world > 1:
        t = torch.tensor([elapsed, 0.0 if ok else 1.0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ok = float(t[0]), float(t[1]) == 0.0
    nbytes = 4 * sum(sz for _, sz in sizes)
    if rank == 0:
        print(json.dumps({"metric": "launch check: bucketed gradient all-reduce steps/s (no model)", "value": round(args.steps / elapsed, 3),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "launchcheck", "parallelism": f"dp{world}", "backend": dist.get_backend() if world > 1 else None,
                                     "rccl_ranks": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else 0),
                                     "gloo_ranks": (dist.get_world_size() if world > 1 and dist.get_backend() == "gloo" else 0),
                                     "device": device.type, "bytes_per_step": nbytes, "averages_correct": ok}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    os.environ.setdefault("MSS_LINEAR_STRICT", "1")      # a Linear outside the MFMA kernels' shapes must not reach the library GEMM unnoticed in a benchmark
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c2_700", "tiny", "launchcheck"],
                    help="c3: 1 pair of 1024x2048 per GPU; c2: 8 pairs of 768x768 (BASELINE's wording of exps/DeepLab.yaml batch 8); "
                         "c2_700: 8 pairs of 700x700 (the crop size exps/DeepLab.yaml actually uses); tiny: smoke; "
                         "launchcheck: launcher + process group + bucketed all-reduce only (runs on CPU/gloo too)")
    ap.add_argument("--stage", type=int, default=2, choices=[1, 2])
    ap.add_argument("--loss-sync", default="local", choices=["local", "global"],
                    help="local: per-rank loss (no loss collectives); global: reference semantics over all ranks' pairs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ood", action="store_true")
    ap.add_argument("--no-m2f", action="store_true", help="skip the Mask2Former legs (`m2f`, N=1 only, ~40 s)")
    ap.add_argument("--no-parity", action="store_true", help="skip the live check against the reference fixture (`parity`)")
    ap.add_argument("--no-split", action="store_true",
                    help="skip the co-headline run on the split-bf16 GEMM route (`value_fp32_via_bf16x3`, N=1 only; never part of `value`)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.workload == "launchcheck":
        return launch_check(args)

    from multishiftseg_amd import _lib, ddp, kernels as K, synth
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    from multishiftseg_amd.loss import RelContrastiveLoss
    from multishiftseg_amd.trainer import LOSS_PARAMS, TrainStep, ood_scores

    rank, world, local_rank, device = ddp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X")

    pairs, H, W = {"c3": (1, 1024, 2048), "c2": (8, 768, 768), "c2_700": (8, 700, 700), "tiny": (1, 128, 256)}[args.workload]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline((H, W))

    params = synth.deepwv3plus_params(0)
    model = DeepWV3Plus(19)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()})
    del params
    model = model.to(device)
    model.uncertainty_func_init()
    crit = RelContrastiveLoss(LOSS_PARAMS, pairing="device", seed=1000 + (rank if args.loss_sync == "local" else 0),
                              sync=args.loss_sync)
    step = TrainStep(model, crit, stage=args.stage)

    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    img = torch.randn((2 * pairs, 3, H, W), device=device, generator=gen)
    target0 = torch.from_numpy(synth.synth_targets(100 + rank, pairs, H, W)).to(device)

    def one_step():
        return step(img, target0.clone())      # the loss mutates its targets; a loader would hand over fresh ones

    parity = None
    if rank == 0 and not args.no_parity and args.workload == "c3":
        try:                        # before any optimizer step: the weights are still the ones the fixture was made with
            parity = parity_check(model, H, W, pairs)
        except Exception as exc:    # never lose the measurement to the checker
            parity = {"error": repr(exc)}

    parity_split = None
    if parity is not None and world == 1 and not args.no_split:
        K.set_gemm_route("bf16x3")
        try:
            parity_split = parity_check(model, H, W, pairs)
        except Exception as exc:
            parity_split = {"error": repr(exc)}
        finally:
            K.set_gemm_route(None)

    for _ in range(args.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    loss_val = float(loss.detach())
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # second, UNTIMED pass of the same K steps with a HIP event pair around every MFMA launch (the roofline leg)
    prof = K.ConvProfile()
    K.set_conv_profile(prof)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    elapsed_profiled = time.perf_counter() - t1
    K.set_conv_profile(None)

    comm = None
    if world > 1:
        # exposed communication = step(N) - the same step with the collectives switched off (everything else -- scaling into
        # the flat buffers, stream waits -- still runs): a fresh TrainStep under MSS_DDP_NO_COMM=1, same inputs
        os.environ["MSS_DDP_NO_COMM"] = "1"
        try:
            step_nc = TrainStep(model, crit, stage=args.stage)
            for _ in range(max(1, args.warmup)):
                step_nc(img, target0.clone())
            dist.barrier()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step_nc(img, target0.clone())
            torch.cuda.synchronize()
            dist.barrier()
            t_nc = torch.tensor([time.perf_counter() - t2], device=device, dtype=torch.float64)
            dist.all_reduce(t_nc, op=dist.ReduceOp.MAX)
            sync = step.sync
            comm = {"gradient_bytes_per_step": sync.bytes_per_step, "buckets": len(sync.buckets),
                    "bucket_bytes": [4 * n for n in sync.sizes], "backend": dist.get_backend(),
                    "ms_per_step_without_collectives": round(1e3 * float(t_nc) / args.steps, 3),
                    "exposed_ms_per_step": round(1e3 * (elapsed - float(t_nc)) / args.steps, 3),
                    "note": "exposed = ms_per_step - the same step with the all-reduces skipped (MSS_DDP_NO_COMM=1), max over ranks"}
        finally:
            os.environ.pop("MSS_DDP_NO_COMM", None)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    images = 2 * pairs * world * args.steps
    value = images / elapsed
    summ = prof.summary()
    # the two forward MFMA kernels (same 128x128x16 inner loop): gemm_nt_kernel takes the 1x1 / batched Winograd-domain
    # products, conv_igemm_kernel the implicit-GEMM shapes; the roofline object is the one with more time in the step
    kinds = {k: summ[k] for k in ("gemm_nt", "conv_igemm") if k in summ and summ[k]["launches"]}
    dom = max(kinds, key=lambda k: kinds[k]["ms"]) if kinds else "conv_igemm"
    conv = kinds.get(dom, dict(launches=0, flops=0.0, ms=1.0))
    achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["launches"] else 0.0
    wg = summ.get("conv_wgrad")
    # algorithmic HBM bytes of a launch: input once + output once + weights once (fp32)
    alg_bytes = 0.0
    for (kind, _f, _s, _e), tag in zip(prof.records, prof.tags):
        if kind == dom:
            n, h, w, c, k, r, st, _d = tag
            alg_bytes += 4.0 * (n * h * w * c + n * (-(-h // st)) * (-(-w // st)) * k + r * r * c * k * (n if h == 1 else 1))
    step_flops_alg = 2 * FWD_GMAC_1024x2048 * 1e9 * (H * W) / (1024 * 2048) * 2 * pairs * \
        (STAGE2_STEP_OVER_FWD if args.stage == 2 else 1.0)

    out = {
        "metric": "train images/sec (DeepLabV3+ WRN38, fwd + RelContrastiveLoss + bwd + Adam)",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {2 * pairs} images ({pairs} orig+aug pair(s)) of {H}x{W} per GPU, "
                               f"stage-{args.stage} trainable set, train-mode BN/Dropout2d on the frozen trunk",
                   "images_per_gpu": 2 * pairs, "height": H, "width": W, "stage": args.stage,
                   "parallelism": f"dp{world}",
                   "rccl_ranks": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else 0),
                   "loss_pairing": "device", "loss_sync": args.loss_sync, "loss": round(loss_val, 4)},
        "roofline": {"bound": "mfma", "kernel": dom + "_kernel (fp32 v_mfma_f32_32x32x2_f32)",
                     "measured": f"HIP events on the launch stream around every launch, second (untimed) pass of the same {args.steps} steps "
                                 f"({1e3 * elapsed_profiled / args.steps:.2f} ms/step with the event records)",
                     "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": round(alg_bytes / max(conv["launches"], 1)),
                     "launches_per_step": conv["launches"] // max(args.steps, 1),
                     "avg_launch_ms": round(conv["ms"] / max(conv["launches"], 1), 4),
                     "kernel_ms_per_step": round(conv["ms"] / max(args.steps, 1), 2)},
        "step_tflops_algorithmic": round(step_flops_alg / (elapsed / args.steps) / 1e12, 2),
    }
    out["mfma_kernels"] = {k + "_kernel": {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                           "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                           "launches_per_step": v["launches"] // max(args.steps, 1),
                                           "avg_launch_ms": round(v["ms"] / max(v["launches"], 1), 4),
                                           "kernel_ms_per_step": round(v["ms"] / max(args.steps, 1), 2)} for k, v in kinds.items()}
    if comm is not None:
        out["comm"] = comm
    if parity is not None:
        out["parity"] = parity
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    # they cannot be collected from inside this process); the summary of the last such run is kept in profiles/
    tpath = os.path.join(ROOT, "profiles", "conv_traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            fam = tj.get("families", {}).get(dom + "_kernel")
            out["roofline"]["traffic"] = round(fam["hbm_bytes_per_launch"] if fam else tj["hbm_bytes_per_launch"])
            # the same counters as a rate over this run's measured launch time (8000 GB/s HBM peak for comparison)
            out["roofline"]["memory_side_GBs"] = round(out["roofline"]["traffic"] / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e9, 1)
            out["roofline"]["traffic_source"] = "from_profile"
            out["roofline"]["traffic_note"] = ("NOT measured in this run: bytes per launch, L2-memory-side (Infinity-Cache hits included), from "
                                               "the rocprofv3 --pmc passes kept in profiles/conv_traffic_latest.json (tree "
                                               + str(tj.get("git_commit", "unknown")) + "): " + tj["correction"])
        except Exception:
            pass
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import peaks
        out["measured_peaks"] = {k: round(v, 1) for k, v in peaks.measure().items()}
    except Exception as exc:   # calibration is informative only
        out["measured_peaks"] = {"error": str(exc)}
    wn = summ.get("conv_winograd")
    if wn:
        out["winograd"] = {"algorithmic_tflops": round(wn["flops"] / (wn["ms"] * 1e-3) / 1e12, 1), "layers_per_step":
                           wn["launches"] // max(args.steps, 1), "ms_per_step": round(wn["ms"] / max(args.steps, 1), 2),
                           "note": "3x3 stride-1 layers with >= 64 channels run as Winograd F(6x6,3x3) (F(4x4) / F(2x2) where the larger tiles "
                                   "would be mostly padding, kernels.wino_tile): transforms + 64 (36, 16) batched GEMMs in one gemm_nt launch; TFLOP/s here = dense-conv FLOPs / time (can exceed the MFMA peak), "
                                   "while roofline.achieved counts only the FLOPs the MFMA kernel really executes"}
    wt = summ.get("wino_transform")
    if wt:
        out["winograd"]["transforms"] = {
            "ms_per_step": round(wt["ms"] / max(args.steps, 1), 2), "launches_per_step": wt["launches"] // max(args.steps, 1),
            "algorithmic_GBs": round(wt["flops"] / (wt["ms"] * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(wt["flops"] / (wt["ms"] * 1e-3) / 8e12, 4),
            "note": "input / output / grad-output transforms (HBM-bound): bytes = the NHWC tensor once + the (m+2)^2-position tensor once"}
    if wg:
        out["wgrad"] = {"achieved": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                        "kernel_ms_per_step": round(wg["ms"] / max(args.steps, 1), 2)}

    if world == 1 and not args.no_split:
        # CO-HEADLINE, reported beside `value` and never part of it: the same step with every product the
        # persistent GEMM kernel takes evaluated on the bf16 matrix cores -- operands as three bf16 terms, six bf16 MFMAs per block,
        # fp32 accumulation (csrc/gemm_bf16x3.hip; kernels.set_gemm_route). Same timing protocol as `value`, its own parity
        # against the reference fixture (taken before the first optimizer step) and its own roofline: fp32-equivalent TFLOP/s over
        # (bf16 dense peak / 6); the weight gradients whose shapes fit (K % 128, C % 256) run the split TN kernel, the implicit-GEMM
        # layers and the narrow heads stay on the native fp32 MFMA.
        K.set_gemm_route("bf16x3")
        try:
            for _ in range(max(1, args.warmup)):
                one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
            prof_s = K.ConvProfile()
            K.set_conv_profile(prof_s)
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            K.set_conv_profile(None)
            ss_ = prof_s.summary()
            gs = ss_.get("gemm_nt_bf16x3", dict(launches=0, flops=0.0, ms=1.0))
            ach = gs["flops"] / (gs["ms"] * 1e-3) / 1e12 if gs["launches"] else 0.0
            rf = {"bound": "mfma", "kernel": "gemm_nt_bf16x3_kernel (six bf16 products per fp32 product: 3 x v_mfma_f32_16x16x32_bf16 on concatenated planes per 16x16x16 block for the products without a prologue, 6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block for the prologue / implicit-GEMM kernels)",
                  "achieved": round(ach, 2), "peak": round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), "unit": "TFLOP/s (fp32-equivalent)",
                  "frac": round(ach / (BF16_MFMA_PEAK_TFLOPS / 6.0), 4), "launches_per_step": gs["launches"] // max(args.steps, 1),
                  "avg_launch_ms": round(gs["ms"] / max(gs["launches"], 1), 4), "kernel_ms_per_step": round(gs["ms"] / max(args.steps, 1), 2),
                  "peak_note": "2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md) / 6 MFMA products per fp32 product; under this load the chip "
                               "holds 1.6 (32x32x16) / 1.85 GHz (16x16x32) of the nominal 2.4 (profiles/r06/pmc_split_mfma32.md, pmc_split_mfma16_linear_image.md: matrix pipe busy share, effective clock)"}
            ppath = os.path.join(ROOT, "profiles", "split_pmc_latest.json")
            if os.path.exists(ppath):
                try:
                    rf["pmc"] = json.load(open(ppath))
                except Exception:
                    pass
            others = {k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "kernel_ms_per_step": round(v["ms"] / max(args.steps, 1), 2)}
                      for k, v in ss_.items() if k in ("gemm_nt", "conv_igemm", "conv_wgrad") and v["launches"]}
            wg3 = ss_.get("conv_wgrad_bf16x3")
            if wg3 and wg3["launches"]:
                a3 = wg3["flops"] / (wg3["ms"] * 1e-3) / 1e12
                rf["weight_gradient_kernel"] = {"kernel": "gemm_tn_bf16x3_kernel", "achieved": round(a3, 2), "frac": round(a3 / (BF16_MFMA_PEAK_TFLOPS / 6.0), 4),
                                                "launches_per_step": wg3["launches"] // max(args.steps, 1), "kernel_ms_per_step": round(wg3["ms"] / max(args.steps, 1), 2)}
            out["value_fp32_via_bf16x3"] = {
                "value": round(2 * pairs / dt, 4), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3),
                "dtype": "f32 (operands split into 3 bf16 terms, 6 bf16 MFMAs, fp32 accumulate)",
                "vs_native_value": round((2 * pairs / dt) / value, 4), "parity": parity_split, "roofline": rf,
                "native_fp32_mfma_kernels_left_in_the_step": others,
                "note": "same step, same timing protocol; route = kernels.set_gemm_route('bf16x3') / MSS_GEMM_SPLIT=1; every reference fixture "
                        "and oracle test that exercises a GEMM runs on both routes with the same bounds (tests/conftest.py gemm_route)"}
        finally:
            K.set_gemm_route(None)
            K.set_conv_profile(None)

    if not args.no_ood:
        # OOD-score path (test_deeplab.py:86-90): eval forward -> per-pixel anomaly score
        e_img = img[:1].contiguous()
        for _ in range(2):
            ood_scores(model, e_img)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_eval = 5
        for _ in range(n_eval):
            score, logit = ood_scores(model, e_img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / n_eval
        # the fused tail kernel alone (23 B per output pixel algorithmic: SURVEY 8d)
        dec = K.Act(torch.randn(1, H // 2, W // 2, 48, device=device))
        for _ in range(3):
            K.ood_score(dec.slice(20, 19), None, H, W, want_logit=False)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            K.ood_score(dec.slice(20, 19), None, H, W, want_logit=False)
        e.record()
        torch.cuda.synchronize()
        k_ms = s.elapsed_time(e) / 20
        # ... and in the form the eval path runs it: score AND the upsampled 19-class logits, 118 B per output pixel
        # (2 x 19 x 4 / 4 read + 4 + 76 written)
        for _ in range(3):
            K.ood_score(dec.slice(20, 19), dec.slice(0, 19), H, W)
        s.record()
        for _ in range(20):
            K.ood_score(dec.slice(20, 19), dec.slice(0, 19), H, W)
        e.record()
        torch.cuda.synchronize()
        kl_ms = s.elapsed_time(e) / 20
        from multishiftseg_amd.trainer import GraphedEval

        def time_eval(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            tt = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - tt) / n
        dt_so = time_eval(lambda: ood_scores(model, e_img, score_only=True))
        # the reference's test loader hands over valid_batch = 2 images at a time (test_deeplab.py:47, exps/DeepLab.yaml:18)
        e_img2 = img[:2].contiguous() if img.shape[0] >= 2 else torch.cat([e_img, e_img])
        dt_b2 = time_eval(lambda: ood_scores(model, e_img2, score_only=True))
        ge = GraphedEval(model, e_img.shape, score_only=True)
        dt_graph = time_eval(lambda: ge(e_img))
        del ge
        K.set_gemm_route("bf16x3")                # co-headline of the OOD-score path: the same eval forward on the split-bf16 GEMM route
        try:
            dt_so3 = time_eval(lambda: ood_scores(model, e_img, score_only=True))
            dt_b23 = time_eval(lambda: ood_scores(model, e_img2, score_only=True))
            ge3 = GraphedEval(model, e_img.shape, score_only=True)
            dt_graph3 = time_eval(lambda: ge3(e_img))
            del ge3
        finally:
            K.set_gemm_route(None)
        out["ood_score"] = {"end_to_end_mpix_s": round(H * W / dt / 1e6, 3), "end_to_end_ms": round(dt * 1e3, 2),
                            "bf16x3_route": {"score_only_mpix_s": round(H * W / dt_so3 / 1e6, 3), "score_only_ms": round(dt_so3 * 1e3, 2),
                                             "score_only_batch2_mpix_s": round(2 * H * W / dt_b23 / 1e6, 3),
                                             "score_only_hipgraph_mpix_s": round(H * W / dt_graph3 / 1e6, 3),
                                             "dtype": "f32 (operands split into 3 bf16 terms, 6 bf16 MFMAs, fp32 accumulate)"},
                            "score_only_mpix_s": round(H * W / dt_so / 1e6, 3), "score_only_ms": round(dt_so * 1e3, 2),
                            "score_only_hipgraph_mpix_s": round(H * W / dt_graph / 1e6, 3), "score_only_hipgraph_ms": round(dt_graph * 1e3, 2),
                            "score_only_batch2_mpix_s": round(2 * H * W / dt_b2 / 1e6, 3), "score_only_batch2_ms": round(dt_b2 * 1e3, 2),
                            "note": "end_to_end: eval forward -> (score, logits), eager; score_only: what test_deeplab.py:92-96 consumes; "
                                    "hipgraph: the same forward captured once and replayed (trainer.GraphedEval); batch2: two images per call, the batch "
                                    "the reference's test loader uses (test_deeplab.py:47 batch_size = valid_batch = 2, exps/DeepLab.yaml:18)",
                            "tail_kernel_mpix_s": round(H * W / (k_ms * 1e-3) / 1e6, 1),
                            "tail_kernel_GBs": round(23.0 * H * W / (k_ms * 1e-3) / 1e9, 1),
                            "tail_kernel_with_logits_us": round(kl_ms * 1e3, 1),
                            "tail_kernel_with_logits_GBs": round(118.0 * H * W / (kl_ms * 1e-3) / 1e9, 1), "hbm_peak_GBs": 8000,
                            "image": f"1x3x{H}x{W}"}

    if world == 1 and not args.no_m2f:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import m2f_legs
            out["m2f"] = m2f_legs.measure()
            if cpu is not None:
                cm = cpu_baseline_msda()
                gpu_ms = out["m2f"]["msda"]["c4_n1"]["forward_ms"]
                cm["gpu_forward_ms"] = gpu_ms
                out["m2f"]["msda"]["cpu_baseline"] = cm
        except Exception as exc:
            out["m2f"] = {"error": repr(exc)}

    if cpu is not None:
        ratio = STAGE2_STEP_OVER_FWD if args.stage == 2 else 1.0
        out["cpu_baseline"] = {
            "value": round(1.0 / (cpu["bench_fwd_s"] * ratio), 6), "unit": "images/s", "cores": cpu["host"]["torch_threads"],
            "kind": "port", "extrapolated": True,
            "sample": f"EXTRAPOLATED, not a timed CPU train step: stock torch CPU ops composing the same graph (oracle/deepv3_torch.py): ONE eval forward of a 1x3x{H}x{W} image "
                      f"({cpu['bench_fwd_s']:.2f} s) x the stage-{args.stage} step/forward FLOP ratio {ratio} (SURVEY 8d) -> train images/s; "
                      "C1 (1x3x512x1024 eval forward + OOD score): 1 warm-up + 3 runs, median, in `c1`",
            "c1": {"image": "1x3x512x1024", "torch_runs_s": cpu["c1_runs_s"], "torch_median_s": round(cpu["c1_median_s"], 3),
                   "torch_mpix_s": round(512 * 1024 / cpu["c1_median_s"] / 1e6, 4),
                   "torch_tflops": round(2 * 1456.8e9 / cpu["c1_median_s"] / 1e12, 3),
                   "numpy_restatement_s": round(cpu["c1_numpy_s"], 2),
                   "numpy_mpix_s": round(512 * 1024 / cpu["c1_numpy_s"] / 1e6, 4),
                   "numpy_vs_torch_max_abs_logit_diff": cpu["c1_numpy_vs_torch_max_abs_logit_diff"]},
            "host": cpu["host"]}
    # the fused loss alone at this configuration (value + both gradients, targets cloned outside the timed region)
    if world == 1 and not args.no_parity:
        try:
            lg = torch.randn(2 * pairs, 19, H, W, device="cuda") * 3
            sc_ = torch.randn(2 * pairs, H, W, device="cuda") * 4
            pool = [target0.clone() for _ in range(12)]
            for _ in range(3):
                crit.value_and_grads(lg, sc_, pool.pop())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                crit.value_and_grads(lg, sc_, pool.pop())
            e1.record()
            torch.cuda.synchronize()
            out["loss_kernel"] = {"ms": round(e0.elapsed_time(e1) / 8, 4), "algorithmic_bytes_per_pixel": 168,
                                  "GBs": round(2 * pairs * H * W * 168 / (e0.elapsed_time(e1) / 8) / 1e6, 1),
                                  "note": "RelContrastiveLoss value + d/dlogits + d/dscore, one C call (csrc/loss.hip)"}
            del lg, sc_, pool
        except Exception as exc:
            out["loss_kernel"] = {"error": repr(exc)}
    # compact summary LAST (the driver keeps the tail of this line): the numbers a reviewer reads first, without the notes
    summ_out = {"ms_per_step": out["ms_per_step"], "img_s": out["value"], "gemm_frac": out["roofline"]["frac"]}
    if "value_fp32_via_bf16x3" in out:
        v3 = out["value_fp32_via_bf16x3"]
        summ_out["bf16x3"] = {"img_s": v3["value"], "ms": v3["ms_per_step"], "gemm_tflops": v3["roofline"]["achieved"], "frac": v3["roofline"]["frac"],
                              "logit_err": (v3["parity"] or {}).get("max_abs_logit_err"), "flips": (v3["parity"] or {}).get("argmax_flips_all_pixels")}
    if "winograd" in out and "transforms" in out["winograd"]:
        summ_out["wino_transforms_ms"] = out["winograd"]["transforms"]["ms_per_step"]
    if "wgrad" in out:
        summ_out["wgrad_tflops"] = out["wgrad"]["achieved"]
    if "ms" in out.get("loss_kernel", {}):
        summ_out["loss_ms"] = out["loss_kernel"]["ms"]
    if "ood_score" in out:
        o = out["ood_score"]
        summ_out["ood"] = {"mpix_s": o["score_only_mpix_s"], "b2_mpix_s": o["score_only_batch2_mpix_s"], "bf16x3_mpix_s": o["bf16x3_route"]["score_only_mpix_s"],
                           "tail_GBs": o["tail_kernel_GBs"], "tail_logits_GBs": o["tail_kernel_with_logits_GBs"]}
    if isinstance(out.get("m2f"), dict) and "msda" in out["m2f"]:
        m = out["m2f"]
        d16 = m["pixel_decoder_forward_features"]["c4_704x704_n16"]
        summ_out["m2f"] = {"msda_n16_fwd_ms": m["msda"]["c4_n16"]["forward_ms"], "msda_n16_bwd_ms": m["msda"]["c4_n16"]["backward_ms"],
                           "msda_n16_bwd_frac": m["msda"]["c4_n16"]["backward_frac_of_hbm_peak"], "msda_n1_bwd_ms": m["msda"]["c4_n1"]["backward_ms"],
                           "decoder_n16_fb_ms": d16["forward_backward_ms"], "decoder_n16_fb_ms_bf16x3": d16.get("bf16x3_route", {}).get("forward_backward_ms"),
                           "fused_score_ms": m["fused_score"]["fused_score_ms"], "metric_GBs": m["metric_sweep"]["update_GBs_of_12B_per_pixel"],
                           "metric_b2_GBs": m["metric_sweep"]["update_batch2_GBs_of_12B_per_pixel"],
                           "metric_many_GBs": m["metric_sweep"]["update_many_GBs_of_12B_per_pixel"]}
    out["summary"] = summ_out
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
