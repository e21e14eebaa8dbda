"""GPU parity of the device-side OOD metrics against the reference's own outputs (golden) and the
CPU oracle: exact rank statistics, so the tolerance is float64 rounding of the final divisions/sums."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import metric as ometric

pytestmark = pytest.mark.gpu
TOL = 1e-11


@pytest.fixture(scope="module")
def M():
    from multishiftseg_amd import metric
    return metric


def test_golden_reference_measures(M):
    g = golden("ood_metrics")
    for tag in sorted(k[:-len("_measures")] for k in g.files if k.endswith("_measures")):
        score = torch.from_numpy(g[tag + "_score"]).cuda()
        label = torch.from_numpy(g[tag + "_label"].astype(np.int64)).cuda()
        got = M.eval_ood_measure(score, label)
        np.testing.assert_allclose(got, g[tag + "_measures"], rtol=0, atol=TOL, err_msg=tag)


@pytest.mark.parametrize("n,quant,ppos,seed", [(1, None, 0.5, 0), (2, None, 0.5, 1), (63, 4, 0.3, 2), (64, 1, 0.5, 3), (65, None, 0.9, 4),
                                               (1000, 2, 0.05, 5), (4097, 16, 0.5, 6), (100000, None, 0.02, 7),
                                               (300000, 64, 0.4, 8), (1 << 20, 1024, 0.1, 9)])
def test_random_vs_oracle(M, n, quant, ppos, seed):
    rng = np.random.default_rng(seed)
    r = rng.random(n)
    label = np.where(r < ppos, 1, np.where(r < ppos + (1 - ppos) * 0.9, 0, 255)).astype(np.int64)
    if n <= 2:
        label[:] = [1, 0][:n]
    score = (rng.standard_normal(n) + 0.7 * (label == 1)).astype(np.float32)
    if quant:
        score = (np.round(score * quant) / quant).astype(np.float32)
    want = ometric.eval_ood_measure(score, label)
    got = M.eval_ood_measure(torch.from_numpy(score).cuda(), torch.from_numpy(label).cuda())
    if want is None:
        assert got is None
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=TOL)


def test_edge_cases(M):
    z = torch.zeros(0, device="cuda")
    assert M.eval_ood_measure(z, z.long()) is None                                   # empty sweep
    s = torch.randn(2, 8, 8, device="cuda")
    assert M.eval_ood_measure(s, torch.zeros(2, 8, 8, dtype=torch.long, device="cuda")) is None      # no OOD pixel
    assert M.eval_ood_measure(s, torch.ones(2, 8, 8, dtype=torch.long, device="cuda")) is None       # no inlier pixel
    assert M.eval_ood_measure(s, torch.full((2, 8, 8), 255, dtype=torch.long, device="cuda")) is None
    # all scores equal: one threshold, AUROC 0.5, AP = prevalence, FPR 1
    lab = torch.tensor([0, 1, 1, 0, 0, 255, 1, 0], device="cuda")
    got = M.eval_ood_measure(torch.full((8,), 3.5, device="cuda"), lab)
    np.testing.assert_allclose(got, (0.5, 3 / 7, 1.0), atol=1e-15)
    # perfectly separated; signed zeros are one threshold
    sc = torch.tensor([-0.0, 0.0, 1.0, 2.0], device="cuda")
    got = M.eval_ood_measure(sc, torch.tensor([0, 1, 1, 1], device="cuda"))
    want = ometric.eval_ood_measure(sc.cpu().numpy(), np.array([0, 1, 1, 1]))
    np.testing.assert_allclose(got, want, atol=1e-15)
    # other class ids (the reference's train_id_in / train_id_out arguments)
    lab = torch.randint(0, 4, (5000,), device="cuda")
    sc = torch.randn(5000, device="cuda")
    got = M.eval_ood_measure(sc, lab, train_id_in=2, train_id_out=3)
    want = ometric.eval_ood_measure(sc.cpu().numpy(), lab.cpu().numpy(), 2, 3)
    np.testing.assert_allclose(got, want, atol=TOL)
    with pytest.raises(RuntimeError):
        M.eval_ood_measure(sc.cpu(), lab.cpu())


def test_streaming_meter_equals_one_shot(M):
    """The sweep form (append per batch, test_deeplab.py:84-102) gives the same numbers as one call on the
    concatenation, whatever the batch split."""
    rng = np.random.default_rng(11)
    scores = [rng.standard_normal((b, 40, 50)).astype(np.float32) for b in (1, 3, 2, 1)]
    labels = [rng.choice([0, 1, 255], size=s.shape, p=[0.8, 0.1, 0.1]).astype(np.int64) for s in scores]
    labels[1][:] = 0                                                                 # a batch without OOD pixels
    meter = M.OODMeter()
    for s, l in zip(scores, labels):
        meter.update(torch.from_numpy(s).cuda(), torch.from_numpy(l).cuda())
    got = meter.compute()
    want = ometric.eval_ood_measure(np.concatenate(scores), np.concatenate(labels))
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    one = M.eval_ood_measure(torch.from_numpy(np.concatenate(scores)).cuda(), torch.from_numpy(np.concatenate(labels)).cuda())
    assert one == got                                                                # deterministic: bitwise equal


def test_full_size_properties(M):
    """BASELINE-size sweep (8 x 1024 x 2048 = 16.8 M pixels), size-independent properties: a monotone transform of
    the scores leaves all three metrics unchanged; negating scores maps AUROC to 1 - AUROC; relabelling in<->out
    with negated scores gives the same AUROC."""
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 8 * 1024 * 2048
    lab = (torch.rand(n, device="cuda", generator=g) < 0.03).long()
    lab[torch.rand(n, device="cuda", generator=g) < 0.05] = 255
    sc = torch.randn(n, device="cuda", generator=g) + 1.2 * (lab == 1)
    a = M.eval_ood_measure(sc, lab)
    b = M.eval_ood_measure(torch.exp(sc * 0.5), lab)           # strictly increasing (fp32 ties may merge: tolerance)
    np.testing.assert_allclose(a, b, atol=1e-6)
    c = M.eval_ood_measure(-sc, lab)
    np.testing.assert_allclose(c[0], 1 - a[0], atol=1e-12)
    d = M.eval_ood_measure(-sc, 1 - lab.clamp(max=2), train_id_in=0, train_id_out=1)   # 255 -> -1.. stays ignored
    np.testing.assert_allclose(d[0], a[0], atol=1e-12)
    assert 0.5 < a[0] < 1 and 0 < a[1] < 1 and 0 < a[2] < 1


@pytest.mark.parametrize("n,kind", [(1, "rand"), (5, "rand"), (4095, "rand"), (4096, "rand"), (4097, "dups"), (70001, "rand"),
                                    (1 << 20, "top"), (5_000_003, "rand"), (4096 * 1024 + 7, "dups"), (40_000_001, "rand")])
def test_own_radix_sort_equals_torch_sort(monkeypatch, n, kind):
    """mss_oodm_sort_u32 (own 4-pass LSD sort): sizes around the tile (4096) and chunk boundaries, more tiles than the
    1024 chunks, all-equal digits in the high bytes ('top': keys below 2^12), heavy duplicates; the rocPRIM A/B route gives
    the same array."""
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import call, ptr
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    if kind == "rand":
        keys = torch.randint(0, 2 ** 31 - 1, (n,), device="cuda", generator=g, dtype=torch.int64)
        keys = (keys * 2 + torch.randint(0, 2, (n,), device="cuda", generator=g)).to(torch.int64)
    elif kind == "dups":
        keys = torch.randint(0, 37, (n,), device="cuda", generator=g, dtype=torch.int64) * 0x01010101
    else:
        keys = torch.randint(0, 4096, (n,), device="cuda", generator=g, dtype=torch.int64)
    want = torch.sort(keys)[0]
    k32 = (keys & 0xffffffff).to(torch.int64)
    k32 = torch.where(k32 >= 2 ** 31, k32 - 2 ** 32, k32).to(torch.int32).contiguous()       # same bits as uint32
    outs = {}
    pad = torch.empty(n + 3, dtype=torch.int32, device="cuda")
    off = n % 4                                          # input slices that are only 4-byte aligned, as the meter's are
    pad[off:off + n] = k32
    k32 = pad[off:off + n]
    for route in ("own", "rocprim"):
        monkeypatch.setenv("MSS_OODM_SORT", route)
        out = torch.empty_like(want, dtype=torch.int32)
        temp = torch.empty(_lib.value("mss_oodm_sort_temp_bytes", n), dtype=torch.uint8, device="cuda")
        call("mss_oodm_sort_u32", ptr(k32), ptr(out), n, ptr(temp), temp.numel())
        outs[route] = out
        got = out.to(torch.int64) & 0xffffffff
        assert torch.equal(got, want), route
    assert torch.equal(outs["own"], outs["rocprim"])


def test_update_many_equals_update_in_a_loop():
    """The batched hand-over (16 maps per launch, one key buffer per group) gives the measures of per-map updates -- 37 ragged maps
    (one empty, one without OOD pixels), so two full groups and a partial one."""
    from multishiftseg_amd import metric as M
    g = torch.Generator(device="cuda").manual_seed(5)
    maps = []
    for i in range(37):
        h, w = 17 + 13 * i, 50 + 7 * (i % 5)
        if i == 9:
            h = 0
        lab = (torch.rand(1, h, w, device="cuda", generator=g) < (0.0 if i == 20 else 0.1)).long()
        lab[torch.rand(1, h, w, device="cuda", generator=g) < 0.05] = 255
        maps.append((torch.randn(1, h, w, device="cuda", generator=g) + 1.5 * (lab == 1), lab))
    a, b = M.OODMeter(), M.OODMeter()
    for s, l in maps:
        a.update(s, l)
    b.update_many(maps)
    ra, rb = a.compute(), b.compute()
    assert ra is not None and ra == rb
