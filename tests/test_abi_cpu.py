"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/mss_hip.h declares; the host mirrors keep the reference's names. No compute."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from multishiftseg_amd import _lib
    return _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "mss_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(mss_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_functions()
    assert len(names) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (mss_\w+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in mss_hip.h but not exported: {missing}"
    unbound = [n for n in names if n not in lib.SIGNATURES]
    assert not unbound, f"declared in mss_hip.h but not bound in _lib.py: {unbound}"
    extra = [n for n in exported if n not in names]
    assert not extra, f"exported but not declared in mss_hip.h: {extra}"
    handle = lib.load()
    assert handle.mss_abi_version() == lib.MSS_ABI_VERSION == int(re.search(r"#define MSS_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "mss_hip.h")).read()).group(1))
    assert lib.value("mss_conv2d_kpad", 19) == 64 and lib.value("mss_conv2d_kpad", 304) == 384


def test_msda_binned_workspace_query_is_host_only(lib):
    """mss_msda_backward_workspace_bytes is pure host arithmetic (tile geometry from a HOST copy of spatial_shapes): 16 bytes per
    sample + halos + counters for the shapes the binned path takes, 0 for the ones it does not."""
    import ctypes

    def q(shapes, N, M, D, Lq, P):
        flat = [int(v) for hw in shapes for v in hw]
        return lib.value("mss_msda_backward_workspace_bytes", (ctypes.c_int64 * len(flat))(*flat), N, M, D, len(shapes), Lq, P)
    c4 = [(22, 22), (44, 44), (88, 88)]
    n16 = q(c4, 16, 8, 32, 10164, 4)
    samples = 16 * 10164 * 8 * 3 * 4
    assert 16 * samples < n16 < 16 * samples * 1.25                 # records dominate; halos + counters < 25 % on top
    assert q(c4, 1, 8, 32, 10164, 4) < n16 // 8
    assert q([(1, 300), (300, 1), (17, 17), (5, 3)], 2, 8, 32, 257, 4) > 0      # thin / tiny levels are taken
    assert q(c4, 16, 8, 64, 10164, 4) == 0                         # D != 32
    assert q([(4, 4)] * 9, 1, 8, 32, 144, 4) == 0                   # more than 8 levels
    assert q(c4, 0, 8, 32, 10164, 4) == 0 and q(c4, 1, 8, 32, 0, 4) == 0
    assert q([(0, 5)], 1, 8, 32, 10, 4) == 0                        # degenerate level


def test_struct_layout_matches_header(lib):
    """sizeof(MssConvArgs)/sizeof(MssRclArgs) as the C compiler sees them."""
    import ctypes
    code = '#include <stdio.h>\n#include "mss_hip.h"\nint main(){printf("%zu %zu\\n", sizeof(MssConvArgs), sizeof(MssRclArgs));return 0;}'
    exe = "/tmp/mss_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=code.encode(), check=True)
    a, b = (int(v) for v in subprocess.check_output([exe], text=True).split())
    assert ctypes.sizeof(lib.MssConvArgs) == a and ctypes.sizeof(lib.MssRclArgs) == b


def test_product_path_fails_loudly_without_gpu():
    import torch
    from multishiftseg_amd.loss import RelContrastiveLoss
    from multishiftseg_amd import MultiScaleDeformableAttention as MSDA
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    crit = RelContrastiveLoss({"inoutaug_contras_margins_tri": [10, 5, 5]})
    with pytest.raises(RuntimeError):
        crit(torch.zeros(2, 19, 4, 4), torch.zeros(2, 4, 4), torch.zeros(2, 4, 4, dtype=torch.int64))
    v = torch.zeros(1, 30, 2, 2)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_forward(v, torch.tensor([[6, 4], [3, 2]]), torch.tensor([0, 24]), torch.zeros(1, 2, 2, 2, 2, 2),
                                    torch.zeros(1, 2, 2, 2, 2), 2)


def test_no_product_import_of_oracle():
    """The oracle is test infrastructure: nothing under multishiftseg_amd/ may import it."""
    pkg = os.path.join(ROOT, "multishiftseg_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
            assert "/root/reference" not in src, fn


def test_deeplab_module_names(deeplab_params):
    import torch
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    with torch.device("meta"):
        m = DeepWV3Plus(19)
    sd = m.state_dict()
    assert list(sd.keys()) == list(deeplab_params.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(deeplab_params[k].shape), k


def test_lazily_packed_wino_weight_refuses_a_modified_source():
    """a WinoWeight made without U (split route) builds U / planes on first use from `src`, which aliases the live
    parameter; after an in-place update it must raise instead of mixing two weight versions in one convolution."""
    import torch
    from multishiftseg_amd import kernels as K
    w = torch.nn.Parameter(torch.randn(128, 16, 3, 3))
    pw = K.WinoWeight(None, 128, 16, 128, 16, 4, src=w.detach().contiguous())
    pw._check_fresh()
    with torch.no_grad():
        w.mul_(0.5)
    with pytest.raises(RuntimeError, match="stale WinoWeight"):
        pw.t
    with pytest.raises(RuntimeError, match="stale WinoWeight"):
        pw.fused_planes()


@pytest.fixture(scope="module")
def split_gemm_compile(tmp_path_factory):
    """gemm_bf16x3.hip compiled ONCE for gfx950 (device side only): hipcc's kernel-resource-usage remarks and the assembly text."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc here")
    src = os.path.join(ROOT, "multishiftseg_amd", "csrc", "gemm_bf16x3.hip")
    out = tmp_path_factory.mktemp("split_gemm") / "gemm_bf16x3.s"
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S", src, "-o", str(out),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr, out.read_text()


def test_hot_split_gemm_kernels_use_no_scratch(split_gemm_compile):
    """the split-bf16 kernels run at the 256-register cap of two workgroups per CU, and a spill inside the K-loop
    shares the vector-memory counter with the tile prefetch (scratch reloads wait with vmcnt(0)). Every instantiation the step's hot
    products take must be compiled WITHOUT scratch: hipcc's own kernel-resource-usage remarks for gemm_bf16x3.hip, parsed here.
    Template arguments in the mangled names: <AFFINE, BN, CONV, ROWAFF, DYN (ticket tile order), MF (MFMA shape)>."""
    import re
    remarks, _ = split_gemm_compile
    usage, name = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name and "AGPRs" not in line:
                usage[name][key] = int(m.group(1))

    def nt(affine, bn, conv, rowaff, dyn, mf):
        tag = f"gemm_nt_bf16x3_kernelILb{affine}ELi{bn}ELb{conv}ELb{rowaff}ELb{dyn}ELi{mf}E"
        hits = [v for k, v in usage.items() if tag in k]
        assert len(hits) == 1, (tag, sorted(usage))
        return hits[0]
    hot = {"plain 128x256, 16x16x32": nt(0, 256, 0, 0, 0, 16), "plain 128x128, 16x16x32": nt(0, 128, 0, 0, 0, 16), "plain 128x128 (fallback)": nt(0, 128, 0, 0, 0, 32),
           "prologue 128x256, ticket order": nt(1, 256, 0, 0, 1, 32), "prologue 128x128, ticket order": nt(1, 128, 0, 0, 1, 32),
           "implicit GEMM 128x256": nt(0, 256, 1, 0, 0, 32), "implicit GEMM 128x128": nt(0, 128, 1, 0, 0, 32),
           "per-sample prologue (ROWAFF)": nt(1, 128, 0, 1, 0, 32)}
    for k, v in usage.items():
        if "gemm_tn_bf16x3_kernelILb0ELb0E" in k or "gemm_tn_bf16x3_kernelILb1ELb0E" in k:          # the unmasked weight-gradient kernels
            hot["TN " + k[-40:]] = v
    assert len(hot) == 10, sorted(hot)
    bad = {k: v for k, v in hot.items() if v["ScratchSize [bytes/lane]"] != 0 or v["VGPRs"] > 256}
    assert not bad, bad
    assert hot["plain 128x256, 16x16x32"]["Occupancy [waves/SIMD]"] == 2 and hot["plain 128x128, 16x16x32"]["Occupancy [waves/SIMD]"] == 3


def test_lds_dma_wait_counts_exactly_the_a_tile_loads(split_gemm_compile):
    """The 16x16x32 split kernels bring the B tile in with LDS-DMA and close every K-step with `s_waitcnt vmcnt(N_A_LOADS)`: the
    vector-memory counter retires in order, so that wait covers the DMA pieces only if EXACTLY N_A_LOADS (2) vector-memory
    instructions - the next A tile's two 16-byte loads - were issued after the last DMA piece. A compiler that splits a load, sinks
    another one below the DMA or inserts a scratch access would turn the wait into a race without any build error; this reads the
    generated gfx950 assembly and checks the instruction order at every such wait (prologue, steady state, tile switch)."""
    import re
    _, asm = split_gemm_compile
    lines = asm.split("\n")
    vmem = re.compile(r"^\s+((?:global|buffer|scratch|flat)_\w+)")
    func, waits = None, {}
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func = m.group(1)
        if "s_waitcnt vmcnt(2) lgkmcnt(0)" not in line:
            continue
        ops, j = [], i - 1
        while j > 0 and "global_load_lds_dwordx4" not in lines[j]:
            assert not re.match(r"^_Z\w+:", lines[j]), (func, "no LDS-DMA before the counted wait")
            m = vmem.match(lines[j])
            if m:
                ops.append(m.group(1))
            j -= 1
        waits.setdefault(func, []).append(ops)
    m16 = {k: v for k, v in waits.items() if "gemm_nt_bf16x3_kernel" in k and "ELi16EEE" in k}
    assert len(m16) == 2, sorted(waits)                                     # the 128x256 and 128x128 plain kernels
    for k, per_wait in m16.items():
        assert len(per_wait) >= 2, (k, per_wait)                            # at least the tile prologue and the steady-state step
        for ops in per_wait:
            assert ops == ["global_load_dwordx4", "global_load_dwordx4"], (k, ops)
