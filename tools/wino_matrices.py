"""Cook-Toom construction of the Winograd F(4x4,3x3) and F(6x6,3x3) matrices used by csrc/winograd.hip.

Interpolation points (0, +-11/16, +-3/2, inf) instead of the textbook (0, +-1, +-2, inf): measured on
ReLU-like inputs with 512 channels, fp32 rms error 1.6e-6 instead of 3.6e-6 (max 2.4e-6 instead of 9.4e-6).
A^T = V_m^T, G = D^-1 V_r, B^T = D V_n^-T with V_k the evaluation matrix of degree-(k-1) polynomials at the
points and D a power-of-two row scaling that keeps the B^T rows near unit magnitude (numerically neutral).
Prints the C initialisers; `python tools/wino_matrices.py check` also verifies the identity in float64 and
reports the fp32 error against a direct float64 convolution.

F(6x6,3x3) (`python tools/wino_matrices.py 6 [check]`): points (0, +-1/2, +-1, +-2, inf). A search over ~1500 dyadic
triples (+-a, +-b, +-c) found nothing better than noise around it (rms 5.7e-6 .. 6.2e-6 of the output at 512 channels,
against 1.9e-6 for F(4x4) above), so the textbook set with its many +-1 entries stays.
"""
import sys
from fractions import Fraction as Fr

import numpy as np

POINTS = (Fr(0), Fr(11, 16), Fr(-11, 16), Fr(3, 2), Fr(-3, 2))
POINTS6 = (Fr(0), Fr(1, 2), Fr(-1, 2), Fr(1), Fr(-1), Fr(2), Fr(-2))


def matrices(pts=POINTS, m=4, r=3):
    n = m + r - 1

    def vand(cols):
        rows = [[p ** k for k in range(cols)] for p in pts]
        rows.append([Fr(0)] * (cols - 1) + [Fr(1)])
        return rows
    at = [[vand(m)[j][i] for j in range(n)] for i in range(m)]
    g = vand(r)
    a = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(vand(n))]
    for c in range(n):                                   # exact Gauss-Jordan inverse
        piv = next(i for i in range(c, n) if a[i][c] != 0)
        a[c], a[piv] = a[piv], a[c]
        pv = a[c][c]
        a[c] = [x / pv for x in a[c]]
        for i in range(n):
            if i != c and a[i][c] != 0:
                f = a[i][c]
                a[i] = [x - f * y for x, y in zip(a[i], a[c])]
    bt = [[a[j][n + i] for j in range(n)] for i in range(n)]
    for j in range(n):
        big = max(abs(x) for x in bt[j])
        s = Fr(1)
        while big * s < Fr(3, 4):
            s *= 2
        while big * s >= Fr(3, 2):
            s /= 2
        bt[j] = [x * s for x in bt[j]]
        g[j] = [x / s for x in g[j]]
    return at, g, bt


def as_np(mat):
    return np.array([[float(x) for x in row] for row in mat])


def c_init(name, mat):
    def lit(x):
        t = f"{float(x):.9g}"
        return "0" if x == 0 else (t if "." in t or "e" in t else t + ".0") + "f"
    rows = ["{" + ", ".join(lit(x) for x in row) + "}" for row in mat]
    return f"static constexpr float {name}[{len(mat)}][{len(mat[0])}] = {{\n      " + ",\n      ".join(rows) + "};"


def check(pts=POINTS, m=4):
    at, g, bt = (as_np(x) for x in matrices(pts, m))
    rng = np.random.default_rng(0)
    C, T, n = 512, 48, m + 2
    d = np.maximum(rng.standard_normal((T, C, n, n)), 0)
    w = rng.standard_normal((C, 3, 3)) / np.sqrt(C * 9)
    ref = np.zeros((T, m, m))
    for i in range(m):
        for j in range(m):
            ref[:, i, j] = (d[:, :, i:i + 3, j:j + 3] * w).sum((1, 2, 3))

    def run(dt):
        a_, g_, b_ = at.astype(dt), g.astype(dt), bt.astype(dt)
        u = np.einsum("ik,ckl,jl->cij", g_, w.astype(dt), g_).astype(dt)
        v = np.einsum("ik,tckl,jl->tcij", b_, d.astype(dt), b_).astype(dt)
        acc = np.zeros((T, n, n), dt)
        for c in range(C):
            acc += u[c] * v[:, c]
        return np.einsum("ik,tkl,jl->tij", a_, acc, a_)
    e64 = np.abs(run(np.float64) - ref).max() / np.abs(ref).max()
    e32 = run(np.float32) - ref
    print(f"float64 identity error {e64:.2e}; float32 rms {np.sqrt((e32 ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e} "
          f"max {np.abs(e32).max() / np.abs(ref).max():.2e}")


if __name__ == "__main__":
    pts, m = (POINTS6, 6) if "6" in sys.argv[1:] else (POINTS, 4)
    at, g, bt = matrices(pts, m)
    print(c_init("Bt", bt)); print(c_init("G", g)); print(c_init("At", at))
    if "check" in sys.argv[1:]:
        check(pts, m)
