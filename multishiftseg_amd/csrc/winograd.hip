// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions with many channels (mod4..mod7 of the
// WideResNet trunk, wider_resnet.py:322-332): 2.25x fewer multiplications, still exact-fp32 MFMA.
//
//   y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, per (k, c), summed over c
//
// Three steps, the middle one being 16 independent GEMMs [T x C] x [C x K] run by the implicit-GEMM
// kernel in batched 1x1 mode (conv_igemm.hip):
//   1. wino_input_transform : NHWC x (+ fused BatchNorm/ReLU prologue, zero padding) -> X'[16][T][C]
//   2. batched GEMM         : X'[p] * W'[p]^T -> Y'[16][T][K]
//   3. wino_output_transform: Y' -> NHWC y (+ residual add), 2x2 pixels per tile
// Dilation d is handled exactly: output pixel (oy, ox) belongs to sub-grid (oy % d, ox % d), inside a
// sub-grid the conv is an ordinary dense 3x3 with padding 1 on the d-subsampled image, so tiles are
// taken per sub-grid. T = N * d*d * ceil(ceil(H/d)/2) * ceil(ceil(W/d)/2).
// Both transforms are HBM-bound float4 kernels; they pay off when C and K are >= 512.
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct WinoGeom {
  int N, H, W, d, Hs, Ws, tH, tW;   // Hs/Ws: sub-grid extent, tH/tW: tiles per sub-grid
  long long T;
};

__host__ __device__ inline WinoGeom wino_geom(int N, int H, int W, int d) {
  WinoGeom g;
  g.N = N; g.H = H; g.W = W; g.d = d;
  g.Hs = (H + d - 1) / d; g.Ws = (W + d - 1) / d;
  g.tH = (g.Hs + 1) / 2; g.tW = (g.Ws + 1) / 2;
  g.T = (long long)N * d * d * g.tH * g.tW;
  return g;
}

// w [K][C][3][3] -> U [16][Kpad][Cp] with U[xi*4+nu][k][c] = (G g G^T)[xi][nu]
__global__ void wino_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int K, int C, int Kpad,
                                         int Cp) {
  const long long total = (long long)Kpad * Cp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp), k = (int)(i / Cp);
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) g[r][s] = (k < K && c < C) ? w[(((size_t)k * C + c) * 3 + r) * 3 + s] : 0.f;
    float t[4][3];   // G g
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const float u0 = t[xi][0], u1 = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]),
                  u2 = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), u3 = t[xi][2];
      const size_t base = ((size_t)(xi * 4) * Kpad + k) * Cp + c;
      const size_t ps = (size_t)Kpad * Cp;
      u[base] = u0; u[base + ps] = u1; u[base + 2 * ps] = u2; u[base + 3 * ps] = u3;
    }
  }
}

__global__ __launch_bounds__(256) void wino_input_transform_kernel(
    const float* __restrict__ x, int ldx, WinoGeom g, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, int relu, float* __restrict__ xt) {
  const int C4 = C >> 2;
  const long long total = g.T * C4;
  const float floor_v = relu ? 0.f : -__builtin_huge_valf();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long t = i / C4;
    const long long tile = t;
    const int tx = (int)(t % g.tW); t /= g.tW;
    const int ty = (int)(t % g.tH); t /= g.tH;
    const int b = (int)(t % g.d); t /= g.d;
    const int a = (int)(t % g.d);
    const int n = (int)(t / g.d);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) { sc = ld4(scale + c); sh = ld4(shift + c); }
    f32x4 dv[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int sy = 2 * ty - 1 + ii;
      const int iy = sy * g.d + a;
      const bool oky = sy >= 0 && iy < g.H;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int sx = 2 * tx - 1 + jj;
        const int ix = sx * g.d + b;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (oky && sx >= 0 && ix < g.W) {
          v = ld4(x + ((size_t)(n * g.H + iy) * g.W + ix) * ldx + c) * sc + sh;
          v.x = fmaxf(v.x, floor_v); v.y = fmaxf(v.y, floor_v); v.z = fmaxf(v.z, floor_v); v.w = fmaxf(v.w, floor_v);
        }
        dv[ii][jj] = v;
      }
    }
    // B^T d B with B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    f32x4 tmp[4][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      tmp[0][jj] = dv[0][jj] - dv[2][jj];
      tmp[1][jj] = dv[1][jj] + dv[2][jj];
      tmp[2][jj] = dv[2][jj] - dv[1][jj];
      tmp[3][jj] = dv[1][jj] - dv[3][jj];
    }
    const size_t ps = (size_t)g.T * C;
    float* o = xt + (size_t)tile * C + c;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      st4(o + (size_t)(xi * 4 + 0) * ps, tmp[xi][0] - tmp[xi][2]);
      st4(o + (size_t)(xi * 4 + 1) * ps, tmp[xi][1] + tmp[xi][2]);
      st4(o + (size_t)(xi * 4 + 2) * ps, tmp[xi][2] - tmp[xi][1]);
      st4(o + (size_t)(xi * 4 + 3) * ps, tmp[xi][1] - tmp[xi][3]);
    }
  }
}

__global__ __launch_bounds__(256) void wino_output_transform_kernel(const float* __restrict__ yt, WinoGeom g, int K,
                                                                    const float* __restrict__ res, int ldres,
                                                                    float* __restrict__ y, int ldy) {
  const int K4 = K >> 2;
  const long long total = g.T * K4;
  const size_t ps = (size_t)g.T * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K4) * 4;
    long long t = i / K4;
    const long long tile = t;
    const int tx = (int)(t % g.tW); t /= g.tW;
    const int ty = (int)(t % g.tH); t /= g.tH;
    const int b = (int)(t % g.d); t /= g.d;
    const int a = (int)(t % g.d);
    const int n = (int)(t / g.d);
    const float* src = yt + (size_t)tile * K + k;
    f32x4 m[4][4];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) m[xi][nu] = ld4(src + (size_t)(xi * 4 + nu) * ps);
    // A^T m A with A^T = [[1,1,1,0],[0,1,-1,-1]]
    f32x4 r0[4], r1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      r0[nu] = m[0][nu] + m[1][nu] + m[2][nu];
      r1[nu] = m[1][nu] - m[2][nu] - m[3][nu];
    }
    f32x4 o[2][2];
    o[0][0] = r0[0] + r0[1] + r0[2]; o[0][1] = r0[1] - r0[2] - r0[3];
    o[1][0] = r1[0] + r1[1] + r1[2]; o[1][1] = r1[1] - r1[2] - r1[3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int oy = (2 * ty + u) * g.d + a;
      if (2 * ty + u >= g.Hs || oy >= g.H) continue;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int ox = (2 * tx + v) * g.d + b;
        if (2 * tx + v >= g.Ws || ox >= g.W) continue;
        const size_t pix = (size_t)(n * g.H + oy) * g.W + ox;
        f32x4 val = o[u][v];
        if (res) val += ld4(res + pix * ldres + k);
        st4(y + pix * ldy + k, val);
      }
    }
  }
}

// dy (NHWC) -> dY' [16][T][K] = A dY A^T per 2x2 tile (zero outside the image), A = (A^T)^T
__global__ __launch_bounds__(256) void wino_grad_output_transform_kernel(const float* __restrict__ dy, int lddy,
                                                                         WinoGeom g, int K, float* __restrict__ dyt) {
  const int K4 = K >> 2;
  const long long total = g.T * K4;
  const size_t ps = (size_t)g.T * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K4) * 4;
    long long t = i / K4;
    const long long tile = t;
    const int tx = (int)(t % g.tW); t /= g.tW;
    const int ty = (int)(t % g.tH); t /= g.tH;
    const int b = (int)(t % g.d); t /= g.d;
    const int a = (int)(t % g.d);
    const int n = (int)(t / g.d);
    f32x4 d[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int oy = (2 * ty + u) * g.d + a;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int ox = (2 * tx + v) * g.d + b;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (2 * ty + u < g.Hs && oy < g.H && 2 * tx + v < g.Ws && ox < g.W)
          val = ld4(dy + ((size_t)(n * g.H + oy) * g.W + ox) * lddy + k);
        d[u][v] = val;
      }
    }
    f32x4 tt[4][2];   // A dY : rows [d0, d0+d1, d0-d1, -d1]
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      tt[0][v] = d[0][v];
      tt[1][v] = d[0][v] + d[1][v];
      tt[2][v] = d[0][v] - d[1][v];
      tt[3][v] = -d[1][v];
    }
    float* o = dyt + (size_t)tile * K + k;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      st4(o + (size_t)(xi * 4 + 0) * ps, tt[xi][0]);
      st4(o + (size_t)(xi * 4 + 1) * ps, tt[xi][0] + tt[xi][1]);
      st4(o + (size_t)(xi * 4 + 2) * ps, tt[xi][0] - tt[xi][1]);
      st4(o + (size_t)(xi * 4 + 3) * ps, -tt[xi][1]);
    }
  }
}

// dU [16][Kpad][Cp] -> dg [K][C][3][3] = G^T dU G
__global__ void wino_weight_grad_transform_kernel(const float* __restrict__ du, float* __restrict__ dw, int K, int C,
                                                  int Kpad, int Cp) {
  const long long total = (long long)K * C;
  const size_t ps = (size_t)Kpad * Cp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), k = (int)(i / C);
    float u[4][4];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) u[xi][nu] = du[(size_t)(xi * 4 + nu) * ps + (size_t)k * Cp + c];
    float t[3][4];   // G^T dU
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t[0][nu] = u[0][nu] + 0.5f * (u[1][nu] + u[2][nu]);
      t[1][nu] = 0.5f * (u[1][nu] - u[2][nu]);
      t[2][nu] = 0.5f * (u[1][nu] + u[2][nu]) + u[3][nu];
    }
    float* o = dw + ((size_t)k * C + c) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      o[r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      o[r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
      o[r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
  }
}

inline int grid_for(long long work_items) {
  long long b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b));
}

}  // namespace

#define S_(x) static_cast<hipStream_t>(x)

extern "C" {

long long mss_wino_num_tiles(int N, int H, int W, int dil) { return wino_geom(N, H, W, dil).T; }

int mss_wino_pack_weights_f32(const float* w, float* u, int K, int C, int Kpad, int Cp, void* stream) {
  if (!w || !u || Kpad < K || Cp < C) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(wino_pack_weights_kernel, dim3(grid_for((long long)Kpad * Cp)), dim3(256), 0, S_(stream), w, u, K,
                     C, Kpad, Cp);
  return mss_launch_status();
}

int mss_wino_input_transform_f32(const float* x, int ldx, int N, int H, int W, int C, int dil, const float* scale,
                                 const float* shift, int relu, float* xt, void* stream) {
  if (!x || !xt || C % 4 || ldx % 4 || dil < 1) return MSS_ERR_BAD_ARG;
  if (scale && !shift) return MSS_ERR_BAD_ARG;
  const WinoGeom g = wino_geom(N, H, W, dil);
  if (g.T == 0) return MSS_OK;
  hipLaunchKernelGGL(wino_input_transform_kernel, dim3(grid_for(g.T * (C / 4))), dim3(256), 0, S_(stream), x, ldx, g, C,
                     scale, shift, relu, xt);
  return mss_launch_status();
}

int mss_wino_output_transform_f32(const float* yt, int N, int H, int W, int K, int dil, const float* res, int ldres,
                                  float* y, int ldy, void* stream) {
  if (!yt || !y || K % 4 || ldy % 4 || (res && ldres % 4) || dil < 1) return MSS_ERR_BAD_ARG;
  const WinoGeom g = wino_geom(N, H, W, dil);
  if (g.T == 0) return MSS_OK;
  hipLaunchKernelGGL(wino_output_transform_kernel, dim3(grid_for(g.T * (K / 4))), dim3(256), 0, S_(stream), yt, g, K, res,
                     ldres, y, ldy);
  return mss_launch_status();
}

int mss_wino_grad_output_transform_f32(const float* dy, int lddy, int N, int H, int W, int K, int dil, float* dyt,
                                       void* stream) {
  if (!dy || !dyt || K % 4 || lddy % 4 || dil < 1) return MSS_ERR_BAD_ARG;
  const WinoGeom g = wino_geom(N, H, W, dil);
  if (g.T == 0) return MSS_OK;
  hipLaunchKernelGGL(wino_grad_output_transform_kernel, dim3(grid_for(g.T * (K / 4))), dim3(256), 0, S_(stream), dy,
                     lddy, g, K, dyt);
  return mss_launch_status();
}

int mss_wino_weight_grad_transform_f32(const float* du, float* dw, int K, int C, int Kpad, int Cp, void* stream) {
  if (!du || !dw || Kpad < K || Cp < C) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(wino_weight_grad_transform_kernel, dim3(grid_for((long long)K * C)), dim3(256), 0, S_(stream), du,
                     dw, K, C, Kpad, Cp);
  return mss_launch_status();
}

}  // extern "C"
