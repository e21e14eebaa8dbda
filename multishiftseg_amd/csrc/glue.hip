// HBM-bound pieces of the DeepWV3Plus path that sit between the MFMA convolutions:
// layout change of the input image, BatchNorm statistics / folding / backward, max-pool,
// global-average-pool, broadcast, bilinear (align_corners=True) resampling, Adam.
// All activations are NHWC fp32 with an explicit pixel stride; channels are always a multiple of
// 4 so every access is a 16-byte vector and consecutive lanes touch consecutive channels.
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}

inline int grid_for(long long work_items, int block = 256, int cap = 256 * 16) {
  long long b = (work_items + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---------------------------------------------------------------- image NCHW -> NHWC (padded)
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW,
                                        int Cp) {
  const long long total = (long long)N * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW, p = i - n * HW;
    const float* src = x + n * C * HW + p;
    float* dst = y + i * Cp;
    for (int c = 0; c < Cp; c += 4) {
      f32x4 v;
      v.x = c + 0 < C ? src[(long long)(c + 0) * HW] : 0.f;
      v.y = c + 1 < C ? src[(long long)(c + 1) * HW] : 0.f;
      v.z = c + 2 < C ? src[(long long)(c + 2) * HW] : 0.f;
      v.w = c + 3 < C ? src[(long long)(c + 3) * HW] : 0.f;
      st4(dst + c, v);
    }
  }
}

// The same for feature maps with many channels (the pixel decoder's res2..res5 inputs arrive NCHW from the backbone:
// msdeformattn.py:314-330): a 64-pixel x 64-channel tile through LDS -- 256-byte runs along the pixels on the way in, 256-byte
// runs along the channels on the way out. (The per-pixel loop above writes 16 bytes per lane at a stride of Cp * 4: fine for a
// 3-channel image, 1.8 TB/s on a 256-channel map.) grid (ceil(HW / 64), ceil(Cp / 64), N).
template <bool VEC>
__global__ __launch_bounds__(256) void nchw_to_nhwc_tiled_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW,
                                                                 int Cp) {
  // r04: 16-byte loads along the pixels (4 per thread instead of 16 dword loads; 1.55 TB/s on the 16 x 256 x 176 x 176 maps of the
  // pixel decoder) and the tile kept TRANSPOSED in LDS, [pixel][channel] with stride 65: a lane group's four scalar writes and the
  // four scalar reads of an output quad both hit 64 different banks. VEC: HW % 4 == 0 and a 16-byte aligned source.
  __shared__ float tile[64][65];                 // [pixel][channel]
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const float* src = x + (long long)n * C * HW;
  if (VEC) {
    const int p4 = 4 * (threadIdx.x & 15), cl = threadIdx.x >> 4;      // 16 lanes x 16 B = the 64 pixels of one channel row
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + cl + 16 * j, p = p0 + p4;
      v[j] = (c < C && p < HW) ? ld4(src + (long long)c * HW + p) : f32x4{0.f, 0.f, 0.f, 0.f};      // HW % 4 == 0: p < HW covers p .. p + 3
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[p4 + k][cl + 16 * j] = v[j][k];
  } else {
    const int pl = threadIdx.x & 63, cl = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = c0 + cl + 4 * j, p = p0 + pl;
      tile[pl][cl + 4 * j] = (c < C && p < HW) ? src[(long long)c * HW + p] : 0.f;
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 15, pr = threadIdx.x >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pl = pr + 16 * j, p = p0 + pl, c = c0 + 4 * q;
    if (p < HW && c < Cp)
      st4(y + ((long long)n * HW + p) * Cp + c, f32x4{tile[pl][4 * q], tile[pl][4 * q + 1], tile[pl][4 * q + 2], tile[pl][4 * q + 3]});
  }
}

// ---------------------------------------------------------------- stem: image NCHW -> 3x3 patches, NHWC 32 channels
// mod1.conv1 (3 -> 64, 3x3, padding 1; wider_resnet.py:303) as a GEMM with K = 27: channel j = c*9 + r*3 + s of output
// pixel (y, x) holds img[n][c][y+r-1][x+s-1] (0 outside), channels 27..31 are 0 -- the order of weight.reshape(64, 27).
// The implicit-GEMM kernel needs 16-channel taps, i.e. K = 144 with 13/16 zeros for a 3-channel image (13.5 TFLOP/s);
// this way the MFMA kernel sees a dense 32-deep 1x1 convolution. A workgroup gathers 256 consecutive pixels of a row
// (coalesced 4-byte loads along x), transposes through LDS and writes 32 KB contiguously.
__global__ __launch_bounds__(256) void im2col3x3_c3_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                           int H, int W) {
  __shared__ float tile[256 * 33];
  const int n = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 256, x = x0 + threadIdx.x;
  const float* base = img + (long long)n * 3 * H * W;
  float* mine = tile + threadIdx.x * 33;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y + r - 1;
      const bool oky = yy >= 0 && yy < H;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int xx = x + t - 1;
        mine[c * 9 + r * 3 + t] = (oky && xx >= 0 && xx < W) ? base[((long long)c * H + yy) * W + xx] : 0.f;
      }
    }
#pragma unroll
  for (int j = 27; j < 32; ++j) mine[j] = 0.f;
  __syncthreads();
  const int npx = min(256, W - x0);
  float* dst = out + (((long long)n * H + y) * W + x0) * 32;
  for (int i = threadIdx.x; i < npx * 8; i += 256) {
    const int px = i >> 3, q = (i & 7) * 4;
    const float* src = tile + px * 33 + q;
    st4(dst + (long long)px * 32 + q, f32x4{src[0], src[1], src[2], src[3]});
  }
}

// ---------------------------------------------------------------- per-channel reductions
// Generic column reducer over an NHWC matrix [M][C]: thread (tx, ty) owns channel quad
// q = blockIdx.x*QPB + tx and walks rows ty, ty+RPB, ... of its row range; partial sums are kept
// in fp32 for at most 256 rows, then flushed to fp64, combined over ty through LDS and stored as this
// block's row of a partial matrix part[gridDim.y][2C] (plain stores: no atomics, so the sums are
// bit-reproducible); col_reduce_final_kernel adds the rows in a fixed order.
// F(row, c0, out a[4], out b[4]) produces the two quantities to be summed.
template <typename F>
__device__ __forceinline__ void col_reduce2(long long M, int C, double* __restrict__ accum, F f) {
  double* __restrict__ part = accum + 2 * (size_t)C + (size_t)blockIdx.y * 2 * C;
  const int C4 = C >> 2;
  const int QPB = C4 < 32 ? C4 : 32;            // channel quads per block
  const int RPB = blockDim.x / QPB;             // row lanes per block
  const int tx = threadIdx.x % QPB, ty = threadIdx.x / QPB;
  const int q = blockIdx.x * QPB + tx;
  const bool active = q < C4 && ty < RPB;
  const long long rows_per_block = (M + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  double da[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  if (active) {
    long long r = r0 + ty;
    while (r < r1) {
      float fa[4] = {0, 0, 0, 0}, fb[4] = {0, 0, 0, 0};
      int it = 0;
      // four rows per trip: their loads are independent, so four 16-byte requests per lane are in flight instead of one
      for (; it + 4 <= 256 && r + 3ll * RPB < r1; it += 4, r += 4ll * RPB) {
        float a[4][4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) f(r + (long long)u * RPB, q * 4, a[u], b[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) { fa[k] += a[u][k]; fb[k] += b[u][k]; }
      }
      for (; it < 256 && r < r1; ++it, r += RPB) {
        float a[4], b[4];
        f(r, q * 4, a, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) { fa[k] += a[k]; fb[k] += b[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { da[k] += fa[k]; db[k] += fb[k]; }
    }
  }
  __shared__ double red[256 * 8];
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = da[k]; red[threadIdx.x * 8 + 4 + k] = db[k]; }
  __syncthreads();
  if (ty == 0 && q < C4) {
    for (int yy = 1; yy < RPB; ++yy) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        da[k] += red[(yy * QPB + tx) * 8 + k];
        db[k] += red[(yy * QPB + tx) * 8 + 4 + k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      part[q * 4 + k] = da[k];
      part[C + q * 4 + k] = db[k];
    }
  }
}

// accum[col] = sum over the nparts rows of part (= accum + ncols). Block 256 = 4 columns x 64 row lanes: lane rl adds
// rows rl, rl+64, ... (all its loads issued before the first add: the partials were just written and sit in L2, the cost
// is latency, not bytes), then the 64 lane sums are combined by a fixed-shape tree. The order never depends on
// timing, so the result is bit-reproducible.
__global__ __launch_bounds__(256) void col_reduce_final_kernel(double* __restrict__ accum, int nparts, int ncols) {
  const double* __restrict__ part = accum + ncols;
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int col = blockIdx.x * 4 + cl;
  double s = 0.0;
  if (col < ncols) {
    for (int r0 = rl; r0 < nparts; r0 += 64 * 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + 64 * u;
        v[u] = r < nparts ? part[(size_t)r * ncols + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int half = 32; half >= 1; half >>= 1) {
    if (rl < half) red[threadIdx.x] += red[threadIdx.x + 4 * half];
    __syncthreads();
  }
  if (rl == 0 && col < ncols) accum[col] = red[threadIdx.x];
}

inline dim3 col_reduce_grid(long long M, int C) {
  const int C4 = C >> 2;
  const int QPB = C4 < 32 ? C4 : 32;
  const int gx = (C4 + QPB - 1) / QPB;
  long long gy = 2048 / gx;
  const long long max_gy = (M + 63) / 64;
  if (gy > max_gy) gy = max_gy;
  if (gy < 1) gy = 1;
  return dim3(gx, (unsigned)gy);
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long long M, int C, int ldx,
                                                       double* __restrict__ accum) {
  col_reduce2(M, C, accum, [&](long long r, int c0, float* a, float* b) {
    f32x4 v = ld4(x + r * ldx + c0);
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    b[0] = v.x * v.x; b[1] = v.y * v.y; b[2] = v.z * v.z; b[3] = v.w * v.w;
  });
}

// column sums of a producer's partial-sum matrix [nparts][2][C] (sum | sum of squares) into accum[2C]
__global__ __launch_bounds__(256) void bn_stats_partials_kernel(const float* __restrict__ part, long long nparts, int C,
                                                                double* __restrict__ accum) {
  col_reduce2(nparts, C, accum, [&](long long r, int c0, float* a, float* b) {
    const f32x4 u = ld4(part + r * 2 * C + c0), v = ld4(part + r * 2 * C + C + c0);
    a[0] = u.x; a[1] = u.y; a[2] = u.z; a[3] = u.w;
    b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
  });
}

__global__ void bn_finalize_train_kernel(const double* __restrict__ accum, long long M, int C,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                         float momentum, float* running_mean, float* running_var, float* scale,
                                         float* shift, float* save_mean, float* save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = accum[c] / (double)M;
  double var = accum[C + c] / (double)M - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  if (save_mean) save_mean[c] = (float)mean;
  if (save_invstd) save_invstd[c] = invstd;
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// col_reduce_final_kernel + bn_finalize_train_kernel in one launch (the BatchNorm of a layer whose producer left partial sums: 40
// of the 43 train-mode folds of a step): a workgroup adds the partial rows of TWO channels' sum and sum-of-squares columns in the
// same fixed order and finishes those two channels.
__global__ __launch_bounds__(256) void bn_final_finalize_kernel(double* __restrict__ accum, int nparts, long long M, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps, float momentum, float* running_mean, float* running_var,
                                                                float* scale, float* shift, float* save_mean, float* save_invstd) {
  const double* __restrict__ part = accum + 2 * C;
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int ch = blockIdx.x * 2 + (cl & 1);
  const int col = (cl >> 1) * C + ch;                  // cl 0, 1: the two channels' sums; 2, 3: their sums of squares
  double s = 0.0;
  if (ch < C) {
    for (int r0 = rl; r0 < nparts; r0 += 64 * 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + 64 * u;
        v[u] = r < nparts ? part[(size_t)r * 2 * C + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int half = 32; half >= 1; half >>= 1) {
    if (rl < half) red[threadIdx.x] += red[threadIdx.x + 4 * half];
    __syncthreads();
  }
  if (threadIdx.x < 2 && ch < C) {
    const int c = ch;
    const double sum = red[threadIdx.x], sumsq = red[threadIdx.x + 2];
    accum[c] = sum; accum[C + c] = sumsq;
    const double mean = sum / (double)M;
    double var = sumsq / (double)M - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (running_mean) {
      const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

__global__ void bn_fold_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                    float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * invstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

__global__ void affine_relu_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                   long long M, int C, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int relu) {
  const int C4 = C >> 2;
  const long long total = M * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    f32x4 v = ld4(x + r * ldx + c);
    if (scale) v = v * ld4(scale + c) + ld4(shift + c);
    if (relu) v = relu4(v);
    st4(y + r * ldy + c, v);
  }
}

// backward of y = relu(x*scale + shift) where (scale, shift) fold a train-mode BatchNorm
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(
    const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, long long M, int C,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, int relu, double* __restrict__ accum) {
  col_reduce2(M, C, accum, [&](long long r, int c0, float* a, float* b) {
    f32x4 g = ld4(dy + r * lddy + c0);
    f32x4 v = ld4(x + r * ldx + c0);
    f32x4 sc = ld4(scale + c0), sh = ld4(shift + c0), mu = ld4(save_mean + c0), is = ld4(save_invstd + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float dz = g[k];
      if (relu && !(v[k] * sc[k] + sh[k] > 0.f)) dz = 0.f;
      a[k] = dz;
      b[k] = dz * (v[k] - mu[k]) * is[k];
    }
  });
}

// dx = scale * (dz - mean(dz) - xhat * mean(dz * xhat)), dz = dy * (y > 0). A thread owns one channel quad (its per-channel
// constants are loaded once and folded: dx = A*dz - B - D*(x - mean)) and walks rows; grid (quad groups, row groups). The
// flat-index form spent more instructions on a 64-bit div/mod and 18 constant loads per element than on the data.
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(
    const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, float* __restrict__ dx, int lddx,
    long long M, int C, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, int relu,
    const double* __restrict__ accum, int QPB) {
  const int C4 = C >> 2;
  const int RPB = 256 / QPB;
  const int tx = threadIdx.x % QPB, ty = threadIdx.x / QPB;
  const int q = blockIdx.x * QPB + tx;
  if (q >= C4 || ty >= RPB) return;
  const int c = q * 4;
  const f32x4 sc = ld4(scale + c), sh = ld4(shift + c);
  f32x4 Bc = {0.f, 0.f, 0.f, 0.f}, Dc = {0.f, 0.f, 0.f, 0.f}, mu = {0.f, 0.f, 0.f, 0.f};
  if (accum) {
    const float invM = 1.f / (float)M;
    mu = ld4(save_mean + c);
    const f32x4 is = ld4(save_invstd + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float s1 = (float)accum[c + k], s2 = (float)accum[C + c + k];
      Bc[k] = sc[k] * (s1 * invM);
      Dc[k] = sc[k] * (s2 * invM) * is[k];
    }
  }
  const long long rstep = (long long)gridDim.y * RPB;
  long long r = (long long)blockIdx.y * RPB + ty;
  auto one = [&](const f32x4 g, const f32x4 v) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float dz = g[k];
      if (relu && !(v[k] * sc[k] + sh[k] > 0.f)) dz = 0.f;
      o[k] = sc[k] * dz - Bc[k] - Dc[k] * (v[k] - mu[k]);
    }
    return o;
  };
  for (; r + 3 * rstep < M; r += 4 * rstep) {
    f32x4 g[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { g[u] = ld4(dy + (r + u * rstep) * lddy + c); v[u] = ld4(x + (r + u * rstep) * ldx + c); }
#pragma unroll
    for (int u = 0; u < 4; ++u) st4(dx + (r + u * rstep) * lddx + c, one(g[u], v[u]));
  }
  for (; r < M; r += rstep) st4(dx + r * lddx + c, one(ld4(dy + r * lddy + c), ld4(x + r * ldx + c)));
}

__global__ void bn_param_grad_kernel(const double* __restrict__ accum, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += (float)accum[c];
  if (dgamma) dgamma[c] += (float)accum[C + c];
}

// ---------------------------------------------------------------- pooling / broadcast
__global__ void maxpool3s2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int N,
                                  int H, int W, int C, int OH, int OW) {
  const int C4 = C >> 2;
  const long long total = (long long)N * OH * OW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long p = i / C4;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    const float ninf = -__builtin_huge_valf();
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        f32x4 v = ld4(x + ((long long)(n * H + iy) * W + ix) * ldx + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    st4(y + ((long long)(n * OH + oy) * OW + ox) * ldy + c, m);
  }
}

// ws[split][n][c] = sum_{p in split} x[n][p][c]; grid (C/64, N, splits), block 256 = 16 channel
// quads x 16 row lanes, so a [2,32768,4096] reduction runs on 4096 workgroups instead of 128. The splits are
// combined by colsum_final_kernel in ascending order (no atomics: bit-reproducible).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, float* __restrict__ ws,
                                                     int HW, int C, int rows_per_split) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + tx) * 4;
  const int n = blockIdx.y;
  const int p0 = blockIdx.z * rows_per_split, p1 = min(HW, p0 + rows_per_split);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* base = x + (long long)n * HW * ldx + c;
    for (int p = p0 + ty; p < p1; p += 16) acc += ld4(base + (long long)p * ldx);
  }
  __shared__ f32x4 red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (ty == 0 && c < C) {
    for (int yy = 1; yy < 16; ++yy) acc += red[yy * 16 + tx];
    st4(ws + ((long long)blockIdx.z * gridDim.y + n) * C + c, acc);
  }
}

// y[i] = mult * sum_sp ws[sp][i]: 16 outputs x 16 split lanes per workgroup -- lane l adds the splits l, l + 16, ... in
// ascending order, the 16 lane sums are added in lane order (fixed tree: bit-reproducible). A Linear's bias gradient
// (one image of 162 624 rows, 256 channels) has 512 splits of 256 outputs: one thread per output walked them in 129 us.
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ ws, float* __restrict__ y,
                                                           long long NC, int splits, float mult) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, l = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + o;
  float s = 0.f;
  if (i < NC)
    for (int sp = l; sp < splits; sp += 16) s += ws[(long long)sp * NC + i];
  red[l][o] = s;
  __syncthreads();
  if (l == 0 && i < NC) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    y[i] = t * mult;
  }
}

struct ColsumPlan { int gx, splits, rps; };
inline ColsumPlan colsum_plan(int N, int HW, int C) {
  ColsumPlan pl;
  pl.gx = (C / 4 + 15) / 16;
  int splits = 2048 / (pl.gx * N > 0 ? pl.gx * N : 1);
  const int max_splits = (HW + 127) / 128;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  pl.rps = (HW + splits - 1) / splits;
  pl.splits = (HW + pl.rps - 1) / pl.rps;
  return pl;
}

inline int launch_colsum(const float* x, int ldx, float* y, int N, int HW, int C, float mult, float* ws,
                         hipStream_t st) {
  const ColsumPlan pl = colsum_plan(N, HW, C);
  hipLaunchKernelGGL(colsum_kernel, dim3(pl.gx, N, pl.splits), dim3(256), 0, st, x, ldx, ws, HW, C, pl.rps);
  const long long NC = (long long)N * C;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((NC + 15) / 16)), dim3(256), 0, st, ws, y, NC, pl.splits,
                     mult);
  return mss_launch_status();
}

__global__ void broadcast_rows_kernel(const float* __restrict__ v, float* __restrict__ y, int ldy, int N, int HW,
                                      int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                      int relu) {
  const int C4 = C >> 2;
  const long long total = (long long)N * HW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long long p = i / C4;
    const int n = (int)(p / HW);
    f32x4 val = ld4(v + (long long)n * C + c);
    if (scale) val = val * ld4(scale + c) + ld4(shift + c);
    if (relu) val = relu4(val);
    st4(y + p * ldy + c, val);
  }
}

// ---------------------------------------------------------------- bilinear, align_corners=True
// Same arithmetic as ATen's upsample_bilinear2d (area_pixel_compute_source_index with
// align_corners): src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1-l1.
// (Tap / ac_tap live in mss_common.h: the Winograd input transform of the decoder's first layer interpolates with them too)
inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// grid (ceil(OW * C/4 / 256), OH, N): the row and the image come from the block index, so the only division left is
// one 32-bit divide by C/4 (the 64-bit div/mod chain of a flat index cost more than the four loads)
__global__ __launch_bounds__(256) void upsample_ac_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                          int ldy, int IH, int IW, int OH, int OW, int C, float sh,
                                                          float sw) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  const unsigned ox = item / C4;
  if (ox >= (unsigned)OW) return;
  const int c = (int)(item - ox * C4) * 4;
  const int oy = blockIdx.y, n = blockIdx.z;
  const Tap ty = ac_tap(oy, sh, IH), tx = ac_tap((int)ox, sw, IW);
  const float* b = x + (long long)n * IH * IW * ldx + c;
  const f32x4 v00 = ld4(b + ((long long)ty.i0 * IW + tx.i0) * ldx), v01 = ld4(b + ((long long)ty.i0 * IW + tx.i1) * ldx);
  const f32x4 v10 = ld4(b + ((long long)ty.i1 * IW + tx.i0) * ldx), v11 = ld4(b + ((long long)ty.i1 * IW + tx.i1) * ldx);
  const f32x4 o = mss_bilerp(ty, tx, v00, v01, v10, v11);
  st4(y + ((long long)(n * OH + oy) * OW + ox) * ldy + c, o);
}

// candidate output range [lo, hi] whose source index can touch input cell i
__device__ __forceinline__ void ac_range(int i, float scale, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f) * inv) - 1;
  hi = (int)ceilf(((float)i + 1.f) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float ac_weight(int o, int i, float scale, int in) {
  const Tap t = ac_tap(o, scale, in);
  float w = 0.f;
  if (t.i0 == i) w += t.l0;
  if (t.i1 == i) w += t.l1;
  return w;
}

__global__ void upsample_ac_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                       int N, int IH, int IW, int OH, int OW, int C, float sh, float sw) {
  const int C4 = C >> 2;
  const long long total = (long long)N * IH * IW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long p = i / C4;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int n = (int)(p / IH);
    int ylo, yhi, xlo, xhi;
    ac_range(iy, sh, OH, ylo, yhi);
    ac_range(ix, sw, OW, xlo, xhi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* b = dy + (long long)n * OH * OW * lddy + c;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = ac_weight(oy, iy, sh, IH);
      if (wy == 0.f) continue;
      f32x4 rowacc = {0.f, 0.f, 0.f, 0.f};
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float wx = ac_weight(ox, ix, sw, IW);
        if (wx == 0.f) continue;
        rowacc += wx * ld4(b + ((long long)oy * OW + ox) * lddy);
      }
      acc += wy * rowacc;
    }
    st4(dx + ((long long)(n * IH + iy) * IW + ix) * lddx + c, acc);
  }
}

// The same gather for up-sampling factors up to ~4 (at most MAXR candidate outputs per axis): grid (ceil(IW * C/4 /
// 256), IH, N), the x weights are computed once per thread, and the MAXR loads of an output row are independent and
// issued together (the generic kernel had one load in flight per lane behind ~15 VALU instructions each).
template <int MAXR>
__global__ __launch_bounds__(256) void upsample_ac_bwd_fast_kernel(const float* __restrict__ dy, int lddy,
                                                                   float* __restrict__ dx, int lddx, int IH, int IW,
                                                                   int OH, int OW, int C, float sh, float sw) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  const unsigned ix = item / C4;
  if (ix >= (unsigned)IW) return;
  const int c = (int)(item - ix * C4) * 4;
  const int iy = blockIdx.y, n = blockIdx.z;
  int ylo, yhi, xlo, xhi;
  ac_range(iy, sh, OH, ylo, yhi);
  ac_range((int)ix, sw, OW, xlo, xhi);
  // ac_range is padded by one or two candidates per side for rounding safety: drop the zero-weight ends
  while (xlo < xhi && ac_weight(xlo, (int)ix, sw, IW) == 0.f) ++xlo;
  while (xhi > xlo && ac_weight(xhi, (int)ix, sw, IW) == 0.f) --xhi;
  float wx[MAXR];
  int xo[MAXR];
#pragma unroll
  for (int k = 0; k < MAXR; ++k) {
    const int ox = xlo + k;
    wx[k] = ox <= xhi ? ac_weight(ox, (int)ix, sw, IW) : 0.f;
    xo[k] = ox <= xhi ? ox : -1;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* b = dy + (long long)n * OH * OW * lddy + c;
  for (int oy = ylo; oy <= yhi; ++oy) {
    const float wy = ac_weight(oy, iy, sh, IH);          // block-uniform
    if (wy == 0.f) continue;
    const float* row = b + (long long)oy * OW * lddy;
    f32x4 v[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) v[k] = xo[k] >= 0 ? ld4(row + (long long)xo[k] * lddy) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rowacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXR; ++k) rowacc += wx[k] * v[k];
    acc += wy * rowacc;
  }
  st4(dx + ((long long)(n * IH + iy) * IW + ix) * lddx + c, acc);
}

// ---------------------------------------------------------------- OOD-score tail
// One thread per output pixel: 4 half-resolution neighbours, each a contiguous C-vector.
template <int C>
__global__ __launch_bounds__(256) void ood_score_kernel(const float* __restrict__ dec2, int ld2,
                                                        const float* __restrict__ dec1, int ld1, int N, int IH,
                                                        int IW, int OH, int OW, float sh, float sw,
                                                        float* __restrict__ score, float* __restrict__ logit,
                                                        uint8_t* __restrict__ label) {
  const long long total = (long long)N * OH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long p = i;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    const Tap ty = ac_tap(oy, sh, IH), tx = ac_tap(ox, sw, IW);
    const long long nb = (long long)n * IH * IW;
    const long long o00 = nb + (long long)ty.i0 * IW + tx.i0, o01 = nb + (long long)ty.i0 * IW + tx.i1;
    const long long o10 = nb + (long long)ty.i1 * IW + tx.i0, o11 = nb + (long long)ty.i1 * IW + tx.i1;
    if (score) {
      float e[4];
      const long long offs[4] = {o00, o01, o10, o11};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* q = dec2 + offs[k] * ld2;
        float v[C];
        float m = -__builtin_huge_valf();
#pragma unroll
        for (int c = 0; c < C; ++c) { v[c] = q[c]; m = fmaxf(m, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) s += expf(v[c] - m);
        e[k] = -(m + logf(s));  // energy_func: -logsumexp (deepv3.py:251-253)
      }
      score[i] = ty.l0 * (tx.l0 * e[0] + tx.l1 * e[1]) + ty.l1 * (tx.l0 * e[2] + tx.l1 * e[3]);
    }
    if (logit || label) {
      const float* q00 = dec1 + o00 * ld1; const float* q01 = dec1 + o01 * ld1;
      const float* q10 = dec1 + o10 * ld1; const float* q11 = dec1 + o11 * ld1;
      float best = -__builtin_huge_valf();
      int arg = 0;
      const long long plane = (long long)OH * OW;
      float* lo = logit ? logit + (long long)n * C * plane + (long long)oy * OW + ox : nullptr;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float val = ty.l0 * (tx.l0 * q00[c] + tx.l1 * q01[c]) + ty.l1 * (tx.l0 * q10[c] + tx.l1 * q11[c]);
        if (lo) lo[c * plane] = val;
        if (val > best || (val != val && best == best)) { best = val; arg = c; }  // first max; NaN wins like torch
      }
      if (label) label[i] = (uint8_t)arg;
    }
  }
}

// Tiled version for the up-sampling case (the only one on the path: x2): a workgroup owns a
// TH x TW block of output pixels, stages the few half-resolution rows it needs in LDS with coalesced
// loads (each source value is read from HBM/L2 once per tile instead of ~4x19 scattered 4-byte
// loads per output pixel), computes -logsumexp once per SOURCE pixel, then interpolates from LDS and
// writes score / NCHW logits / labels with 128 consecutive pixels per store instruction.
template <int C, int TH, int TW>
__global__ __launch_bounds__(256) void ood_score_tiled_kernel(const float* __restrict__ dec2, int ld2,
                                                              const float* __restrict__ dec1, int ld1, int IH,
                                                              int IW, int OH, int OW, float sh, float sw, int SRmax,
                                                              int SCmax, float* __restrict__ score,
                                                              float* __restrict__ logit, uint8_t* __restrict__ label) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* en = lds;                       // [SRmax*SCmax] energies
  float* dv = lds + SRmax * SCmax;       // [SRmax*SCmax][C] staged channel vectors
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int oy1 = min(oy0 + TH, OH) - 1, ox1 = min(ox0 + TW, OW) - 1;
  const int r0 = ac_tap(oy0, sh, IH).i0, r1 = ac_tap(oy1, sh, IH).i1;
  const int c0 = ac_tap(ox0, sw, IW).i0, c1 = ac_tap(ox1, sw, IW).i1;
  const int SR = r1 - r0 + 1, SC = c1 - c0 + 1;   // <= SRmax, SCmax by construction
  const int npix = SR * SC;
  const long long nb = (long long)n * IH * IW;
  auto stage = [&](const float* __restrict__ src, int ld) {
    for (int idx = threadIdx.x; idx < npix * C; idx += 256) {
      const int pix = idx / C, c = idx - pix * C;
      const int rr = pix / SC, cc = pix - rr * SC;
      dv[idx] = src[(nb + (long long)(r0 + rr) * IW + (c0 + cc)) * ld + c];
    }
  };
  if (score) {
    stage(dec2, ld2);
    __syncthreads();
    for (int pix = threadIdx.x; pix < npix; pix += 256) {
      const float* q = dv + pix * C;
      float m = -__builtin_huge_valf();
#pragma unroll
      for (int c = 0; c < C; ++c) m = fmaxf(m, q[c]);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) s += expf(q[c] - m);
      en[pix] = -(m + logf(s));
    }
    __syncthreads();
  }
  if (logit || label) {
    stage(dec1, ld1);
    __syncthreads();
  }
  const int tx = threadIdx.x % TW;
  const int ox = ox0 + tx;
  if (ox > ox1) return;
  const Tap tapx = ac_tap(ox, sw, IW);
  const int x0 = tapx.i0 - c0, x1 = tapx.i1 - c0;
  const long long plane = (long long)OH * OW;
  for (int ty = threadIdx.x / TW; ty < TH; ty += 256 / TW) {
    const int oy = oy0 + ty;
    if (oy > oy1) break;
    const Tap tapy = ac_tap(oy, sh, IH);
    const int p00 = (tapy.i0 - r0) * SC + x0, p01 = (tapy.i0 - r0) * SC + x1;
    const int p10 = (tapy.i1 - r0) * SC + x0, p11 = (tapy.i1 - r0) * SC + x1;
    const long long o = (long long)n * plane + (long long)oy * OW + ox;
    if (score)
      score[o] = tapy.l0 * (tapx.l0 * en[p00] + tapx.l1 * en[p01]) + tapy.l1 * (tapx.l0 * en[p10] + tapx.l1 * en[p11]);
    if (logit || label) {
      float best = -__builtin_huge_valf();
      int arg = 0;
      float* lo = logit ? logit + (long long)n * C * plane + (long long)oy * OW + ox : nullptr;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float val = tapy.l0 * (tapx.l0 * dv[p00 * C + c] + tapx.l1 * dv[p01 * C + c]) +
                          tapy.l1 * (tapx.l0 * dv[p10 * C + c] + tapx.l1 * dv[p11 * C + c]);
        if (lo) lo[c * plane] = val;
        if (val > best || (val != val && best == best)) { best = val; arg = c; }
      }
      if (label) label[o] = (uint8_t)arg;
    }
  }
}

// Vector version of the tiled tail for OW % 4 == 0 and a 40-channel-contiguous source (the fused K = 48 heads buffer:
// dec1 = channels 0..18, dec2 = channels 20..38 of the same pixel row): a workgroup owns TH x TW = 8 x 128 output
// pixels; every source pixel of its footprint is read ONCE as ten 16-byte loads (160 contiguous bytes), the dec1 part
// goes to LDS channel-quad-major (slots XOR-swizzled so the stride-2 bilinear taps of neighbouring lanes fall on
// different bank groups), the dec2 part is reduced to -logsumexp on the spot. Each lane then produces 4 consecutive
// output pixels of one row: ds_read_b128 taps, float4 stores on every NCHW class plane (uniform plane base + one 32-bit
// lane offset), one uchar4 label store.
template <int C, int TH, int TW>
__global__ __launch_bounds__(256) void ood_score_v4_kernel(const float* __restrict__ dec, int ld, int c1, int c2,
                                                           int IH, int IW, int OH, int OW, float sh, float sw,
                                                           int SRmax, int SCmax, float* __restrict__ score,
                                                           float* __restrict__ logit, uint8_t* __restrict__ label) {
  constexpr int CQ = (C + 3) / 4;          // channel quads (5 for 19 classes)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int NP = SRmax * SCmax + 16;       // slots per quad plane (swizzle stays inside: p ^ 1 < NP)
  float* en = lds;                         // [NP] energies
  f32x4* sv = reinterpret_cast<f32x4*>(lds + ((NP + 3) & ~3));   // [CQ][NP] float4
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int oy1 = min(oy0 + TH, OH) - 1, ox1 = min(ox0 + TW, OW) - 1;
  const int r0 = ac_tap(oy0, sh, IH).i0, r1 = ac_tap(oy1, sh, IH).i1;
  const int cc0 = ac_tap(ox0, sw, IW).i0, cc1 = ac_tap(ox1, sw, IW).i1;
  const int SR = r1 - r0 + 1, SC = cc1 - cc0 + 1;
  const int npix = SR * SC;
  const long long nb = (long long)n * IH * IW;
  auto slot = [](int p) { return p ^ ((p >> 3) & 1); };
  const bool want_logit = logit || label;
  for (int pix = threadIdx.x; pix < npix; pix += 256) {
    const int rr = pix / SC, cc = pix - rr * SC;
    const float* q = dec + (nb + (long long)(r0 + rr) * IW + (cc0 + cc)) * ld;
    if (want_logit) {
#pragma unroll
      for (int k = 0; k < CQ; ++k) sv[k * NP + slot(pix)] = ld4(q + c1 + 4 * k);
    }
    if (score) {
      float v[CQ * 4];
#pragma unroll
      for (int k = 0; k < CQ; ++k) {
        const f32x4 t = ld4(q + c2 + 4 * k);
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
      }
      float m = v[0];
#pragma unroll
      for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) sum += expf(v[c] - m);
      en[pix] = -(m + logf(sum));          // energy_func: -logsumexp (deepv3.py:251-253)
    }
  }
  __syncthreads();
  const int ty = threadIdx.x / (TW / 4), xq = threadIdx.x % (TW / 4);
  const int oy = oy0 + ty, ox = ox0 + 4 * xq;
  if (oy > oy1 || ox > ox1) return;        // OW % 4 == 0: a quad is inside or outside as a whole
  const Tap tapy = ac_tap(oy, sh, IH);
  const int rowa = (tapy.i0 - r0) * SC - cc0, rowb = (tapy.i1 - r0) * SC - cc0;
  int pa0[4], pa1[4], pb0[4], pb1[4];
  float l0[4], l1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const Tap t = ac_tap(ox + e, sw, IW);
    pa0[e] = rowa + t.i0; pa1[e] = rowa + t.i1; pb0[e] = rowb + t.i0; pb1[e] = rowb + t.i1;
    l0[e] = t.l0; l1[e] = t.l1;
  }
  const long long plane = (long long)OH * OW;
  const long long opix = (long long)oy * OW + ox;
  if (score) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = tapy.l0 * (l0[e] * en[pa0[e]] + l1[e] * en[pa1[e]]) + tapy.l1 * (l0[e] * en[pb0[e]] + l1[e] * en[pb1[e]]);
    st4(score + (long long)n * plane + opix, o);
  }
  if (want_logit) {
    float best[4] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    int arg[4] = {0, 0, 0, 0};
    const unsigned voff = (unsigned)(((long long)n * C * plane + opix) * 4);    // launcher: N*C*OH*OW*4 < 2^32
    char* lbase = reinterpret_cast<char*>(logit);
#pragma unroll
    for (int k = 0; k < CQ; ++k) {
      f32x4 val[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 a0 = sv[k * NP + slot(pa0[e])], a1 = sv[k * NP + slot(pa1[e])];
        const f32x4 b0 = sv[k * NP + slot(pb0[e])], b1 = sv[k * NP + slot(pb1[e])];
        val[e] = tapy.l0 * (l0[e] * a0 + l1[e] * a1) + tapy.l1 * (l0[e] * b0 + l1[e] * b1);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * k + j;
        if (c >= C) break;
        const f32x4 o = {val[0][j], val[1][j], val[2][j], val[3][j]};
        if (logit) *reinterpret_cast<f32x4*>(lbase + (size_t)c * plane * 4 + voff) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (o[e] > best[e] || (o[e] != o[e] && best[e] == best[e])) { best[e] = o[e]; arg[e] = c; }   // first max; NaN wins like torch
      }
    }
    if (label)
      *reinterpret_cast<uint32_t*>(label + (long long)n * plane + opix) =
          (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
  }
}

// backward of the tail: one thread per half-resolution pixel (gather form of the transpose)
template <int C>
__global__ __launch_bounds__(256) void ood_score_bwd_kernel(const float* __restrict__ dec2, int ld2,
                                                            const float* __restrict__ dscore,
                                                            const float* __restrict__ dlogit, int N, int IH,
                                                            int IW, int OH, int OW, float sh, float sw,
                                                            float* __restrict__ ddec2, int ldd2,
                                                            float* __restrict__ ddec1, int ldd1) {
  const long long total = (long long)N * IH * IW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long p = i;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int n = (int)(p / IH);
    int ylo, yhi, xlo, xhi;
    ac_range(iy, sh, OH, ylo, yhi);
    ac_range(ix, sw, OW, xlo, xhi);
    const long long plane = (long long)OH * OW;
    float gs = 0.f;
    float gl[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gl[c] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = ac_weight(oy, iy, sh, IH);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float w = wy * ac_weight(ox, ix, sw, IW);
        if (w == 0.f) continue;
        const long long o = (long long)oy * OW + ox;
        if (dscore) gs += w * dscore[(long long)n * plane + o];
        if (dlogit) {
          const float* q = dlogit + (long long)n * C * plane + o;
#pragma unroll
          for (int c = 0; c < C; ++c) gl[c] += w * q[c * plane];
        }
      }
    }
    if (ddec1) {
      float* o = ddec1 + i * ldd1;
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = gl[c];
    }
    if (ddec2) {
      const float* q = dec2 + i * ld2;
      float v[C];
      float m = -__builtin_huge_valf();
#pragma unroll
      for (int c = 0; c < C; ++c) { v[c] = q[c]; m = fmaxf(m, v[c]); }
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
      const float k = -gs / s;  // d(-lse)/dx_c = -softmax_c
      float* o = ddec2 + i * ldd2;
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = k * v[c];
    }
  }
}

// Tiled backward of the tail for the x2 case (OW % 4 == 0, at most MAXW contributing outputs per axis): a workgroup
// owns TSY x TSX = 4 x 64 half-resolution pixels (one per thread) and walks the 19 class planes of dlogit plus dscore
// as 5 groups of 4 planes: each group's output footprint (~10 rows x 132 columns per plane) is staged in LDS with
// coalesced 16-byte loads, one plane row at a time, instead of ~30 candidate positions x 19 strided 4-byte loads per
// thread (which ran at 1 TB/s). Every thread then writes its 48-channel row of the fused-heads gradient (ddec1 =
// channels c1.., ddec2 = channels c2.., the padding channels as zeros, so the buffer needs no memset).
template <int C, int MAXW>
__global__ __launch_bounds__(256) void ood_score_bwd_tiled_kernel(
    const float* __restrict__ dec2, int ld2, const float* __restrict__ dscore, const float* __restrict__ dlogit, int IH,
    int IW, int OH, int OW, float sh, float sw, float* __restrict__ dd, int ldd, int c1, int c2, int RR, int RC) {
  constexpr int TSY = 4, TSX = 64, G = 4;            // G planes per pass; planes 0..C-1 = dlogit, plane C = dscore
  extern __shared__ __attribute__((aligned(16))) float lds[];     // [G][RR][RC]
  const int n = blockIdx.z, iy0 = blockIdx.y * TSY, ix0 = blockIdx.x * TSX;
  const int tx = threadIdx.x % TSX, tyl = threadIdx.x / TSX;
  const int iy = iy0 + tyl, ix = ix0 + tx;
  const bool live = iy < IH && ix < IW;
  // footprint of the tile in the output (block-uniform)
  int ylo, yhi, xlo, xhi, t0, t1;
  ac_range(iy0, sh, OH, ylo, t1);
  ac_range(min(iy0 + TSY, IH) - 1, sh, OH, t0, yhi);
  ac_range(ix0, sw, OW, xlo, t1);
  ac_range(min(ix0 + TSX, IW) - 1, sw, OW, t0, xhi);
  xlo &= ~3;                                          // 16-byte aligned rows
  const int nrow = yhi - ylo + 1, ncol4 = (xhi - xlo + 4) >> 2;   // <= RR, <= RC / 4 by construction
  // this thread's weights (trimmed to the non-zero ones)
  int oy_a = 0, ox_a = 0, ny = 0, nx = 0;
  float wy[MAXW], wx[MAXW];
#pragma unroll
  for (int k = 0; k < MAXW; ++k) { wy[k] = 0.f; wx[k] = 0.f; }
  if (live) {
    int a, b;
    ac_range(iy, sh, OH, a, b);
    while (a < b && ac_weight(a, iy, sh, IH) == 0.f) ++a;
    while (b > a && ac_weight(b, iy, sh, IH) == 0.f) --b;
    oy_a = a; ny = b - a + 1;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) if (k < ny) wy[k] = ac_weight(a + k, iy, sh, IH);
    ac_range(ix, sw, OW, a, b);
    while (a < b && ac_weight(a, ix, sw, IW) == 0.f) ++a;
    while (b > a && ac_weight(b, ix, sw, IW) == 0.f) --b;
    ox_a = a; nx = b - a + 1;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) if (k < nx) wx[k] = ac_weight(a + k, ix, sw, IW);
  }
  const int ro = oy_a - ylo, co = ox_a - xlo;
  const long long plane = (long long)OH * OW;
  float gl[C + 1];
#pragma unroll
  for (int c = 0; c <= C; ++c) gl[c] = 0.f;
#pragma unroll
  for (int pass = 0; pass < (C + 1 + G - 1) / G; ++pass) {
    __syncthreads();
    // stage G planes x nrow rows x ncol4 float4
    const int per_plane = nrow * ncol4;
    for (int i = threadIdx.x; i < G * per_plane; i += 256) {
      const int g = i / per_plane, rem = i - g * per_plane;
      const int rr = rem / ncol4, c4 = rem - rr * ncol4;
      const int pl = pass * G + g;
      const int ox = xlo + 4 * c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pl <= C && ox < OW) {
        const float* src = pl < C ? (dlogit ? dlogit + ((long long)n * C + pl) * plane : nullptr)
                                  : (dscore ? dscore + (long long)n * plane : nullptr);
        if (src) v = ld4(src + (long long)(ylo + rr) * OW + ox);
      }
      *reinterpret_cast<f32x4*>(&lds[(g * RR + rr) * RC + 4 * c4]) = v;
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int pl = pass * G + g;
        if (pl > C) break;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < MAXW; ++a) {
          if (a >= ny) break;
          const float* row = &lds[(g * RR + ro + a) * RC + co];
          float racc = 0.f;
#pragma unroll
          for (int b = 0; b < MAXW; ++b) if (b < nx) racc += wx[b] * row[b];
          acc += wy[a] * racc;
        }
        gl[pl] = acc;
      }
    }
  }
  if (!live) return;
  const long long spix = ((long long)n * IH + iy) * IW + ix;
  float* o = dd + spix * ldd;
  // ddec1: channels c1 .. c1+C-1 (+ padding up to the next multiple of 4)
  constexpr int CQ = (C + 3) / 4;
#pragma unroll
  for (int k = 0; k < CQ; ++k) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (4 * k + j < C && dlogit) ? gl[4 * k + j] : 0.f;
    st4(o + c1 + 4 * k, v);
  }
  // ddec2 = -gs * softmax(dec2)   (d(-lse)/dx_c = -softmax_c)
  {
    const float* q = dec2 + spix * ld2;
    float v[CQ * 4];
#pragma unroll
    for (int k = 0; k < CQ; ++k) { const f32x4 t = ld4(q + 4 * k); v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
    float m = v[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); sum += v[c]; }
    const float kk = dscore ? -gl[C] / sum : 0.f;
#pragma unroll
    for (int k = 0; k < CQ; ++k) {
      f32x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = 4 * k + j < C ? kk * v[4 * k + j] : 0.f;
      st4(o + c2 + 4 * k, w);
    }
  }
  // channels outside the two heads (the GEMM's padding rows): zeros
  for (int c = 0; c < ldd; c += 4)
    if ((c < c1 || c >= c1 + 4 * CQ) && (c < c2 || c >= c2 + 4 * CQ)) st4(o + c, f32x4{0.f, 0.f, 0.f, 0.f});
}

// ---------------------------------------------------------------- Mask2Former anomaly score
// block = one row segment of 256 pixels of one image; class probabilities of the image's Q
// queries live in LDS as [Q][CP] (CP = C rounded up to 4) and are read as broadcast b128.
template <int CP>
__global__ __launch_bounds__(256) void m2f_score_kernel(const float* __restrict__ cls, const float* __restrict__ mask,
                                                        int Q, int C, int H, int W, int Hm, int Wm,
                                                        float* __restrict__ score) {
  extern __shared__ __attribute__((aligned(16))) float prob[];  // [Q][CP]
  const int b = blockIdx.y;
  for (int q = threadIdx.x; q < Q; q += blockDim.x) {
    const float* row = cls + ((long long)b * Q + q) * (C + 1);
    float m = -__builtin_huge_valf();
    for (int c = 0; c <= C; ++c) m = fmaxf(m, row[c]);
    float s = 0.f;
    for (int c = 0; c <= C; ++c) s += expf(row[c] - m);
    const float inv = 1.f / s;
    for (int c = 0; c < CP; ++c) prob[q * CP + c] = c < C ? expf(row[c] - m) * inv : 0.f;
  }
  __syncthreads();
  const long long HW = (long long)H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW;
       i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i / W), w = (int)(i - (long long)h * W);
    const float* mp = mask + (long long)b * Q * Hm * Wm + (long long)h * Wm + w;
    f32x4 acc[CP / 4];
#pragma unroll
    for (int k = 0; k < CP / 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < Q; ++q) {
      const float x = mp[(long long)q * Hm * Wm];
      const float sg = 1.f / (1.f + expf(-x));
#pragma unroll
      for (int k = 0; k < CP / 4; ++k) acc[k] += sg * *reinterpret_cast<const f32x4*>(&prob[q * CP + 4 * k]);
    }
    float best = -__builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < CP / 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * k + e < C) best = fmaxf(best, acc[k][e]);
    score[(long long)b * HW + i] = 1.f - best;
  }
}

// torch.optim.Adam's single-tensor arithmetic, operation for operation (torch/optim/adam.py _single_tensor_adam, the path
// the reference's CPU run takes): grad += wd*p ; m = lerp(m, grad, 1-b1) ; v = v*b2 + (1-b2)*grad*grad ;
// denom = sqrt(v)/sqrt(bias2) + eps ; p += (-lr/bias1) * (m/denom). step_size and sqrt(bias2) come from the host in double.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float one_minus_b1, float b2, float one_minus_b2,
                            float eps, float wd, float neg_step_size, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float grad = __fmaf_rn(wd, pi, g[i]);                      // grad.add(param, alpha=wd)
    const float m0 = m[i];
    const float mi = __fmaf_rn(one_minus_b1, grad - m0, m0);         // exp_avg.lerp_(grad, 1-b1), weight < 0.5 branch
    const float vi = __fmaf_rn(one_minus_b2 * grad, grad, v[i] * b2); // mul_(b2).addcmul_(grad, grad, value=1-b2)
    m[i] = mi; v[i] = vi;
    const float denom = __fsqrt_rn(vi) / bc2_sqrt + eps;
    p[i] = __fmaf_rn(neg_step_size, mi / denom, pi);                 // addcdiv_(exp_avg, denom, value=-step_size)
  }
}

}  // namespace

#define S_(x) static_cast<hipStream_t>(x)

extern "C" {

int mss_nchw_to_nhwc_pad_f32(const float* x, float* y, int N, int C, int H, int W, int Cp, void* stream) {
  if (!x || !y || Cp % 4 || Cp < C) return MSS_ERR_BAD_ARG;
  const long long HW = (long long)H * W;
  if (C >= 16 && N <= 65535 && (Cp + 63) / 64 <= 65535 && HW < (1ll << 31) && N > 0 && HW > 0) {
    const dim3 grid((unsigned)((HW + 63) / 64), (Cp + 63) / 64, N);
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
      hipLaunchKernelGGL(nchw_to_nhwc_tiled_kernel<true>, grid, dim3(256), 0, S_(stream), x, y, C, (int)HW, Cp);
    else
      hipLaunchKernelGGL(nchw_to_nhwc_tiled_kernel<false>, grid, dim3(256), 0, S_(stream), x, y, C, (int)HW, Cp);
    return mss_launch_status();
  }
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, S_(stream), x, y,
                     N, C, H * W, Cp);
  return mss_launch_status();
}

int mss_im2col3x3_c3_f32(const float* img, float* out, int N, int H, int W, void* stream) {
  if (!img || !out || N <= 0 || H <= 0 || W <= 0 || H > 65535 || N > 65535) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(im2col3x3_c3_kernel, dim3((W + 255) / 256, H, N), dim3(256), 0, S_(stream), img, out, H, W);
  return mss_launch_status();
}

// doubles the `accum` argument of the three per-channel reductions below must hold: [2C] results followed by the
// partial matrix of the first stage (contents irrelevant on entry; nothing has to be zeroed)
long long mss_col_reduce_accum_doubles(long long M, int C) {
  if (M <= 0 || C <= 0) return 2 * (long long)(C > 0 ? C : 0);
  return 2 * (long long)C * (1 + (long long)col_reduce_grid(M, C).y);
}
long long mss_colsum_workspace_floats(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  return (long long)colsum_plan(N, HW, C).splits * N * C;
}

int mss_bn_stats_nhwc_f32(const float* x, long long M, int C, int ldx, double* accum, void* stream) {
  if (!x || !accum || C % 4 || ldx % 4) return MSS_ERR_BAD_ARG;
  if (M <= 0) return MSS_OK;
  const dim3 grid = col_reduce_grid(M, C);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, S_(stream), x, M, C, ldx, accum);
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, S_(stream), accum, (int)grid.y, 2 * C);
  return mss_launch_status();
}

int mss_bn_stats_partials_f32(const float* partials, long long nparts, int C, double* accum, void* stream) {
  if (!partials || !accum || C % 4) return MSS_ERR_BAD_ARG;
  if (nparts <= 0) return MSS_OK;
  const dim3 grid = col_reduce_grid(nparts, C);
  hipLaunchKernelGGL(bn_stats_partials_kernel, grid, dim3(256), 0, S_(stream), partials, nparts, C, accum);
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, S_(stream), accum, (int)grid.y, 2 * C);
  return mss_launch_status();
}

// mss_bn_stats_partials_f32 + mss_bn_finalize_train_f32 in two launches instead of three (same sums in the same order, same
// arithmetic: bit-identical). accum as for mss_bn_stats_partials_f32; M = the number of rows the partial sums cover.
int mss_bn_fold_train_from_partials_f32(const float* partials, long long nparts, int C, double* accum, long long M,
                                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                        float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                                        void* stream) {
  if (!partials || !accum || !scale || !shift || C % 4 || nparts <= 0 || M <= 0) return MSS_ERR_BAD_ARG;
  const dim3 grid = col_reduce_grid(nparts, C);
  hipLaunchKernelGGL(bn_stats_partials_kernel, grid, dim3(256), 0, S_(stream), partials, nparts, C, accum);
  hipLaunchKernelGGL(bn_final_finalize_kernel, dim3((C + 1) / 2), dim3(256), 0, S_(stream), accum, (int)grid.y, M, C, gamma, beta, eps,
                     momentum, running_mean, running_var, scale, shift, save_mean, save_invstd);
  return mss_launch_status();
}

int mss_bn_finalize_train_f32(const double* accum, long long M, int C, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, float* scale,
                              float* shift, float* save_mean, float* save_invstd, void* stream) {
  if (!accum || !scale || !shift || M <= 0) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((C + 255) / 256), dim3(256), 0, S_(stream), accum, M, C, gamma,
                     beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd);
  return mss_launch_status();
}

int mss_bn_fold_eval_f32(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
  if (!running_mean || !running_var || !scale || !shift) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, S_(stream), gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  return mss_launch_status();
}

int mss_affine_relu_nhwc_f32(const float* x, int ldx, float* y, int ldy, long long M, int C, const float* scale,
                             const float* shift, int relu, void* stream) {
  if (!x || !y || C % 4 || ldx % 4 || ldy % 4) return MSS_ERR_BAD_ARG;
  if (M <= 0) return MSS_OK;
  hipLaunchKernelGGL(affine_relu_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, S_(stream), x, ldx, y, ldy, M,
                     C, scale, shift, relu);
  return mss_launch_status();
}

int mss_bn_relu_bwd_reduce_f32(const float* dy, int lddy, const float* x, int ldx, long long M, int C,
                               const float* scale, const float* shift, const float* save_mean,
                               const float* save_invstd, int relu, double* accum, void* stream) {
  if (!dy || !x || !scale || !shift || !save_mean || !save_invstd || !accum) return MSS_ERR_BAD_ARG;
  if (C % 4 || ldx % 4 || lddy % 4) return MSS_ERR_BAD_ARG;
  if (M <= 0) return MSS_OK;
  const dim3 grid = col_reduce_grid(M, C);
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, grid, dim3(256), 0, S_(stream), dy, lddy, x, ldx,
                     M, C, scale, shift, save_mean, save_invstd, relu, accum);
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, S_(stream), accum, (int)grid.y, 2 * C);
  return mss_launch_status();
}

int mss_bn_relu_bwd_apply_f32(const float* dy, int lddy, const float* x, int ldx, float* dx, int lddx,
                              long long M, int C, const float* gamma, const float* scale, const float* shift,
                              const float* save_mean, const float* save_invstd, int relu, const double* accum,
                              float* dgamma, float* dbeta, void* stream) {
  (void)gamma;
  if (!dy || !x || !dx || !scale || !shift) return MSS_ERR_BAD_ARG;
  if (accum && (!save_mean || !save_invstd)) return MSS_ERR_BAD_ARG;
  if (C % 4 || ldx % 4 || lddy % 4 || lddx % 4) return MSS_ERR_BAD_ARG;
  if (M <= 0) return MSS_OK;
  {
    const int C4 = C / 4;
    const int QPB = C4 < 64 ? C4 : 64;
    const int gx = (C4 + QPB - 1) / QPB, RPB = 256 / QPB;
    long long gy = 4096 / gx;                               // ~4096 workgroups
    const long long max_gy = (M + (long long)RPB * 8 - 1) / ((long long)RPB * 8);
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, S_(stream), dy, lddy, x, ldx, dx,
                       lddx, M, C, scale, shift, save_mean, save_invstd, relu, accum, QPB);
  }
  if (accum && (dgamma || dbeta))
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, S_(stream), accum, C, dgamma,
                       dbeta);
  return mss_launch_status();
}

int mss_maxpool3s2_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int OH,
                            int OW, void* stream) {
  if (!x || !y || C % 4 || ldx % 4 || ldy % 4) return MSS_ERR_BAD_ARG;
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for((long long)N * OH * OW * (C / 4))), dim3(256), 0, S_(stream),
                     x, ldx, y, ldy, N, H, W, C, OH, OW);
  return mss_launch_status();
}

// AdaptiveAvgPool2d(1) from the per-64-row column sums the PRODUCING kernel's epilogue left (MssConvArgs.stats: [blocks][2][C], the
// BatchNorm-statistics epilogue of DESIGN 3.4; first half of a block's row = sums): y[n][c] = sum over the image's blocks / HW.
// The 1.07 GB map is not read again for the image-pooling branch (deepv3.py:84-88). 64 channels x 4 block lanes per workgroup,
// float64 accumulation, fixed order.
__global__ __launch_bounds__(256) void gap_from_partials_kernel(const float* __restrict__ part, int blocks_per_image, int C, double inv_hw,
                                                                float* __restrict__ y) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  double s = 0.0;
  if (c < C) {
    const float* p = part + (size_t)n * blocks_per_image * 2 * C + c;
    for (int r0 = rl; r0 < blocks_per_image; r0 += 4 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u;
        v[u] = r < blocks_per_image ? p[(size_t)r * 2 * C] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
  }
  __shared__ double red[4][64];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) y[(size_t)n * C + c] = (float)((((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl]) * inv_hw);
}

int mss_gap_from_partials_f32(const float* partials, int N, int HW, int C, float* y, void* stream) {
  if (!partials || !y || N <= 0 || HW <= 0 || C <= 0) return MSS_ERR_BAD_ARG;
  if (HW % 64) return MSS_ERR_UNSUPPORTED;          // the 64-row blocks of the partial sums would straddle images
  hipLaunchKernelGGL(gap_from_partials_kernel, dim3((C + 63) / 64, N), dim3(256), 0, S_(stream), partials, HW / 64, C, 1.0 / (double)HW, y);
  return mss_launch_status();
}

int mss_gap_nhwc_f32(const float* x, int ldx, float* y, int N, int HW, int C, float* ws, void* stream) {
  if (!x || !y || !ws || C % 4 || ldx % 4 || HW <= 0) return MSS_ERR_BAD_ARG;
  return launch_colsum(x, ldx, y, N, HW, C, 1.f / (float)HW, ws, S_(stream));
}

int mss_colsum_nhwc_f32(const float* dy, int lddy, float* dv, int N, int HW, int C, float* ws, void* stream) {
  if (!dy || !dv || !ws || C % 4 || lddy % 4 || HW <= 0) return MSS_ERR_BAD_ARG;
  return launch_colsum(dy, lddy, dv, N, HW, C, 1.f, ws, S_(stream));
}

int mss_broadcast_rows_nhwc_f32(const float* v, float* y, int ldy, int N, int HW, int C, const float* scale,
                                const float* shift, int relu, void* stream) {
  if (!v || !y || C % 4 || ldy % 4) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(grid_for((long long)N * HW * (C / 4))), dim3(256), 0, S_(stream),
                     v, y, ldy, N, HW, C, scale, shift, relu);
  return mss_launch_status();
}

int mss_upsample_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int IH, int IW, int OH, int OW,
                             int C, void* stream) {
  if (!x || !y || C % 4 || ldx % 4 || ldy % 4) return MSS_ERR_BAD_ARG;
  if (OH > 65535 || N > 65535) return MSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(upsample_ac_kernel, dim3((unsigned)(((long long)OW * (C / 4) + 255) / 256), OH, N), dim3(256), 0,
                     S_(stream), x, ldx, y, ldy, IH, IW, OH, OW, C, ac_scale(IH, OH), ac_scale(IW, OW));
  return mss_launch_status();
}

int mss_upsample_ac_nhwc_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int N, int IH, int IW, int OH,
                                 int OW, int C, void* stream) {
  if (!dy || !dx || C % 4 || lddx % 4 || lddy % 4) return MSS_ERR_BAD_ARG;
  const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
  // candidates per axis: ceil((i+1)/s) + 1 - (floor((i-1)/s) - 1) + 1 <= 2/s + 5
  constexpr int MAXR = 14;
  if (sw > 0.f && 2.f / sw + 5.f <= (float)MAXR && IH <= 65535 && N <= 65535) {
    hipLaunchKernelGGL(upsample_ac_bwd_fast_kernel<MAXR>, dim3((unsigned)(((long long)IW * (C / 4) + 255) / 256), IH, N),
                       dim3(256), 0, S_(stream), dy, lddy, dx, lddx, IH, IW, OH, OW, C, sh, sw);
    return mss_launch_status();
  }
  hipLaunchKernelGGL(upsample_ac_bwd_kernel, dim3(grid_for((long long)N * IH * IW * (C / 4))), dim3(256), 0,
                     S_(stream), dy, lddy, dx, lddx, N, IH, IW, OH, OW, C, sh, sw);
  return mss_launch_status();
}

int mss_ood_score_f32(const float* dec2, int ld2, const float* dec1, int ld1, int N, int IH, int IW, int C,
                      int OH, int OW, float* score, float* logit_nchw, uint8_t* label, void* stream) {
  if (C != 19) return MSS_ERR_UNSUPPORTED;  // Cityscapes, the only class count on the path
  if (score && !dec2) return MSS_ERR_BAD_ARG;
  if ((logit_nchw || label) && !dec1) return MSS_ERR_BAD_ARG;
  if ((long long)N * OH * OW == 0) return MSS_OK;
  {
    constexpr int TH = 8, TW = 128;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    const int SRmax = (int)((TH - 1) * sh) + 3, SCmax = (int)((TW - 1) * sw) + 3;
    // both heads in one pixel-contiguous buffer (the K = 48 fused heads GEMM output), 16-byte aligned everywhere
    const float* base = dec1 ? (dec2 ? (dec1 < dec2 ? dec1 : dec2) : dec1) : dec2;
    const long long off1 = dec1 ? dec1 - base : 0, off2 = dec2 ? dec2 - base : 0;
    const int ld = dec1 ? ld1 : ld2;
    const int NP = SRmax * SCmax + 16;
    const size_t smem4 = ((size_t)((NP + 3) & ~3) + (size_t)5 * NP * 4) * sizeof(float);
    const bool v4 = OW % 4 == 0 && (!dec1 || !dec2 || ld1 == ld2) && ld % 4 == 0 && off1 % 4 == 0 && off2 % 4 == 0 &&
                    off1 + 20 <= ld && off2 + 20 <= ld && smem4 <= 64 * 1024 && N <= 65535 &&
                    (long long)N * 19 * OH * OW * 4 < (1ll << 32) &&
                    ((reinterpret_cast<uintptr_t>(base) | reinterpret_cast<uintptr_t>(score) |
                      reinterpret_cast<uintptr_t>(logit_nchw) | reinterpret_cast<uintptr_t>(label)) & 15) == 0;
    if (v4) {
      hipLaunchKernelGGL((ood_score_v4_kernel<19, TH, TW>), dim3((OW + TW - 1) / TW, (OH + TH - 1) / TH, N), dim3(256),
                         smem4, S_(stream), base, ld, (int)off1, (int)off2, IH, IW, OH, OW, sh, sw, SRmax, SCmax, score,
                         logit_nchw, label);
      return mss_launch_status();
    }
    const size_t smem = (size_t)SRmax * SCmax * (19 + 1) * sizeof(float);
    if (smem <= 48 * 1024 && N <= 65535) {   // up-sampling: the source footprint of a tile is small
      hipLaunchKernelGGL((ood_score_tiled_kernel<19, TH, TW>), dim3((OW + TW - 1) / TW, (OH + TH - 1) / TH, N),
                         dim3(256), smem, S_(stream), dec2, ld2, dec1, ld1, IH, IW, OH, OW, sh, sw, SRmax, SCmax, score,
                         logit_nchw, label);
      return mss_launch_status();
    }
  }
  hipLaunchKernelGGL(ood_score_kernel<19>, dim3(grid_for((long long)N * OH * OW, 256, 1 << 20)), dim3(256), 0,
                     S_(stream), dec2, ld2, dec1, ld1, N, IH, IW, OH, OW, ac_scale(IH, OH), ac_scale(IW, OW), score,
                     logit_nchw, label);
  return mss_launch_status();
}

int mss_ood_score_bwd_f32(const float* dec2, int ld2, const float* dscore, const float* dlogit_nchw, int N,
                          int IH, int IW, int C, int OH, int OW, float* ddec2, int ldd2, float* ddec1, int ldd1,
                          void* stream) {
  if (C != 19) return MSS_ERR_UNSUPPORTED;
  if (ddec2 && !dec2) return MSS_ERR_BAD_ARG;
  if ((long long)N * IH * IW == 0) return MSS_OK;
  {
    // tiled path: both heads' gradients go into one pixel-contiguous buffer (the 48-wide input of the fused-heads dgrad)
    constexpr int MAXW = 6, TSY = 4, TSX = 64;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    const bool up = sh > 0.f && sw > 0.f && (int)(2.f / sh) + 2 <= MAXW && (int)(2.f / sw) + 2 <= MAXW;
    if (up && ddec1 && ddec2 && ldd1 == ldd2 && OW % 4 == 0 && ldd1 % 4 == 0 && ld2 % 4 == 0 && N <= 65535) {
      float* base = ddec1 < ddec2 ? ddec1 : ddec2;
      const long long c1 = ddec1 - base, c2 = ddec2 - base;
      const int RR = (int)((TSY + 1) / sh) + 6, RC = ((int)((TSX + 1) / sw) + 10 + 3) & ~3;
      const size_t smem = (size_t)4 * RR * RC * sizeof(float);
      const bool ok = c1 % 4 == 0 && c2 % 4 == 0 && c1 + 20 <= ldd1 && c2 + 20 <= ldd1 && (c1 + 20 <= c2 || c2 + 20 <= c1) &&
                      smem <= 64 * 1024 &&
                      ((reinterpret_cast<uintptr_t>(base) | reinterpret_cast<uintptr_t>(dec2) | reinterpret_cast<uintptr_t>(dscore) |
                        reinterpret_cast<uintptr_t>(dlogit_nchw)) & 15) == 0;
      if (ok) {
        hipLaunchKernelGGL((ood_score_bwd_tiled_kernel<19, MAXW>), dim3((IW + TSX - 1) / TSX, (IH + TSY - 1) / TSY, N),
                           dim3(256), smem, S_(stream), dec2, ld2, dscore, dlogit_nchw, IH, IW, OH, OW, sh, sw, base, ldd1,
                           (int)c1, (int)c2, RR, RC);
        return mss_launch_status();
      }
    }
  }
  hipLaunchKernelGGL(ood_score_bwd_kernel<19>, dim3(grid_for((long long)N * IH * IW, 256, 1 << 20)), dim3(256), 0,
                     S_(stream), dec2, ld2, ddec2 ? dscore : nullptr, ddec1 ? dlogit_nchw : nullptr, N, IH, IW, OH,
                     OW, ac_scale(IH, OH), ac_scale(IW, OW), ddec2, ldd2, ddec1, ldd1);
  return mss_launch_status();
}

int mss_m2f_score_f32(const float* cls, const float* mask, int B, int Q, int C, int H, int W, int Hm, int Wm,
                      float* score, void* stream) {
  if (!cls || !mask || !score || H > Hm || W > Wm) return MSS_ERR_BAD_ARG;
  if (C > 20 || C < 1) return MSS_ERR_UNSUPPORTED;
  if ((long long)B * H * W == 0) return MSS_OK;
  constexpr int CP = 20;
  const size_t smem = (size_t)Q * CP * sizeof(float);
  if (smem > 65536) return MSS_ERR_UNSUPPORTED;
  const int gx = grid_for((long long)H * W, 256, 1024);
  hipLaunchKernelGGL(m2f_score_kernel<CP>, dim3(gx, B), dim3(256), smem, S_(stream), cls, mask, Q, C, H, W, Hm, Wm,
                     score);
  return mss_launch_status();
}

static std::atomic<int> g_env_generation{0};
int mss_env_generation(void) { return g_env_generation.load(std::memory_order_relaxed); }
int mss_env_reset(void) { return g_env_generation.fetch_add(1, std::memory_order_relaxed) + 1; }

int mss_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                      double beta1, double beta2, double eps, double weight_decay, int step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return MSS_ERR_BAD_ARG;
  if (n <= 0) return MSS_OK;
  // bias corrections in double, as Python computes them (1 - beta**step); only the two derived scalars are rounded to float
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float neg_step_size = (float)(-(lr / bc1));
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), param, grad, exp_avg, exp_avg_sq, n,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                     neg_step_size, bc2_sqrt);
  return mss_launch_status();
}

}  // extern "C"
