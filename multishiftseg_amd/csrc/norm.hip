// Normalisation / resampling glue of the Mask2Former pixel decoder, NHWC fp32:
//   * residual add + LayerNorm in one pass, forward and backward (msdeformattn.py:116-131: norm1(src + attn), norm2(src +
//     ffn)) -- one wave per token row, statistics by DPP/shuffle sums, no LDS;
//   * GroupNorm(32, C) of the input projections and FPN convs (msdeformattn.py:215-219,262-281): per-(sample, group)
//     statistics in two deterministic stages (no atomics), then a streaming apply (+ReLU) that can write straight into
//     the encoder's token buffer [N, sum(HW), C] (a per-sample output stride);
//   * bilinear align_corners=False up-sampling fused with the lateral add of the FPN top-down path (msdeformattn.py:344);
//   * NHWC -> NCHW for the module boundary (the reference returns NCHW maps).
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ------------------------------------------------------------------------------------------ LayerNorm
// y = LN(x + res) * gamma + beta over the last dimension C = 4 * 64 * Q (Q float4 per lane); one wave per row.
template <int Q>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            long long rows, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ y, float* __restrict__ stat,
                                                            const float* __restrict__ pos = nullptr, long long pos_rows = 0,
                                                            float* __restrict__ qout = nullptr) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[Q];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int c = (lane + 64 * k) * 4;
    v[k] = ld4(x + row * C + c);
    if (res) v[k] += ld4(res + row * C + c);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = mss_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const f32x4 d = v[k] - mean;
    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
  const float rstd = rsqrtf(mss_wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int c = (lane + 64 * k) * 4;
    const f32x4 yv = (v[k] - mean) * rstd * ld4(gamma + c) + ld4(beta + c);
    st4(y + row * C + c, yv);
    // r04: the NEXT encoder layer's query q = y + pos (msdeformattn.py:116: with_pos_embed) leaves with the same pass; pos has
    // pos_rows rows (one image's tokens when it is shared by the batch)
    if (qout) st4(qout + row * C + c, yv + ld4(pos + (row % pos_rows) * C + c));
  }
  if (stat && lane == 0) { stat[2 * row] = mean; stat[2 * row + 1] = rstd; }
}

// backward: dz = rstd * (g*gamma - mean_C(g*gamma) - xhat * mean_C(g*gamma*xhat)), the same for x and res; per-workgroup
// partial sums of (g*xhat | g) per channel go to part[block][2][C] and are added in block order by ln_param_grad_kernel.
// SUM: the workgroup also leaves the per-channel sums of dz (part[block][3][C]): the bias gradient of the Linear that produced
// `res` (d bias = column sums of the gradient of its output = dz), without another pass over the [rows, C] tensor.
template <int Q, bool SUM = false>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                const float* __restrict__ res, const float* __restrict__ stat,
                                                                long long rows, int C, const float* __restrict__ gamma,
                                                                float* __restrict__ dz, float* __restrict__ part,
                                                                int rows_per_block, const float* __restrict__ gy2 = nullptr) {
  constexpr int NP = SUM ? 3 : 2;
  __shared__ f32x4 red[NP][4][64 * Q];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 dg[Q], db[Q], gm[Q], dzs[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    dg[k] = f32x4{0.f, 0.f, 0.f, 0.f}; db[k] = dg[k]; dzs[k] = dg[k];
    gm[k] = ld4(gamma + (lane + 64 * k) * 4);
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (long long row = r0 + wave; row < r1; row += 4) {
    const float mean = stat[2 * row], rstd = stat[2 * row + 1];
    f32x4 g[Q], xh[Q];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int c = (lane + 64 * k) * 4;
      f32x4 z = ld4(x + row * C + c);
      if (res) z += ld4(res + row * C + c);
      g[k] = ld4(gy + row * C + c);
      if (gy2) g[k] += ld4(gy2 + row * C + c);          // a second consumer of y (the next layer's query): gradients summed on load
      xh[k] = (z - mean) * rstd;
      const f32x4 gg = g[k] * gm[k];
      s1 += (gg.x + gg.y) + (gg.z + gg.w);
      const f32x4 t = gg * xh[k];
      s2 += (t.x + t.y) + (t.z + t.w);
      dg[k] += g[k] * xh[k];
      db[k] += g[k];
    }
    const float m1 = mss_wave_sum(s1) / (float)C, m2 = mss_wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int c = (lane + 64 * k) * 4;
      const f32x4 d = rstd * (g[k] * gm[k] - m1 - xh[k] * m2);
      st4(dz + row * C + c, d);
      if (SUM) dzs[k] += d;
    }
  }
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    red[0][wave][lane + 64 * k] = dg[k]; red[1][wave][lane + 64 * k] = db[k];
    if (SUM) red[NP - 1][wave][lane + 64 * k] = dzs[k];
  }
  __syncthreads();
  if (wave == 0) {
    float* o = part + (size_t)blockIdx.x * NP * C;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int i = lane + 64 * k;
      st4(o + i * 4, ((red[0][0][i] + red[0][1][i]) + red[0][2][i]) + red[0][3][i]);
      st4(o + C + i * 4, ((red[1][0][i] + red[1][1][i]) + red[1][2][i]) + red[1][3][i]);
      if (SUM) st4(o + 2 * C + i * 4, ((red[NP - 1][0][i] + red[NP - 1][1][i]) + red[NP - 1][2][i]) + red[NP - 1][3][i]);
    }
  }
}

// out[col] = sum over blocks (ascending) of part[block][col]; 4 columns x 64 row lanes per workgroup, fixed-shape tree
__global__ __launch_bounds__(256) void ordered_colsum_kernel(const float* __restrict__ part, int nparts, int ncols,
                                                             long long row_stride, float* __restrict__ out) {
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int col = blockIdx.x * 4 + cl;
  float s = 0.f;
  if (col < ncols)
    for (int r = rl; r < nparts; r += 64) s += part[(size_t)r * row_stride + col];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int half = 32; half >= 1; half >>= 1) {
    if (rl < half) red[threadIdx.x] += red[threadIdx.x + 4 * half];
    __syncthreads();
  }
  if (rl == 0 && col < ncols) out[col] = red[threadIdx.x];
}

// ------------------------------------------------------------------------------------------ GroupNorm
// stage 1: grid (chunks, N); lane = channel quad (C/4 <= 256 quads over the 256 threads' x dimension), rows walked by
// the remaining threads; partial (sum, sumsq) per quad -> part[n][chunk][2][C/4]
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int ldx, long long sample_stride,
                                                       int HW, int C, float* __restrict__ part, int rows_per_chunk) {
  const int C4 = C >> 2;
  const int QPB = C4 < 256 ? C4 : 256, RPB = 256 / QPB;
  const int tx = threadIdx.x % QPB, ty = threadIdx.x / QPB;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
  __shared__ float red[2][256];
  float s = 0.f, q = 0.f;
  if (ty < RPB) {
    const float* b = x + (long long)n * sample_stride + tx * 4;
    for (int p = p0 + ty; p < p1; p += RPB) {
      const f32x4 v = ld4(b + (long long)p * ldx);
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
  __syncthreads();
  if (ty == 0) {
    for (int yy = 1; yy < RPB; ++yy) { s += red[0][yy * QPB + tx]; q += red[1][yy * QPB + tx]; }
    float* o = part + (((size_t)n * gridDim.x + blockIdx.x) * 2) * C4;
    o[tx] = s; o[C4 + tx] = q;
  }
}

// stage 2: one WAVE per (n, group): lane l adds chunks l, l + 64, ... (and the group's quads) in order, then a fixed-shape shuffle
// tree combines the 64 lane sums -- all in double, the order never depends on timing. (One THREAD per (n, group) walked up to a
// thousand chunks serially: 188 us for the 256 x 512 map of one 1024 x 2048 image.)
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int N, int nchunks, int C, int groups, int HW,
                                                         float eps, float* __restrict__ stat) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= N * groups) return;
  const int n = i / groups, g = i - n * groups;
  const int C4 = C >> 2, qpg = (C / groups) >> 2;      // quads per group
  double s = 0.0, q = 0.0;
  for (int ch = lane; ch < nchunks; ch += 64) {
    const float* o = part + (((size_t)n * nchunks + ch) * 2) * C4;
    for (int k = 0; k < qpg; ++k) { s += o[g * qpg + k]; q += o[C4 + g * qpg + k]; }
  }
  s = mss_wave_sum_d(s);
  q = mss_wave_sum_d(q);
  if (lane != 0) return;
  const double cnt = (double)HW * (C / groups);
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0) var = 0;
  stat[2 * i] = (float)mean;
  stat[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// apply: grid (ceil(HW * C/4 / 256), N)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ldx, long long x_sample_stride,
                                                       int HW, int C, int groups, const float* __restrict__ stat,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int relu, float* __restrict__ y, int ldy, long long y_sample_stride) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  const unsigned p = item / C4;
  if (p >= (unsigned)HW) return;
  const int c = (int)(item - p * C4) * 4;
  const int n = blockIdx.y;
  const int g = c / (C / groups);                       // C/groups is a multiple of 4: a quad never straddles groups
  const float mean = stat[2 * (n * groups + g)], rstd = stat[2 * (n * groups + g) + 1];
  f32x4 v = (ld4(x + (long long)n * x_sample_stride + (long long)p * ldx + c) - mean) * rstd * ld4(gamma + c) + ld4(beta + c);
  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  st4(y + (long long)n * y_sample_stride + (long long)p * ldy + c, v);
}

// ------------------------------------------------------------------------------------------ FPN top-down add
// y = lat + bilinear(top -> OH x OW), align_corners=False (half-pixel centres, source index clamped at 0 as ATen's
// area_pixel_compute_source_index does); grid (ceil(OW * C/4 / 256), OH, N)
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ top, int ldt, long long top_ss, int IH,
                                                           int IW, const float* __restrict__ lat, int ldl,
                                                           float* __restrict__ y, int ldy, int OH, int OW, int C, float sh,
                                                           float sw) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  const unsigned ox = item / C4;
  if (ox >= (unsigned)OW) return;
  const int c = (int)(item - ox * C4) * 4;
  const int oy = blockIdx.y, n = blockIdx.z;
  auto tap = [](int o, float scale, int in, int& i0, int& i1, float& l1) {
    float src = ((float)o + 0.5f) * scale;
    asm volatile("" : "+v"(src));                       // ATen rounds the product before subtracting 0.5
    src -= 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
  };
  int y0, y1, x0, x1;
  float ly, lx;
  tap(oy, sh, IH, y0, y1, ly);
  tap((int)ox, sw, IW, x0, x1, lx);
  const float* b = top + (long long)n * top_ss + c;
  const f32x4 v00 = ld4(b + ((long long)y0 * IW + x0) * ldt), v01 = ld4(b + ((long long)y0 * IW + x1) * ldt);
  const f32x4 v10 = ld4(b + ((long long)y1 * IW + x0) * ldt), v11 = ld4(b + ((long long)y1 * IW + x1) * ldt);
  const long long o = ((long long)(n * OH + oy) * OW + ox);
  const f32x4 up = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  st4(y + o * ldy + c, ld4(lat + o * ldl + c) + up);
}

// ------------------------------------------------------------------------------------------ NHWC -> NCHW
// 64 pixels x 64 channels per workgroup through LDS; grid (ceil(HW/64), ceil(C/64), N)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int ldx, long long x_ss, int HW,
                                                           int C, float* __restrict__ y) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? x[(long long)n * x_ss + (long long)p * ldx + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) y[((long long)n * C + c) * HW + p] = tile[tx][r];
  }
}

}  // namespace

#define S_(x) static_cast<hipStream_t>(x)

extern "C" {

int mss_add_layernorm_f32(const float* x, const float* res, long long rows, int C, const float* gamma, const float* beta,
                          float eps, float* y, float* stat, void* stream) {
  if (!x || !gamma || !beta || !y || rows < 0) return MSS_ERR_BAD_ARG;
  if (rows == 0) return MSS_OK;
  if (C % 256 || C > 1024) return MSS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((rows + 3) / 4));
  switch (C / 256) {
    case 1: hipLaunchKernelGGL(add_layernorm_kernel<1>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat); break;
    case 2: hipLaunchKernelGGL(add_layernorm_kernel<2>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat); break;
    case 3: hipLaunchKernelGGL(add_layernorm_kernel<3>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat); break;
    default: hipLaunchKernelGGL(add_layernorm_kernel<4>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat); break;
  }
  return mss_launch_status();
}

// The same, also writing q = y + pos[row % pos_rows] (the query of the next MSDeformAttn encoder layer, msdeformattn.py:116-118)
int mss_add_layernorm_q_f32(const float* x, const float* res, long long rows, int C, const float* gamma, const float* beta,
                            float eps, float* y, float* stat, const float* pos, long long pos_rows, float* q, void* stream) {
  if (!x || !gamma || !beta || !y || rows < 0 || !pos || !q || pos_rows <= 0) return MSS_ERR_BAD_ARG;
  if (rows == 0) return MSS_OK;
  if (C % 256 || C > 1024) return MSS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((rows + 3) / 4));
  switch (C / 256) {
    case 1: hipLaunchKernelGGL(add_layernorm_kernel<1>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat, pos, pos_rows, q); break;
    case 2: hipLaunchKernelGGL(add_layernorm_kernel<2>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat, pos, pos_rows, q); break;
    case 3: hipLaunchKernelGGL(add_layernorm_kernel<3>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat, pos, pos_rows, q); break;
    default: hipLaunchKernelGGL(add_layernorm_kernel<4>, grid, dim3(256), 0, S_(stream), x, res, rows, C, gamma, beta, eps, y, stat, pos, pos_rows, q); break;
  }
  return mss_launch_status();
}

// workspace floats for the backward's per-workgroup partial sums of (dgamma | dbeta)
long long mss_add_layernorm_bwd_workspace_floats(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  long long blocks = (rows + 63) / 64;
  if (blocks > 1024) blocks = 1024;
  return blocks * 2 * C;
}

}  // extern "C"

namespace {
template <bool SUM>
int launch_add_layernorm_bwd(const float* gy, const float* x, const float* res, const float* stat, long long rows, int C,
                             const float* gamma, float* dz, float* dgamma, float* dbeta, float* dzsum, float* ws, void* stream,
                             const float* gy2 = nullptr) {
  if (!gy || !x || !stat || !gamma || !dz || !ws || rows < 0) return MSS_ERR_BAD_ARG;
  if (rows == 0) return MSS_OK;
  if (C % 256 || C > 1024) return MSS_ERR_UNSUPPORTED;
  long long blocks = (rows + 63) / 64;
  if (blocks > 1024) blocks = 1024;
  const int rpb = (int)((rows + blocks - 1) / blocks);
  blocks = (rows + rpb - 1) / rpb;
  const dim3 grid((unsigned)blocks);
  switch (C / 256) {
    case 1: hipLaunchKernelGGL((add_layernorm_bwd_kernel<1, SUM>), grid, dim3(256), 0, S_(stream), gy, x, res, stat, rows, C, gamma, dz, ws, rpb, gy2); break;
    case 2: hipLaunchKernelGGL((add_layernorm_bwd_kernel<2, SUM>), grid, dim3(256), 0, S_(stream), gy, x, res, stat, rows, C, gamma, dz, ws, rpb, gy2); break;
    case 3: hipLaunchKernelGGL((add_layernorm_bwd_kernel<3, SUM>), grid, dim3(256), 0, S_(stream), gy, x, res, stat, rows, C, gamma, dz, ws, rpb, gy2); break;
    default: hipLaunchKernelGGL((add_layernorm_bwd_kernel<4, SUM>), grid, dim3(256), 0, S_(stream), gy, x, res, stat, rows, C, gamma, dz, ws, rpb, gy2); break;
  }
  // part is [block][NP][C]: rows of dgamma at stride NP*C from offset 0, rows of dbeta from offset C, of sum(dz) from 2C
  const long long st = (SUM ? 3ll : 2ll) * C;
  if (dgamma) hipLaunchKernelGGL(ordered_colsum_kernel, dim3((C + 3) / 4), dim3(256), 0, S_(stream), ws, (int)blocks, C, st, dgamma);
  if (dbeta) hipLaunchKernelGGL(ordered_colsum_kernel, dim3((C + 3) / 4), dim3(256), 0, S_(stream), ws + C, (int)blocks, C, st, dbeta);
  if (SUM && dzsum) hipLaunchKernelGGL(ordered_colsum_kernel, dim3((C + 3) / 4), dim3(256), 0, S_(stream), ws + 2 * C, (int)blocks, C, st, dzsum);
  return mss_launch_status();
}
}  // namespace

extern "C" {

int mss_add_layernorm_bwd_f32(const float* gy, const float* x, const float* res, const float* stat, long long rows, int C,
                              const float* gamma, float* dz, float* dgamma, float* dbeta, float* ws, void* stream) {
  return launch_add_layernorm_bwd<false>(gy, x, res, stat, rows, C, gamma, dz, dgamma, dbeta, nullptr, ws, stream);
}

// the same with dzsum [C] = per-channel sums of dz (ws: 3/2 of mss_add_layernorm_bwd_workspace_floats)
int mss_add_layernorm_bwd_sum_f32(const float* gy, const float* x, const float* res, const float* stat, long long rows, int C,
                                  const float* gamma, float* dz, float* dgamma, float* dbeta, float* dzsum, float* ws, void* stream) {
  if (!dzsum) return MSS_ERR_BAD_ARG;
  return launch_add_layernorm_bwd<true>(gy, x, res, stat, rows, C, gamma, dz, dgamma, dbeta, dzsum, ws, stream);
}

// the same for an output with TWO consumers: the gradient is gy + gy2 (gy2 may be NULL), added while it is loaded
int mss_add_layernorm_bwd_sum2_f32(const float* gy, const float* gy2, const float* x, const float* res, const float* stat, long long rows,
                                   int C, const float* gamma, float* dz, float* dgamma, float* dbeta, float* dzsum, float* ws,
                                   void* stream) {
  if (!dzsum) return MSS_ERR_BAD_ARG;
  return launch_add_layernorm_bwd<true>(gy, x, res, stat, rows, C, gamma, dz, dgamma, dbeta, dzsum, ws, stream, gy2);
}

// GroupNorm over NHWC x [N][HW][C] (pixel stride ldx, sample stride x_sample_stride floats): y = (x - mean[n,g]) *
// rstd[n,g] * gamma[c] + beta[c] (+ReLU) written with its own pixel / sample strides. ws: mss_groupnorm_workspace_floats.
long long mss_groupnorm_workspace_floats(int N, int HW, int C, int groups) {
  if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
  long long chunks = (HW + 255) / 256;
  if (chunks > 512) chunks = 512;
  return (long long)N * chunks * 2 * (C / 4) + 2ll * N * groups;
}

int mss_groupnorm_nhwc_f32(const float* x, int ldx, long long x_sample_stride, int N, int HW, int C, int groups,
                           const float* gamma, const float* beta, float eps, int relu, float* y, int ldy,
                           long long y_sample_stride, float* ws, void* stream) {
  if (!x || !gamma || !beta || !y || !ws || N < 0 || HW <= 0) return MSS_ERR_BAD_ARG;
  if (N == 0) return MSS_OK;
  if (C % 4 || ldx % 4 || ldy % 4 || groups <= 0 || C % groups || (C / groups) % 4 || C / 4 > 256 || N > 65535)
    return MSS_ERR_UNSUPPORTED;
  long long chunks = (HW + 255) / 256;
  if (chunks > 512) chunks = 512;
  const int rpc = (int)((HW + chunks - 1) / chunks);
  chunks = (HW + rpc - 1) / rpc;
  float* part = ws;
  float* stat = ws + (size_t)N * chunks * 2 * (C / 4);
  hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)chunks, N), dim3(256), 0, S_(stream), x, ldx, x_sample_stride, HW, C, part, rpc);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * groups), dim3(64), 0, S_(stream), part, N, (int)chunks, C, groups, HW, eps, stat);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)(((long long)HW * (C / 4) + 255) / 256), N), dim3(256), 0, S_(stream), x, ldx,
                     x_sample_stride, HW, C, groups, stat, gamma, beta, relu, y, ldy, y_sample_stride);
  return mss_launch_status();
}

// y = lat + bilinear(top [N,IH,IW,C] -> OH x OW, align_corners=False); all NHWC with their own pixel strides
int mss_upsample_bilinear_add_nhwc_f32(const float* top, int ldt, long long top_sample_stride, int N, int IH, int IW,
                                       const float* lat, int ldl, float* y, int ldy, int OH, int OW, int C, void* stream) {
  if (!top || !lat || !y || C % 4 || ldt % 4 || ldl % 4 || ldy % 4) return MSS_ERR_BAD_ARG;
  if ((long long)N * OH * OW == 0) return MSS_OK;
  if (OH > 65535 || N > 65535) return MSS_ERR_UNSUPPORTED;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;     // area_pixel_compute_scale, align_corners=False
  hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)(((long long)OW * (C / 4) + 255) / 256), OH, N), dim3(256), 0,
                     S_(stream), top, ldt, top_sample_stride, IH, IW, lat, ldl, y, ldy, OH, OW, C, sh, sw);
  return mss_launch_status();
}

int mss_nhwc_to_nchw_f32(const float* x, int ldx, long long x_sample_stride, int N, int HW, int C, float* y, void* stream) {
  if (!x || !y || N < 0 || HW <= 0 || C <= 0) return MSS_ERR_BAD_ARG;
  if (N == 0) return MSS_OK;
  if (N > 65535 || (C + 63) / 64 > 65535) return MSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 63) / 64, (C + 63) / 64, N), dim3(256), 0, S_(stream), x, ldx, x_sample_stride, HW, C, y);
  return mss_launch_status();
}

}  // extern "C"

// ================================================================================================ backward kernels
// (the pixel decoder is trained in Mask2Former stage 2: train_m2f.py:291-310 unfreezes sem_seg_head.pixel_decoder)
namespace {

// GroupNorm backward, stage 1: per (n, chunk, quad) partial sums of g*gamma and g*gamma*xhat (for dx) -> part[n][chunk][2][C/4],
// and per (n, chunk, channel) partial sums of g*xhat / g (for dgamma / dbeta) -> pgb[n][chunk][2][C]
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float* __restrict__ gy, int ldg, long long g_ss,
                                                           const float* __restrict__ x, int ldx, long long x_ss, int HW, int C,
                                                           int groups, const float* __restrict__ stat,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int relu, float* __restrict__ part, float* __restrict__ pgb,
                                                           int rows_per_chunk) {
  const int C4 = C >> 2;
  const int QPB = C4 < 256 ? C4 : 256, RPB = 256 / QPB;
  const int tx = threadIdx.x % QPB, ty = threadIdx.x / QPB;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
  __shared__ f32x4 red[2][256];
  __shared__ float reds[2][256];
  f32x4 sgx = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  if (ty < RPB) {
    const int c = tx * 4;
    const int g = c / (C / groups);
    const float mean = stat[2 * (n * groups + g)], rstd = stat[2 * (n * groups + g) + 1];
    const f32x4 gm = ld4(gamma + c), bt = ld4(beta + c);
    for (int p = p0 + ty; p < p1; p += RPB) {
      const f32x4 xh = (ld4(x + (long long)n * x_ss + (long long)p * ldx + c) - mean) * rstd;
      f32x4 gv = ld4(gy + (long long)n * g_ss + (long long)p * ldg + c);
      if (relu) {
        const f32x4 yv = xh * gm + bt;
        gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f; gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
      }
      sgx += gv * xh; sg += gv;
      const f32x4 gg = gv * gm;
      s1 += (gg.x + gg.y) + (gg.z + gg.w);
      const f32x4 t = gg * xh;
      s2 += (t.x + t.y) + (t.z + t.w);
    }
  }
  red[0][threadIdx.x] = sgx; red[1][threadIdx.x] = sg;
  reds[0][threadIdx.x] = s1; reds[1][threadIdx.x] = s2;
  __syncthreads();
  if (ty == 0) {
    for (int yy = 1; yy < RPB; ++yy) {
      sgx += red[0][yy * QPB + tx]; sg += red[1][yy * QPB + tx];
      s1 += reds[0][yy * QPB + tx]; s2 += reds[1][yy * QPB + tx];
    }
    const size_t blk = (size_t)n * gridDim.x + blockIdx.x;
    part[blk * 2 * C4 + tx] = s1;
    part[blk * 2 * C4 + C4 + tx] = s2;
    st4(pgb + blk * 2 * C + tx * 4, sgx);
    st4(pgb + blk * 2 * C + C + tx * 4, sg);
  }
}

// stage 2: per (n, group) the two means of the dx formula, chunks and quads added in a fixed order in double
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ part, int N, int nchunks, int C, int groups,
                                                             int HW, float* __restrict__ m12) {   // one wave per (n, group), as gn_finalize_kernel
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= N * groups) return;
  const int n = i / groups, g = i - n * groups;
  const int C4 = C >> 2, qpg = (C / groups) >> 2;
  double a = 0.0, b = 0.0;
  for (int ch = lane; ch < nchunks; ch += 64) {
    const float* o = part + (((size_t)n * nchunks + ch) * 2) * C4;
    for (int k = 0; k < qpg; ++k) { a += o[g * qpg + k]; b += o[C4 + g * qpg + k]; }
  }
  a = mss_wave_sum_d(a);
  b = mss_wave_sum_d(b);
  if (lane != 0) return;
  const double cnt = (double)HW * (C / groups);
  m12[2 * i] = (float)(a / cnt);
  m12[2 * i + 1] = (float)(b / cnt);
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)); grid (ceil(HW*C/4 / 256), N)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ gy, int ldg, long long g_ss,
                                                           const float* __restrict__ x, int ldx, long long x_ss, int HW, int C,
                                                           int groups, const float* __restrict__ stat, const float* __restrict__ m12,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int relu, float* __restrict__ dx, int lddx) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  const unsigned p = item / C4;
  if (p >= (unsigned)HW) return;
  const int c = (int)(item - p * C4) * 4;
  const int n = blockIdx.y;
  const int g = c / (C / groups);
  const float mean = stat[2 * (n * groups + g)], rstd = stat[2 * (n * groups + g) + 1];
  const float m1 = m12[2 * (n * groups + g)], m2 = m12[2 * (n * groups + g) + 1];
  const f32x4 gm = ld4(gamma + c);
  const f32x4 xh = (ld4(x + (long long)n * x_ss + (long long)p * ldx + c) - mean) * rstd;
  f32x4 gv = ld4(gy + (long long)n * g_ss + (long long)p * ldg + c);
  if (relu) {
    const f32x4 yv = xh * gm + ld4(beta + c);
    gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f; gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
  }
  st4(dx + ((long long)n * HW + p) * lddx + c, rstd * (gv * gm - m1 - xh * m2));
}

// transpose of upsample_add_kernel's bilinear part: dtop[n][iy][ix] = sum over outputs of weight * dy; gather form with
// the candidate output range of each source cell; grid (ceil(IW * C/4 / 256), IH, N). accumulate: dtop += (token buffers
// that already hold another gradient).
__global__ __launch_bounds__(256) void upsample_hp_bwd_kernel(const float* __restrict__ dy, int lddy, int OH, int OW,
                                                              float* __restrict__ dtop, int ldt, long long top_ss, int IH, int IW,
                                                              int C, float sh, float sw, int accumulate) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned item = blockIdx.x * 256u + threadIdx.x;
  const unsigned ix = item / C4;
  if (ix >= (unsigned)IW) return;
  const int c = (int)(item - ix * C4) * 4;
  const int iy = blockIdx.y, n = blockIdx.z;
  auto weight = [](int o, int i, float scale, int in) {
    float src = ((float)o + 0.5f) * scale;
    asm volatile("" : "+v"(src));
    src -= 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
    const float l1 = src - (float)i0;
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    return w;
  };
  // outputs whose source index can touch cell i: src in (i-1, i+1) -> o in ((i-0.5)/s - 0.5 - 1, (i+1.5)/s - 0.5 + 1), padded
  const float ish = 1.f / sh, isw = 1.f / sw;
  int ylo = (int)floorf(((float)iy - 1.f) * ish) - 2, yhi = (int)ceilf(((float)iy + 2.f) * ish) + 2;
  int xlo = (int)floorf(((float)ix - 1.f) * isw) - 2, xhi = (int)ceilf(((float)ix + 2.f) * isw) + 2;
  ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
  yhi = yhi > OH - 1 ? OH - 1 : yhi; xhi = xhi > OW - 1 ? OW - 1 : xhi;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* b = dy + (long long)n * OH * OW * lddy + c;
  for (int oy = ylo; oy <= yhi; ++oy) {
    const float wy = weight(oy, iy, sh, IH);
    if (wy == 0.f) continue;
    f32x4 racc = {0.f, 0.f, 0.f, 0.f};
    for (int ox = xlo; ox <= xhi; ++ox) {
      const float wx = weight(ox, (int)ix, sw, IW);
      if (wx != 0.f) racc += wx * ld4(b + ((long long)oy * OW + ox) * lddy);
    }
    acc += wy * racc;
  }
  float* o = dtop + (long long)n * top_ss + ((long long)iy * IW + ix) * ldt + c;
  if (accumulate) acc += ld4(o);
  st4(o, acc);
}

// dst (token rows of a level: pixel stride ldd, sample stride d_ss) = NCHW gradient g (contiguous), optional accumulate.
// 64 pixels x 64 channels per workgroup through LDS; grid (ceil(HW/64), ceil(C/64), N)
__global__ __launch_bounds__(256) void nchw_to_nhwc_strided_kernel(const float* __restrict__ g, int HW, int C,
                                                                   float* __restrict__ dst, int ldd, long long d_ss,
                                                                   int accumulate) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + tx;
    tile[r][tx] = (c < C && p < HW) ? g[((long long)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) {
      float* o = dst + (long long)n * d_ss + (long long)p * ldd + c;
      *o = accumulate ? *o + tile[tx][r] : tile[tx][r];
    }
  }
}

}  // namespace

extern "C" {

// backward of mss_groupnorm_nhwc_f32 (same x, strides and statistics layout; relu: the forward fused a ReLU): dx (contiguous
// NHWC, pixel stride lddx), dgamma / dbeta [C]. ws: mss_groupnorm_bwd_workspace_floats floats. stat = the forward's
// [N*groups][2] (mean, rstd), which the forward leaves at ws_fwd + N*chunks*2*(C/4) (see mss_groupnorm_stat_offset).
long long mss_groupnorm_stat_offset(int N, int HW, int C) {
  long long chunks = (HW + 255) / 256;
  if (chunks > 512) chunks = 512;
  const int rpc = (int)((HW + chunks - 1) / chunks);
  chunks = (HW + rpc - 1) / rpc;
  return (long long)N * chunks * 2 * (C / 4);
}
long long mss_groupnorm_bwd_workspace_floats(int N, int HW, int C, int groups) {
  if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
  long long chunks = (HW + 255) / 256;
  if (chunks > 512) chunks = 512;
  return (long long)N * chunks * (2 * (C / 4) + 2 * C) + 2ll * N * groups;
}
int mss_groupnorm_nhwc_bwd_f32(const float* gy, int ldg, long long g_sample_stride, const float* x, int ldx,
                               long long x_sample_stride, int N, int HW, int C, int groups, const float* stat,
                               const float* gamma, const float* beta, int relu, float* dx, int lddx, float* dgamma,
                               float* dbeta, float* ws, void* stream) {
  if (!gy || !x || !stat || !gamma || !beta || !dx || !ws || N < 0 || HW <= 0) return MSS_ERR_BAD_ARG;
  if (N == 0) return MSS_OK;
  if (C % 4 || ldx % 4 || ldg % 4 || lddx % 4 || groups <= 0 || C % groups || (C / groups) % 4 || C / 4 > 256 || N > 65535)
    return MSS_ERR_UNSUPPORTED;
  long long chunks = (HW + 255) / 256;
  if (chunks > 512) chunks = 512;
  const int rpc = (int)((HW + chunks - 1) / chunks);
  chunks = (HW + rpc - 1) / rpc;
  float* part = ws;
  float* pgb = part + (size_t)N * chunks * 2 * (C / 4);
  float* m12 = pgb + (size_t)N * chunks * 2 * C;
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3((unsigned)chunks, N), dim3(256), 0, S_(stream), gy, ldg, g_sample_stride, x, ldx,
                     x_sample_stride, HW, C, groups, stat, gamma, beta, relu, part, pgb, rpc);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N * groups), dim3(64), 0, S_(stream), part, N, (int)chunks, C, groups, HW, m12);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)(((long long)HW * (C / 4) + 255) / 256), N), dim3(256), 0, S_(stream), gy, ldg,
                     g_sample_stride, x, ldx, x_sample_stride, HW, C, groups, stat, m12, gamma, beta, relu, dx, lddx);
  const int nparts = (int)(N * chunks);
  if (dgamma) hipLaunchKernelGGL(ordered_colsum_kernel, dim3((C + 3) / 4), dim3(256), 0, S_(stream), pgb, nparts, C, 2ll * C, dgamma);
  if (dbeta) hipLaunchKernelGGL(ordered_colsum_kernel, dim3((C + 3) / 4), dim3(256), 0, S_(stream), pgb + C, nparts, C, 2ll * C, dbeta);
  return mss_launch_status();
}

// transpose of the bilinear part of mss_upsample_bilinear_add_nhwc_f32: dtop (+)= B^T dy; the lateral gradient is dy itself
int mss_upsample_bilinear_bwd_nhwc_f32(const float* dy, int lddy, int N, int OH, int OW, float* dtop, int ldt,
                                       long long top_sample_stride, int IH, int IW, int C, int accumulate, void* stream) {
  if (!dy || !dtop || C % 4 || lddy % 4 || ldt % 4) return MSS_ERR_BAD_ARG;
  if ((long long)N * IH * IW == 0) return MSS_OK;
  if (IH > 65535 || N > 65535) return MSS_ERR_UNSUPPORTED;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  hipLaunchKernelGGL(upsample_hp_bwd_kernel, dim3((unsigned)(((long long)IW * (C / 4) + 255) / 256), IH, N), dim3(256), 0, S_(stream),
                     dy, lddy, OH, OW, dtop, ldt, top_sample_stride, IH, IW, C, sh, sw, accumulate);
  return mss_launch_status();
}

// NCHW gradient -> rows of an NHWC / token buffer (pixel stride ldd, sample stride d_sample_stride floats), optional +=
int mss_nchw_to_nhwc_strided_f32(const float* g, int N, int C, int HW, float* dst, int ldd, long long d_sample_stride,
                                 int accumulate, void* stream) {
  if (!g || !dst || N < 0 || HW <= 0 || C <= 0) return MSS_ERR_BAD_ARG;
  if (N == 0) return MSS_OK;
  if (N > 65535 || (C + 63) / 64 > 65535) return MSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(nchw_to_nhwc_strided_kernel, dim3((HW + 63) / 64, (C + 63) / 64, N), dim3(256), 0, S_(stream), g, HW, C, dst, ldd,
                     d_sample_stride, accumulate);
  return mss_launch_status();
}

}  // extern "C"
