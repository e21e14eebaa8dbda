// Mask2Former anomaly score fused with the x4 mask upsample.
//
// Reference chain per image (B x Q=100 queries, C=19 classes):
//   pred_masks = einsum("bqc,bchw->bqhw", mask_embed, mask_features)              mask2former_transformer_decoder.py:544-548
//   pred_masks = F.interpolate(pred_masks, size=image, mode="bilinear", align_corners=False)   maskformer_model.py:264-277
//   score = 1 - max_c sum_q softmax(cls)[q, c<C] * sigmoid(pred_masks)[q]         train_m2f.py:387-407 (cropped to `size`)
// which materialises [B,100,H,W] logits (839 MB at 1024x2048) and reads them back twice. Here the first line is one
// batched GEMM on conv_igemm writing pixel-major low-resolution logits [B, hm, wm, ldq] (52 MB), and this kernel
// does the rest: a workgroup stages the low-resolution footprint of its 16x64 output tile and the image's class
// probabilities in LDS, then every thread interpolates, applies the sigmoid and mixes the classes for 4 output
// pixels. HBM traffic: the low-resolution logits once + 4 B per output pixel.
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

constexpr int TX = 64, NT = 256, CP = 20, PPT = 4;   // tile width, threads, padded classes, pixels per thread

struct SrcCoord { int i0, i1; float l; };
// F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5) * in/out - 0.5, clamped at 0
__device__ __forceinline__ SrcCoord src_coord(int dst, float scale, int in_size) {
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  SrcCoord c;
  c.i0 = (int)s;
  if (c.i0 > in_size - 1) c.i0 = in_size - 1;
  c.i1 = c.i0 + (c.i0 < in_size - 1 ? 1 : 0);
  c.l = s - (float)c.i0;
  return c;
}

__global__ __launch_bounds__(NT) void m2f_fused_score_kernel(const float* __restrict__ cls, const float* __restrict__ logit,
                                                             int Q, int C, int hm, int wm, int ldq, int H, int W, int TY,
                                                             float sy, float sx, float* __restrict__ score) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int QP = Q | 1;                         // odd row stride: neighbouring footprint pixels fall in different banks
  float* prob = smem;                           // [Q][CP]
  float* foot = smem + Q * CP;                  // [fh*fw][QP]
  const int b = blockIdx.z;
  const int ox0 = blockIdx.x * TX, oy0 = blockIdx.y * TY;
  // class probabilities of this image (softmax over C+1, "no object" column dropped)
  for (int q = threadIdx.x; q < Q; q += NT) {
    const float* row = cls + ((long long)b * Q + q) * (C + 1);
    float m = -__builtin_huge_valf();
    for (int c = 0; c <= C; ++c) m = fmaxf(m, row[c]);
    float s = 0.f;
    for (int c = 0; c <= C; ++c) s += expf(row[c] - m);
    const float inv = 1.f / s;
    for (int c = 0; c < CP; ++c) prob[q * CP + c] = c < C ? expf(row[c] - m) * inv : 0.f;
  }
  // low-resolution footprint of the tile
  const int oy_last = min(oy0 + TY, H) - 1, ox_last = min(ox0 + TX, W) - 1;
  const int fy0 = src_coord(oy0, sy, hm).i0, fy1 = src_coord(oy_last, sy, hm).i1;
  const int fx0 = src_coord(ox0, sx, wm).i0, fx1 = src_coord(ox_last, sx, wm).i1;
  const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;
  const int Q4 = Q >> 2;                        // Q % 4 == 0 and ldq % 4 == 0 are preconditions: float4 rows
  for (int i = threadIdx.x; i < fh * fw * Q4; i += NT) {
    const int pix = i / Q4, q4 = i - pix * Q4;
    const int fy = pix / fw, fx = pix - fy * fw;
    const f32x4 v = *reinterpret_cast<const f32x4*>(logit + (((long long)b * hm + fy0 + fy) * wm + fx0 + fx) * ldq + 4 * q4);
    float* d = foot + pix * QP + 4 * q4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;      // 4 thread rows; rows ty, ty+4, ...
  const int ox = ox0 + tx;
  if (ox >= W) return;
  const SrcCoord cx = src_coord(ox, sx, wm);
  const int rows = TY / (NT / TX);                                    // output rows per thread (<= PPT)
  const float wx0 = 1.f - cx.l, wx1 = cx.l;
  int o00[PPT], o01[PPT], o10[PPT], o11[PPT];
  float hy0[PPT], hy1[PPT];
  bool live[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int oy = oy0 + ty + k * (NT / TX);
    live[k] = k < rows && oy < H;
    const SrcCoord cy = src_coord(live[k] ? oy : oy0, sy, hm);
    const int r0 = (cy.i0 - fy0) * fw, r1 = (cy.i1 - fy0) * fw;
    o00[k] = (r0 + cx.i0 - fx0) * QP; o01[k] = (r0 + cx.i1 - fx0) * QP;
    o10[k] = (r1 + cx.i0 - fx0) * QP; o11[k] = (r1 + cx.i1 - fx0) * QP;
    hy0[k] = 1.f - cy.l; hy1[k] = cy.l;
  }
  f32x4 acc[PPT][CP / 4];
#pragma unroll
  for (int k = 0; k < PPT; ++k)
#pragma unroll
    for (int j = 0; j < CP / 4; ++j) acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < Q; ++q) {
    f32x4 p[CP / 4];
#pragma unroll
    for (int j = 0; j < CP / 4; ++j) p[j] = *reinterpret_cast<const f32x4*>(&prob[q * CP + 4 * j]);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      // same association as ATen's upsample_bilinear2d: h0 * (w0 v00 + w1 v01) + h1 * (w0 v10 + w1 v11)
      const float v = hy0[k] * (wx0 * foot[o00[k] + q] + wx1 * foot[o01[k] + q]) +
                      hy1[k] * (wx0 * foot[o10[k] + q] + wx1 * foot[o11[k] + q]);
      const float sg = 1.f / (1.f + expf(-v));
#pragma unroll
      for (int j = 0; j < CP / 4; ++j) acc[k][j] += sg * p[j];
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (!live[k]) continue;
    float best = -__builtin_huge_valf();
#pragma unroll
    for (int j = 0; j < CP / 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * j + e < C) best = fmaxf(best, acc[k][j][e]);
    score[((long long)b * H + oy0 + ty + k * (NT / TX)) * W + ox] = 1.f - best;
  }
}


// r04: the class mix on the matrix cores. Per output pixel the old kernel spends 100 x (interpolate + sigmoid) AND 100 x 19 (20)
// multiply-adds on the vector ALUs -- 0.316 ms at 1024 x 2048, VALU-bound at 0.024 of the HBM roofline. The
// mix is a [classes x Q] x [Q x pixels] product: v_mfma_f32_32x32x2_f32 with A = prob^T (32 class rows, 19 used) and B = the
// sigmoids of 32 pixels for 2 queries, so a lane computes ONE interpolation + sigmoid per MFMA (its pixel = lane & 31, its query
// = 2 s + (lane >> 5)) and the 64 accumulated cycles of the matrix pipe run beside the next step's vector work. D layout: column
// = pixel = lane & 31, rows = classes (r & 3) + 8 (r >> 2) + 4 (lane >> 5): the class maximum is 16 in-lane compares + one
// shuffle with lane ^ 32. A wave owns 32 consecutive x of two output rows at a time (two independent accumulators).
constexpr int PCOLS = 32;   // prob row stride: classes padded to the MFMA's 32 rows
constexpr int NTM = 512;    // 8 waves: (x half) x (4 row groups); two workgroups per CU = 4 waves per SIMD to interleave VALU and MFMA

// class probabilities of every image, once: softmax over C + 1, "no object" column dropped, classes >= C zero -> [B][Q][PCOLS]
// (every workgroup of the score kernel used to recompute its image's 100 x 20 table: ~2 us in front of each tile)
__global__ __launch_bounds__(256) void m2f_prob_kernel(const float* __restrict__ cls, int BQ, int C, float* __restrict__ prob) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BQ) return;
  const float* row = cls + (long long)i * (C + 1);
  float m = -__builtin_huge_valf();
  for (int c = 0; c <= C; ++c) m = fmaxf(m, row[c]);
  float sum = 0.f;
  for (int c = 0; c <= C; ++c) sum += expf(row[c] - m);
  const float inv = 1.f / sum;
  for (int c = 0; c < PCOLS; ++c) prob[(long long)i * PCOLS + c] = c < C ? expf(row[c] - m) * inv : 0.f;
}

__global__ __launch_bounds__(NTM) void m2f_fused_score_mfma_kernel(const float* __restrict__ probg, const float* __restrict__ logit,
                                                                   int Q, int C, int hm, int wm, int ldq, int H, int W, int TY,
                                                                   float sy, float sx, float* __restrict__ score) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int QP = Q | 1;
  float* prob = smem;                           // [Q][PCOLS]
  float* foot = smem + Q * PCOLS;               // [fh*fw][QP]
  const int b = blockIdx.z;
  const int ox0 = blockIdx.x * TX, oy0 = blockIdx.y * TY;
  for (int i = threadIdx.x; i < Q * PCOLS / 4; i += NTM)
    reinterpret_cast<f32x4*>(prob)[i] = reinterpret_cast<const f32x4*>(probg + (long long)b * Q * PCOLS)[i];
  const int oy_last = min(oy0 + TY, H) - 1, ox_last = min(ox0 + TX, W) - 1;
  const int fy0 = src_coord(oy0, sy, hm).i0, fy1 = src_coord(oy_last, sy, hm).i1;
  const int fx0 = src_coord(ox0, sx, wm).i0, fx1 = src_coord(ox_last, sx, wm).i1;
  const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;
  const int Q4 = Q >> 2;
  for (int i = threadIdx.x; i < fh * fw * Q4; i += NTM) {
    const int pix = i / Q4, q4 = i - pix * Q4;
    const int fy = pix / fw, fx = pix - fy * fw;
    const f32x4 v = *reinterpret_cast<const f32x4*>(logit + (((long long)b * hm + fy0 + fy) * wm + fx0 + fx) * ldq + 4 * q4);
    float* d = foot + pix * QP + 4 * q4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 31, kh = lane >> 5;
  const int ox = ox0 + (wave & 1) * 32 + px;
  const SrcCoord cx = src_coord(ox < W ? ox : W - 1, sx, wm);
  const float wx0 = 1.f - cx.l, wx1 = cx.l;
  const int c0 = cx.i0 - fx0, c1 = cx.i1 - fx0;
  const int steps = Q >> 1;                                         // even: Q % 4 == 0
  for (int rA = wave >> 1; rA < TY; rA += 8) {                      // this wave's rows of the tile: rA and rA + 4
    const int oyA = oy0 + rA, oyB = oyA + 4;
    const bool liveB = rA + 4 < TY;
    const SrcCoord ya = src_coord(oyA < H ? oyA : H - 1, sy, hm), yb = src_coord(oyB < H && liveB ? oyB : min(oyA, H - 1), sy, hm);
    const float* fa0 = foot + ((ya.i0 - fy0) * fw) * QP + kh;
    const float* fa1 = foot + ((ya.i1 - fy0) * fw) * QP + kh;
    const float* fb0 = foot + ((yb.i0 - fy0) * fw) * QP + kh;
    const float* fb1 = foot + ((yb.i1 - fy0) * fw) * QP + kh;
    const float* a00 = fa0 + c0 * QP; const float* a01 = fa0 + c1 * QP; const float* a10 = fa1 + c0 * QP; const float* a11 = fa1 + c1 * QP;
    const float* b00 = fb0 + c0 * QP; const float* b01 = fb0 + c1 * QP; const float* b10 = fb1 + c0 * QP; const float* b11 = fb1 + c1 * QP;
    const float* pb = prob + kh * PCOLS + px;                       // prob[(2 s + kh)][class = px] at pb[s * 2 * PCOLS]
    const float ha0 = 1.f - ya.l, ha1 = ya.l, hb0 = 1.f - yb.l, hb1 = yb.l;
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    // u-th step after the pointers: its nine LDS addresses are immediates
    auto step = [&](int u) {
      const float p = pb[u * 2 * PCOLS];
      // same association as ATen's upsample_bilinear2d: h0 * (w0 v00 + w1 v01) + h1 * (w0 v10 + w1 v11)
      const float va = ha0 * (wx0 * a00[2 * u] + wx1 * a01[2 * u]) + ha1 * (wx0 * a10[2 * u] + wx1 * a11[2 * u]);
      const float vb = hb0 * (wx0 * b00[2 * u] + wx1 * b01[2 * u]) + hb1 * (wx0 * b10[2 * u] + wx1 * b11[2 * u]);
      const float sa = __builtin_amdgcn_rcpf(1.f + __expf(-va));
      const float sb = __builtin_amdgcn_rcpf(1.f + __expf(-vb));
      accA = __builtin_amdgcn_mfma_f32_32x32x2f32(p, sa, accA, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x2f32(p, sb, accB, 0, 0, 0);
    };
    int s2 = 0;
    for (; s2 + 4 <= steps; s2 += 4) {
      step(0); step(1); step(2); step(3);
      pb += 8 * PCOLS;
      a00 += 8; a01 += 8; a10 += 8; a11 += 8; b00 += 8; b01 += 8; b10 += 8; b11 += 8;
    }
    if (s2 < steps) { step(0); step(1); }
    float bestA = -__builtin_huge_valf(), bestB = bestA;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (c < C) { bestA = fmaxf(bestA, accA[r]); bestB = fmaxf(bestB, accB[r]); }
    }
    bestA = fmaxf(bestA, __shfl_xor(bestA, 32));
    bestB = fmaxf(bestB, __shfl_xor(bestB, 32));
    if (kh == 0 && ox < W) {
      if (oyA < H) score[((long long)b * H + oyA) * W + ox] = 1.f - bestA;
      if (liveB && oyB < H) score[((long long)b * H + oyB) * W + ox] = 1.f - bestB;
    }
  }
}

}  // namespace

extern "C" {

// prob_ws: device scratch of B * Q * 32 floats (the class-probability table of the MFMA kernel); nullptr selects the all-VALU kernel
int mss_m2f_fused_score_ws_f32(const float* cls, const float* logit, int B, int Q, int C, int hm, int wm, int ldq, int Hi,
                               int Wi, int H, int W, float* score, float* prob_ws, void* stream) {
  if (!cls || !logit || !score || H > Hi || W > Wi || hm < 1 || wm < 1 || Hi < 1 || Wi < 1) return MSS_ERR_BAD_ARG;
  if (C > CP || C < 1 || Q % 4 || ldq % 4 || ldq < Q || Q < 4) return MSS_ERR_UNSUPPORTED;
  if ((long long)B * H * W == 0) return MSS_OK;
  if (B > 65535) return MSS_ERR_UNSUPPORTED;
  const float sy = (float)hm / (float)Hi, sx = (float)wm / (float)Wi;     // ATen: scale = in / out when no scale_factor is given
  const int QP = Q | 1;
  const bool mfma = prob_ws && MSS_ENV_INT("MSS_M2F_MFMA", 1) != 0;   // 0: the round-2 all-VALU kernel (A/B, second formulation in the tests)
  const int pcols = mfma ? PCOLS : CP;
  int TY = 16, fh = 0;
  const int fw = (int)floorf((TX - 1) * sx) + 3;   // i0(first) .. i1(last) spans at most floor((n-1)*scale) + 3 source pixels
  size_t smem = 0;
  for (; TY >= 4; TY >>= 1) {
    fh = (int)floorf((TY - 1) * sy) + 3;
    smem = ((size_t)Q * pcols + (size_t)fh * fw * QP) * sizeof(float);
    if (smem <= 60 * 1024) break;
  }
  if (TY < 4) return MSS_ERR_UNSUPPORTED;       // strong down-sampling: not what this path is for
  const dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, B);
  if (mfma) {
    hipLaunchKernelGGL(m2f_prob_kernel, dim3((B * Q + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), cls, B * Q, C, prob_ws);
    hipLaunchKernelGGL(m2f_fused_score_mfma_kernel, grid, dim3(NTM), smem, static_cast<hipStream_t>(stream), prob_ws, logit, Q, C, hm,
                       wm, ldq, H, W, TY, sy, sx, score);
  } else
    hipLaunchKernelGGL(m2f_fused_score_kernel, grid, dim3(NT), smem, static_cast<hipStream_t>(stream), cls, logit, Q, C, hm, wm, ldq,
                       H, W, TY, sy, sx, score);
  return mss_launch_status();
}

int mss_m2f_fused_score_f32(const float* cls, const float* logit, int B, int Q, int C, int hm, int wm, int ldq, int Hi,
                            int Wi, int H, int W, float* score, void* stream) {
  return mss_m2f_fused_score_ws_f32(cls, logit, B, Q, C, hm, wm, ldq, Hi, Wi, H, W, score, nullptr, stream);
}

}  // extern "C"
