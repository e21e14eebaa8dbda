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

}  // namespace

extern "C" {

int mss_m2f_fused_score_f32(const float* cls, const float* logit, int B, int Q, int C, int hm, int wm, int ldq, int Hi,
                            int Wi, int H, int W, float* score, void* stream) {
  if (!cls || !logit || !score || H > Hi || W > Wi || hm < 1 || wm < 1 || Hi < 1 || Wi < 1) return MSS_ERR_BAD_ARG;
  if (C > CP || C < 1 || Q % 4 || ldq % 4 || ldq < Q || Q < 4) return MSS_ERR_UNSUPPORTED;
  if ((long long)B * H * W == 0) return MSS_OK;
  if (B > 65535) return MSS_ERR_UNSUPPORTED;
  const float sy = (float)hm / (float)Hi, sx = (float)wm / (float)Wi;     // ATen: scale = in / out when no scale_factor is given
  const int QP = Q | 1;
  int TY = 16, fh = 0;
  const int fw = (int)floorf((TX - 1) * sx) + 3;   // i0(first) .. i1(last) spans at most floor((n-1)*scale) + 3 source pixels
  size_t smem = 0;
  for (; TY >= 4; TY >>= 1) {
    fh = (int)floorf((TY - 1) * sy) + 3;
    smem = ((size_t)Q * CP + (size_t)fh * fw * QP) * sizeof(float);
    if (smem <= 60 * 1024) break;
  }
  if (TY < 4) return MSS_ERR_UNSUPPORTED;       // strong down-sampling: not what this path is for
  hipLaunchKernelGGL(m2f_fused_score_kernel, dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B), dim3(NT), smem,
                     static_cast<hipStream_t>(stream), cls, logit, Q, C, hm, wm, ldq, H, W, TY, sy, sx, score);
  return mss_launch_status();
}

}  // extern "C"
