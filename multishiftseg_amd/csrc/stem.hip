// The stem of the WideResNet trunk in ONE kernel: mod1.conv1 (3 -> 64, 3x3, padding 1, no bias; wider_resnet.py:343-345) followed
// directly by pool2 = MaxPool2d(3, stride 2, padding 1) (wider_resnet.py:353-355 -- there is no BatchNorm between them: mod2's
// first operation is the pre-activation BN of the POOLED map), NCHW image in, pooled NHWC map out.
//
// The two-kernel path (im2col + K = 32 GEMM, then the pool) writes and re-reads the full-resolution 64-channel map: 1.07 GB each
// way at 2 x 1024 x 2048, plus 0.54 GB of patches -- 2.7 GB of traffic for an output of 268 MB. Here nothing but the 50 MB image
// is read and nothing but the pooled map is written; the convolution runs on the fp32 matrix cores straight out of registers:
//
//  * a WAVE owns one pooled row (a segment of it) and sweeps it left to right in blocks of 32 convolution columns = 16 pooled
//    columns. Per block it computes the three convolution rows 2py-1, 2py, 2py+1 x 32 columns x 64 channels as 3 x 2 MFMA
//    accumulators (v_mfma_f32_32x32x2_f32, K = 27 taps in 15 instructions) -- each convolution row is computed by the two pooled
//    rows that use it (1.5x the MFMA work, 0.44 TFLOP in all) in exchange for NO communication between waves: no LDS, no barrier;
//  * the A operand (lane = convolution column, two taps per instruction) comes straight from the image by coalesced 4-byte loads
//    (the image rows stay in L1/L2: every value is used by 27 taps x 3 rows), the B operand (the 64 x 27 weights) lives in 30
//    registers per lane for the whole sweep;
//  * the accumulator layout (lane = channel, registers = 16 of the 32 columns) makes the horizontal 3-max of a stride-2 window an
//    in-lane operation except for the one column to the left of every fourth pair, which sits in the other half of the wave (one
//    cross-half shuffle per 8 columns) or, at the block's left edge, in the previous block (carried in a register);
//  * the vertical max is over the wave's own three accumulators; the pooled values leave as 128-byte runs (32 channels) per lane half.
//
// The K-order of the products (tap pairs (s, s+4) of every 8 taps) is the one the K = 32 GEMM of the two-kernel path uses, so the
// convolution values -- and with them the pooled map -- are the same numbers.
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

constexpr int STEM_K = 64, STEM_TAPS = 27, STEM_M = 15;      // MFMA m multiplies taps k0(m) (lane half 0) and k0(m) + 4 (half 1)
__device__ __forceinline__ constexpr int stem_k0(int m) { return (m >> 2) * 8 + (m & 3); }

__global__ __launch_bounds__(256, 2) void stem_conv_pool_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             float* __restrict__ y, int ldy, int N, int H, int W, int OH, int OW,
                                                             int nblk, int nseg, int bps) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, in SGPRs
  const int seg = (int)(gw % nseg);
  const int py = (int)((gw / nseg) % OH);
  const long long n = gw / ((long long)nseg * OH);
  if (n >= N) return;
  const float NEG = -__builtin_huge_valf();
  // ---- B operand: this lane's weight of tap k0(m) + 4h for channel nb * 32 + i; the A operand's offset from (row, column)
  float bw[2][STEM_M];
  int toff[STEM_M];
#pragma unroll
  for (int m = 0; m < STEM_M; ++m) {
    const int k = stem_k0(m) + 4 * h;
    const bool tok = k < STEM_TAPS;
    const int kk = tok ? k : 0;                         // taps 27..31 do not exist: weight 0 on a value of the pixel's own window
    const int c = kk / 9, r = (kk % 9) / 3, t = kk % 3;
    toff[m] = (c * H + (r - 1)) * W + (t - 1);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bw[nb][m] = tok ? w[(nb * 32 + i) * STEM_TAPS + k] : 0.f;
  }
  // the image of this wave as a uniform base + 32-bit byte offsets (3 H W < 2^29 elements, checked by the launcher)
  const char* base = reinterpret_cast<const char*>(img + n * 3 * (long long)H * W);
  const int cy0 = 2 * py - 1;                          // first of the three convolution rows
  const bool rows_inside = cy0 - 1 >= 0 && cy0 + 3 < H;   // wave-uniform: every tap row of the three convolution rows is in the image
  const int b0 = seg * bps, b1 = min(b0 + bps, nblk);
  const int bfirst = b0 > 0 ? b0 - 1 : b0;             // a segment that does not start at the image edge runs one block ahead for the carry
  // operands of block b: coalesced 4-byte loads straight from the image; interior blocks need no bounds test at all
  auto load_block = [&](int b, float (&a)[3][STEM_M]) {
    const int cx = 32 * b + i;
    if (rows_inside && 32 * b - 1 >= 0 && 32 * b + 32 < W) {
#pragma unroll
      for (int R = 0; R < 3; ++R) {
        const unsigned o = (unsigned)((cy0 + R) * W + cx);
#pragma unroll
        for (int m = 0; m < STEM_M; ++m) a[R][m] = *reinterpret_cast<const float*>(base + 4u * (o + (unsigned)toff[m]));
      }
    } else {
#pragma unroll
      for (int R = 0; R < 3; ++R) {
        const int cy = cy0 + R;
#pragma unroll
        for (int m = 0; m < STEM_M; ++m) {
          const int k0 = stem_k0(m), k1 = k0 + 4;
          const int kk = h ? (k1 < STEM_TAPS ? k1 : 0) : k0;
          const int yy = cy + (kk % 9) / 3 - 1, xx = cx + kk % 3 - 1;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          a[R][m] = ok ? *reinterpret_cast<const float*>(base + 4u * (unsigned)(cy * W + cx + toff[m])) : 0.f;
        }
      }
    }
  };
  float carry[3][2];                                   // column 32 b - 1 of the three rows (per channel; used by lane half 0)
#pragma unroll
  for (int R = 0; R < 3; ++R) { carry[R][0] = NEG; carry[R][1] = NEG; }
  float a[3][STEM_M], an[3][STEM_M];
  load_block(bfirst, an);
  for (int b = bfirst; b < b1; ++b) {
#pragma unroll
    for (int R = 0; R < 3; ++R)
#pragma unroll
      for (int m = 0; m < STEM_M; ++m) a[R][m] = an[R][m];
    if (b + 1 < b1) load_block(b + 1, an);             // the next block's operands travel while this block multiplies
    float* yrow = y + ((n * OH + py) * (long long)OW) * ldy;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      f32x16 acc[3];
#pragma unroll
      for (int R = 0; R < 3; ++R)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[R][r] = 0.f;
#pragma unroll
      for (int m = 0; m < STEM_M; ++m)
#pragma unroll
        for (int R = 0; R < 3; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[R][m], bw[nb][m], acc[R], 0, 0, 0);
      // ---- pooling. Register r = 4 q + e of an accumulator is convolution column 32 b + 8 q + 4 h + e.
      float out[4][2];                                 // [q][u]: pooled column 16 b + 4 q + 2 h + u
#pragma unroll
      for (int q = 0; q < 4; ++q) { out[q][0] = NEG; out[q][1] = NEG; }
#pragma unroll
      for (int R = 0; R < 3; ++R) {
        const int cy = cy0 + R;
        const bool row_ok = cy >= 0 && cy < H;         // wave-uniform
        f32x16 v = acc[R];
        if (32 * b + 32 > W) {                         // last block: columns past the image do not take part in the max
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * b + 8 * (r >> 2) + 4 * h + (r & 3) >= W) v[r] = NEG;
        }
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = __shfl_xor(v[4 * q + 3], 32);     // the other half's last column of group q
        if (row_ok) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float left = h ? x[q] : (q ? x[q > 0 ? q - 1 : 0] : carry[R][nb]);
            out[q][0] = fmaxf(out[q][0], fmaxf(fmaxf(v[4 * q], v[4 * q + 1]), left));
            out[q][1] = fmaxf(out[q][1], fmaxf(fmaxf(v[4 * q + 2], v[4 * q + 3]), v[4 * q + 1]));
          }
        }
        carry[R][nb] = x[3];                           // half 0 now holds half 1's column 31 of this block
      }
      if (b >= b0) {                                   // (the run-ahead block only produces the carry)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int px = 16 * b + 4 * q + 2 * h + u;
            if (px < OW) yrow[(long long)px * ldy + nb * 32 + i] = out[q][u];
          }
      }
    }
  }
}

}  // namespace

extern "C" {

// y [N][OH][OW][64] (NHWC, pixel stride ldy >= 64) = MaxPool2d(3, 2, 1)(conv3x3(img [N][3][H][W] NCHW, w [64][3][3][3], padding 1)),
// OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1. Replaces mss_im2col3x3_c3_f32 + mss_conv2d_forward_f32 + mss_maxpool3s2_nhwc_f32
// for the trunk's stem (wider_resnet.py:343-345,353-355).
int mss_stem_conv_pool_f32(const float* img, const float* w, float* y, int ldy, int N, int H, int W, void* stream) {
  if (!img || !w || !y || N < 0 || H <= 0 || W <= 0 || ldy < STEM_K) return MSS_ERR_BAD_ARG;
  if (N == 0) return MSS_OK;
  if ((long long)3 * H * W >= (1ll << 29)) return MSS_ERR_UNSUPPORTED;      // 32-bit byte offsets inside one image
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int nblk = (OW + 15) / 16;
  // segments per pooled row: enough waves for ~4 per SIMD (1024 SIMDs), but at least 8 blocks per segment (the run-ahead block)
  int nseg = (int)((4096 + (long long)N * OH - 1) / ((long long)N * OH));
  if (nseg > nblk / 8) nseg = nblk / 8;
  if (nseg < 1) nseg = 1;
  const int bps = (nblk + nseg - 1) / nseg;
  nseg = (nblk + bps - 1) / bps;
  const long long waves = (long long)N * OH * nseg;
  const long long grid = (waves + 3) / 4;
  if (grid >= (1ll << 31)) return MSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(stem_conv_pool_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), img, w, y, ldy, N,
                     H, W, OH, OW, nblk, nseg, bps);
  return mss_launch_status();
}

}  // extern "C"
