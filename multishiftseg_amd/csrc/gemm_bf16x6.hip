// EXPERIMENTAL, OPT-IN (MSS_GEMM_BF16X6=1; never the default, never the benchmark's headline number):
// the fp32 NT GEMM of gemm.hip computed on the bf16 matrix cores by operand splitting.
//
//   x = x_hi + x_mid + x_lo  exactly to 24 bits (three round-to-nearest bf16 terms), so
//   a * b = a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid) + O(2^-24 |a b|)
// -- six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) per 32x32x16 block instead of eight v_mfma_f32_32x32x2_f32.
// The bf16 pipe is 16x faster per FLOP than the fp32 one, so the six-product form has 2.67x the MFMA throughput of
// the native fp32 instruction at fp32 accuracy: against float64 the result errs by 2e-7 of max|y| (fp32 MFMA / BLAS:
// 3-6e-7), measured in numpy and by tests/test_gpu_ops.py::test_bf16x6_gemm_is_fp32_accurate.
//
// Same tiling as gemm.hip (128 x 128 x 16, 4 waves of 64 x 64); one MFMA covers a whole K-step. LDS holds three bf16
// planes per operand, rows of 16 bf16 = 32 B; the two 16-B halves of row r are swapped when bit 3 of r is set, so the
// 16 lanes of a ds_read_b128 group (rows r..r+15, same half) hit 16 disjoint 4-bank spans.
#include "mss_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256, BM = 128, BN = 128, BK = 16;
constexpr int WTM = 64, WTN = 64, TM = 2, TN = 2;
constexpr int ROW_B = BK * 2;                 // bytes per row per plane
constexpr int PLANE_B = BM * ROW_B;           // 4 KB
constexpr int OPER_B = 3 * PLANE_B;           // 12 KB: hi, mid, lo
constexpr int BUF_B = 2 * OPER_B;             // A then B

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Split { bf16x4 hi, mid, lo; };
__device__ __forceinline__ Split split3(f32x4 v) {
  Split s;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)v[i];
    const float r1 = v[i] - (float)h;         // exact
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;           // exact
    s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)r2;
  }
  return s;
}

template <bool AFFINE>
__global__ __launch_bounds__(NT, 3) void gemm_nt_bf16x6_kernel(MssConvArgs p, int tiles_per_batch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][A: 3 planes | B: 3 planes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int chunk = tid & 3, row0 = tid >> 2;                             // staging: 4 floats of row row0 (+64)
  const int b = blockIdx.x / tiles_per_batch;
  const int v = mss_xcd_remap(blockIdx.x - b * tiles_per_batch, tiles_per_batch);
  const int mt = v / p.ntiles, nt = v - mt * p.ntiles;
  const int n_it = p.C / BK;

  const float* a_ptr[2];
  const float* b_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = mt * BM + row0 + j * 64;
    row = row < p.M ? row : p.M - 1;                                      // never stored
    a_ptr[j] = p.x + (size_t)b * p.x_bs + (size_t)row * p.ldx + chunk * 4;
    b_ptr[j] = p.w + (size_t)b * p.w_bs + (size_t)(nt * BN + row0 + j * 64) * p.C + chunk * 4;
  }
  // LDS byte offset of this thread's 4-element (8 B) group inside a plane: row, swapped half, quarter
  int st_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = row0 + j * 64;
    st_off[j] = r * ROW_B + (((chunk >> 1) ^ ((r >> 3) & 1)) * 16) + (chunk & 1) * 8;
  }
  // fused BatchNorm + ReLU prologue on X (fp32, before the split): one affine per tile, as in gemm.hip
  const float* s_ptr = nullptr;
  const float* h_ptr = nullptr;
  if (AFFINE) {
    const size_t so = (size_t)((mt * BM) / p.H) * p.in_ss_stride + chunk * 4;   // p.H = rows per affine group
    s_ptr = p.in_scale + so;
    h_ptr = p.in_shift + so;
  }
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();
  f32x4 areg[2], breg[2], sreg, hreg;
  auto issue = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) { areg[j] = *reinterpret_cast<const f32x4*>(a_ptr[j]); breg[j] = *reinterpret_cast<const f32x4*>(b_ptr[j]); }
    if (AFFINE) { sreg = *reinterpret_cast<const f32x4*>(s_ptr); hreg = *reinterpret_cast<const f32x4*>(h_ptr); }
  };
  auto stage = [&](int buf) {
    unsigned char* base = smem + buf * BUF_B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 av = areg[j];
      if (AFFINE) {
        av = av * sreg + hreg;
        av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor); av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
      }
      const Split sa = split3(av), sb = split3(breg[j]);
      *reinterpret_cast<bf16x4*>(base + 0 * PLANE_B + st_off[j]) = sa.hi;
      *reinterpret_cast<bf16x4*>(base + 1 * PLANE_B + st_off[j]) = sa.mid;
      *reinterpret_cast<bf16x4*>(base + 2 * PLANE_B + st_off[j]) = sa.lo;
      *reinterpret_cast<bf16x4*>(base + OPER_B + 0 * PLANE_B + st_off[j]) = sb.hi;
      *reinterpret_cast<bf16x4*>(base + OPER_B + 1 * PLANE_B + st_off[j]) = sb.mid;
      *reinterpret_cast<bf16x4*>(base + OPER_B + 2 * PLANE_B + st_off[j]) = sb.lo;
    }
  };
  // fragment: lane = (row l % 32, k-block l / 32); tile rows are l % 32 + multiples of 32 (bit 3 unchanged)
  const int frow = lane & 31, fkb = lane >> 5;
  const int fr_off = frow * ROW_B + ((fkb ^ ((frow >> 3) & 1)) * 16);
  const int fa_off = (wm * WTM) * ROW_B + fr_off, fb_off = OPER_B + (wn * WTN) * ROW_B + fr_off;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue();
  stage(0);
  __syncthreads();
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    if (it + 1 < n_it) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { a_ptr[j] += BK; b_ptr[j] += BK; }
      if (AFFINE) { s_ptr += BK; h_ptr += BK; }
      issue();
    }
    const unsigned char* base = smem + buf * BUF_B;
    bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE_B + fa_off + i * 32 * ROW_B);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE_B + fb_off + j * 32 * ROW_B);
    }
    // smallest terms first: lo*hi, hi*lo, mid*mid, then mid*hi, hi*mid, then hi*hi
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[PB[t]][j], acc[i][j], 0, 0, 0);
    if (it + 1 < n_it) stage(buf ^ 1);
    __syncthreads();
  }
  mss_epilogue_store<TM, TN>(acc, p, p.y + (size_t)b * p.y_bs, mt * BM + wm * WTM, nt * BN + wn * WTN, lane);
}

}  // namespace

// Called by mss_gemm_nt_dispatch (gemm.hip) when MSS_GEMM_BF16X6=1; p.H / mtiles / ntiles are set as for gemm_nt_kernel.
int mss_gemm_nt_bf16x6_launch(MssConvArgs p, void* stream) {
  const int batch = p.batch > 1 ? p.batch : 1;
  const int tiles_per_batch = p.mtiles * p.ntiles;
  const size_t smem = 2 * BUF_B;
  const dim3 grid((unsigned)(tiles_per_batch * batch));
  if (p.in_scale)
    hipLaunchKernelGGL(gemm_nt_bf16x6_kernel<true>, grid, dim3(NT), smem, static_cast<hipStream_t>(stream), p, tiles_per_batch);
  else
    hipLaunchKernelGGL(gemm_nt_bf16x6_kernel<false>, grid, dim3(NT), smem, static_cast<hipStream_t>(stream), p, tiles_per_batch);
  return mss_launch_status();
}
