// Epilogue shared by conv_igemm.hip and gemm.hip: MFMA accumulators (32x32 C/D layout: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) -> NHWC rows, with the optional per-channel affine, residual add
// and ReLU. Written so that the compiler never needs an `s_waitcnt vmcnt(0)` between stores: the first version
// (one guarded element at a time, residual load inside the guard) serialised the 64 stores of a tile on the memory
// round trip -- ~6 K-steps' worth of time per 128x128 tile, i.e. 20-40 % of a short-K tile.
//  * wave-uniform case split: full tile rows + nothing fused -> straight stores;
//  * otherwise all residual loads of a 32x32 sub-tile are issued first (rows clamped, never out of bounds), then the
//    arithmetic, then the guarded stores.
#pragma once
#include "mss_common.h"
#include "../../include/mss_hip.h"

//  * p.stats: the wave also reduces its 64 rows per column (sum, sum of squares of the stored values) and writes them
//    to row group row_base/64 of the partial-sum matrix -- the next layer's train-mode BatchNorm statistics without
//    re-reading the activation (TM * 32 == 64 in every instantiation).
template <int TM, int TN>
__device__ __forceinline__ void mss_epilogue_store(const f32x16 (&acc)[TM][TN], const MssConvArgs& p, float* __restrict__ y,
                                                   int row_base, int col_base, int lane) {
  static_assert(TM * 32 == 64, "statistics row groups are 64 rows");
  const int colq = lane & 31, rowq = 4 * (lane >> 5);
  const bool full_rows = row_base + TM * 32 <= p.M;                       // wave-uniform
  const bool plain = !p.out_scale && !p.res && !p.out_relu && !p.stats;   // kernel-uniform
  float* stats_row = (p.stats && row_base < p.M) ? p.stats + (size_t)(row_base >> 6) * 2 * p.K : nullptr;
  const size_t ldy = (size_t)p.ldy;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col_base + j * 32 + colq;
    if (col < p.K) {
      if (full_rows && plain) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float* yp = y + (size_t)(row_base + i * 32 + rowq) * ldy + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy] = acc[i][j][r];
        }
      } else {
        float osc = 1.f, osh = 0.f;
        if (p.out_scale) { osc = p.out_scale[col]; osh = p.out_shift[col]; }
        const float floor_v = p.out_relu ? 0.f : -__builtin_huge_valf();
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row0 = row_base + i * 32 + rowq;
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = 0.f;
          if (p.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              int row = row0 + (r & 3) + 8 * (r >> 2);
              row = row < p.M ? row : p.M - 1;
              rv[r] = p.res[(size_t)row * p.ldres + col];
            }
          }
          if (full_rows) {                     // no per-element guard: the 16 stores go out back to back
            float* yp = y + (size_t)row0 * ldy + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float lin = acc[i][j][r] * osc + osh;
              const float val = fmaxf(p.res_mask ? (rv[r] > 0.f ? lin : 0.f) : lin + rv[r], floor_v);
              yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy] = val;
              ssum += val; ssq += val * val;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = row0 + (r & 3) + 8 * (r >> 2);
              const float lin = acc[i][j][r] * osc + osh;
              const float val = fmaxf(p.res_mask ? (rv[r] > 0.f ? lin : 0.f) : lin + rv[r], floor_v);
              if (row < p.M) { y[(size_t)row * ldy + col] = val; ssum += val; ssq += val * val; }
            }
          }
        }
        if (stats_row) {                       // lanes l and l + 32 hold the two halves of this column's 64 rows
          ssum += __shfl_xor(ssum, 32);
          ssq += __shfl_xor(ssq, 32);
          if (lane < 32) { stats_row[col] = ssum; stats_row[p.K + col] = ssq; }
        }
      }
    }
  }
}

// The same epilogue for 16x16 accumulators of v_mfma_f32_16x16x32_bf16 issued with the WEIGHTS as the first operand (round 6,
// gemm_bf16x3.hip): D[channel][pixel], lane l holds pixel l & 15 and the four consecutive channels 4 * (l >> 4) .. + 3 -- acc[i][j][r] =
// y[row_base + 16 i + (l & 15)][col_base + 16 j + 4 (l >> 4) + r]. A lane's four values are one 16-byte store (four lanes cover 64
// consecutive bytes of a pixel's row, a wave-instruction 16 pixels), the residual and the per-channel affine 16-byte loads.
// Requires K % 4 == 0, ldy % 4 == 0, 16-byte aligned y (and res / ldres, out_scale / out_shift, stats when given): the caller checks.
// Statistics: as above, per 64-row group and channel, over the rows < M (TI * 16 == 64).
template <int TI, int TJ>
__device__ __forceinline__ void mss_epilogue_store16(const f32x4 (&acc)[TI][TJ], const MssConvArgs& p, float* __restrict__ y,
                                                     int row_base, int col_base, int lane) {
  static_assert(TI * 16 == 64, "statistics row groups are 64 rows");
  const int rl = lane & 15, cq = 4 * (lane >> 4);
  const bool plain = !p.out_scale && !p.res && !p.out_relu && !p.stats;   // kernel-uniform
  const bool full_rows = row_base + TI * 16 <= p.M;                       // wave-uniform
  float* stats_row = (p.stats && row_base < p.M) ? p.stats + (size_t)(row_base >> 6) * 2 * p.K : nullptr;
  const float floor_v = p.out_relu ? 0.f : -__builtin_huge_valf();
  // rows past the end (last row tile only): loads are clamped to row M - 1, the value is masked out of the statistics and its store
  // goes to row M - 1 too -- but only from the lane that OWNS row M - 1 would that be right, so those lanes do not store at all
  const int last = p.M - 1;
  const size_t ldy = (size_t)p.ldy;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = col_base + j * 16 + cq;
    if (col < p.K) {
      if (plain && full_rows) {
        float* yp = y + (size_t)(row_base + rl) * ldy + col;
#pragma unroll
        for (int i = 0; i < TI; ++i) *reinterpret_cast<f32x4*>(yp + (size_t)(i * 16) * ldy) = acc[i][j];
      } else {
        f32x4 osc = {1.f, 1.f, 1.f, 1.f}, osh = {0.f, 0.f, 0.f, 0.f};
        if (p.out_scale) {
          osc = *reinterpret_cast<const f32x4*>(p.out_scale + col);
          osh = *reinterpret_cast<const f32x4*>(p.out_shift + col);
        }
        f32x4 rv[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int row = row_base + i * 16 + rl;
          if (p.res) rv[i] = *reinterpret_cast<const f32x4*>(p.res + (size_t)(row < p.M ? row : last) * p.ldres + col);
        }
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int row = row_base + i * 16 + rl;
          const f32x4 lin = acc[i][j] * osc + osh;
          f32x4 val;
#pragma unroll
          for (int r = 0; r < 4; ++r) val[r] = fmaxf(p.res_mask ? (rv[i][r] > 0.f ? lin[r] : 0.f) : lin[r] + rv[i][r], floor_v);
          if (full_rows || row < p.M) {
            *reinterpret_cast<f32x4*>(y + (size_t)row * ldy + col) = val;
            ssum += val;
            ssq += val * val;
          }
        }
        if (stats_row) {                       // the 16 lanes of a group hold the 64 rows of these four channels
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
              ssum[r] += __shfl_xor(ssum[r], m);
              ssq[r] += __shfl_xor(ssq[r], m);
            }
          }
          if (rl == 0) {
            *reinterpret_cast<f32x4*>(stats_row + col) = ssum;
            *reinterpret_cast<f32x4*>(stats_row + p.K + col) = ssq;
          }
        }
      }
    }
  }
}
