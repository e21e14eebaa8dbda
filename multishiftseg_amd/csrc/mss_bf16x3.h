// Layout of the split-bf16 weight planes (gemm_bf16x3.hip copies them into LDS as they lie) and the three-term split itself, shared by
// the GEMM kernels and by the producers of planes (split_weights_kernel, wino_pack_split_kernel).
#pragma once
#include "mss_common.h"

// gemm_bf16x3.hip: allocate the ticket-counter pool of the split kernels on the current device (idempotent; a no-op while `stream` is
// being captured). Every producer of weight planes calls it, so the pool exists before anything can launch on those planes.
void mss_sched_init(hipStream_t stream);

namespace mss_bf16x3 {

constexpr int BK = 16;
constexpr int ROW_B = BK * 2;                 // bytes per row per plane
constexpr int PLANE = 128 * ROW_B;            // 4 KB: one plane of a 128-row operand block
constexpr int OPER = 3 * PLANE;               // 12 KB: hi, mid, lo
// Rows are stored as they are -- 32 bytes, k 0..7 then k 8..15 -- with NO half swap (round 6; round 5 swapped the two 16-byte halves
// where bit 3 of the row was set, which the 32-row fragment of v_mfma_f32_32x32x16_bf16 needs). ds_read_b128 is served in four groups
// of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32; MI355X_MICROARCH.md, LDS); the 16-row fragment of
// v_mfma_f32_16x16x32_bf16 (lane = row l & 15, half (l >> 4) & 1) puts each group on sixteen distinct 16-byte slots of the 256-byte
// bank row exactly when the image is linear (measured with the bit-3 swap: every fragment read two-way conflicted, 470 M conflict
// cycles per launch). The TN weight-gradient kernel stages both of its operands itself and keeps the bit-3 swap for its 32-row fragments.

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {        // v_cvt_pk_bf16_f32: round to nearest even, low half = a
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (a, b) -> packed hi / mid / lo bf16 pairs; the residuals are exact in fp32 (Sterbenz / aligned-exponent subtraction)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(ra, rb);
  const float qa = ra - __uint_as_float(mid << 16), qb = rb - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(qa, qb);
}
// byte offset of the 16 bytes (8 consecutive k, k0 % 8 == 0) of row n of batch entry b in plane 0; + PLANE / + 2 PLANE: mid / lo
__device__ __forceinline__ size_t plane_chunk_offset(long long b, int Kpad, int nk, int n, int k0) {
  const int s = k0 >> 4, h = (k0 >> 3) & 1;
  return ((size_t)(b * (Kpad / 128) + n / 128) * nk + s) * OPER + (size_t)(n & 127) * ROW_B + (size_t)(h * 16);
}

}  // namespace mss_bf16x3
