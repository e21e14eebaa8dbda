// Shared device/host helpers for the MI355X (gfx950) kernels of multishiftseg_amd.
// Everything here is written for wave64 / CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#define MSS_OK 0
#define MSS_ERR_BAD_ARG 1001     // a precondition of the C-ABI entry point was violated
#define MSS_ERR_UNSUPPORTED 1002 // shape outside what the kernels are built for

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Every launcher ends with this: kernel-launch failures are returned, never printf'd
// (the reference only printf'd them: ms_deform_im2col_cuda.cuh:953-957).
static inline int mss_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MSS_OK : (int)e;
}

// Tuning / test switches (MSS_* environment variables) read on launch paths: each call site caches its value and re-reads the
// environment only after mss_env_reset() (include/mss_hip.h) has bumped the generation -- no getenv() per kernel launch.
// A process that sets its switches before the first call needs nothing; tests that flip them call the reset hook.
extern "C" int mss_env_generation(void);
static inline int mss_env_lookup(const char* name, int dflt, std::atomic<long long>& slot) {
  const int gen = mss_env_generation();
  const long long v = slot.load(std::memory_order_relaxed);
  if ((int)(v >> 32) == gen) return (int)(unsigned)(v & 0xffffffffll);
  const char* e = getenv(name);
  const int val = e ? atoi(e) : dflt;
  slot.store(((long long)gen << 32) | (unsigned)val, std::memory_order_relaxed);
  return val;
}
#define MSS_ENV_INT(name, dflt) \
  ([]() -> int { static std::atomic<long long> slot{-(1ll << 32)}; return mss_env_lookup(name, dflt, slot); }())

__host__ __device__ static inline int mss_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Blocks b and b+8 share an XCD (and its L2) under round-robin dispatch. Map the hardware
// block id to a virtual id so that every XCD walks one contiguous chunk of the tile grid.
// Bijective for any grid size (guide: "XCD swizzle must be bijective").
__device__ __forceinline__ int mss_xcd_remap(int bid, int nwg) {
  int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// DPP add across lanes, used for reductions inside 8/16/32/64-lane groups without LDS.
template <int CTRL>
__device__ __forceinline__ float mss_dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(t);
}
// sum over aligned groups of 8 consecutive lanes; every lane of the group gets the sum
__device__ __forceinline__ float mss_sum8(float v) {
  v = mss_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = mss_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = mss_dpp_add<0x141>(v);  // row_half_mirror: lane i <-> 7-i inside each 8-lane half row
  return v;
}
// sum over the whole wave (64 lanes); every lane gets the sum
__device__ __forceinline__ float mss_wave_sum(float v) {
  v = mss_sum8(v);
  v = mss_dpp_add<0x140>(v);  // row_mirror: lane i <-> 15-i inside each 16-lane row
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double mss_wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float mss_wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- bilinear, align_corners=True: same arithmetic as ATen's upsample_bilinear2d (area_pixel_compute_source_index with
// align_corners): src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1-l1.
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap ac_tap(int o, float scale, int in) {
  Tap t;
  // ATen rounds scale * o to fp32 first (area_pixel_compute_source_index) and subtracts afterwards. Under the default
  // -ffp-contract=fast the product would be fused into the subtraction below (fma(scale, o, -i0): more accurate, but
  // it moves the weights by up to 3e-5 at o ~ 700 and the interpolated values by 2e-4); the empty asm pins the rounded
  // product in a register (__fmul_rn is a plain multiply in HIP and does not prevent the contraction)
  float src = scale * (float)o;
  asm volatile("" : "+v"(src));
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}
// The four-tap blend with a FIXED contraction -- top = fma(lx0, v00, lx1*v01), bottom likewise, out = fma(ly0, top, ly1*bottom) --
// so that every kernel interpolating the same map (upsample_ac_kernel, the Winograd input transform that upsamples on the fly)
// produces the same bits; left to -ffp-contract the compiler picks a different fusion per call site.
__device__ __forceinline__ f32x4 mss_bilerp(const Tap& ty, const Tap& tx, f32x4 v00, f32x4 v01, f32x4 v10, f32x4 v11) {
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float top = __builtin_fmaf(tx.l0, v00[k], tx.l1 * v01[k]);
    const float bot = __builtin_fmaf(tx.l0, v10[k], tx.l1 * v11[k]);
    o[k] = __builtin_fmaf(ty.l0, top, ty.l1 * bot);
  }
  return o;
}
static inline float mss_ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
