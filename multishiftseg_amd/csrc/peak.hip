// Calibration kernels: what THIS device sustains for the two rooflines the path is priced against.
//   mss_peak_mfma_f32  : back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators, operands
//                        in registers, one or two waves per SIMD -> achievable fp32 matrix rate
//   mss_peak_stream_f32: float4 copy of a large buffer -> achievable HBM bandwidth
// bench.py reports them next to the datasheet peaks (157.3 TFLOP/s, 8 TB/s).
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

__global__ __launch_bounds__(256) void peak_mfma_kernel(float* __restrict__ out, int iters, float seed) {
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
  float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 * 0.5f, b0 = seed - threadIdx.x * 1e-3f, b1 = b0 * 0.25f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bf16 matrix rate under load (round 5, the split-bf16 GEMM route's ceiling): register-only loops on pseudo-random operands (the chip
// lowers its clock under bf16 MFMA load on random data, MI355X_MICROARCH.md "DVFS give-back": the datasheet 2.5 PFLOP/s is not what a
// loop can deliver). SHAPE 0: v_mfma_f32_32x32x16_bf16 on the 64 x 128 wave tile of gemm_nt_bf16x3_kernel (2 x 4 blocks, 6 products each
// = 48 MFMAs per K-step); SHAPE 1: v_mfma_f32_16x16x32_bf16 on the same tile (4 x 8 blocks, 3 K=32 products each = 96 MFMAs).
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pk_bf16x8 peak_rand_frag(unsigned& st) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st = st * 1664525u + 1013904223u;
    // two bf16 in [-2, 2) with random mantissas: sign | exponent 0x3f / 0x3e.. | 7 random bits
    const unsigned lo = ((st >> 3) & 0x807fu) | 0x3f00u, hi = ((st >> 11) & 0x807fu) | 0x3e80u;
    w[i] = lo | (hi << 16);
  }
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(pk_bf16x8, u4{w[0], w[1], w[2], w[3]});
}
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void peak_mfma_bf16_kernel(float* __restrict__ out, int iters, unsigned seed) {
  unsigned st = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float s = 0.f;
  if (SHAPE == 0) {
    pk_bf16x8 a[3][2], b[3][4];
    f32x16 acc[2][4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      for (int i = 0; i < 2; ++i) a[p][i] = peak_rand_frag(st);
      for (int j = 0; j < 4; ++j) b[p][j] = peak_rand_frag(st);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int PA[6] = {2, 1, 0, 0, 1, 0}, PB[6] = {0, 0, 0, 1, 1, 2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    pk_bf16x8 a[2][4], b[3][8];
    pk_f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[0][i] = peak_rand_frag(st); a[1][i] = peak_rand_frag(st); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { b[0][j] = peak_rand_frag(st); b[1][j] = peak_rand_frag(st); b[2][j] = peak_rand_frag(st); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = pk_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t == 2 ? 1 : 0][i], b[t][j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// variant 0: one 16-B load and store per thread and iteration (round 1: 4.8 TB/s)
// variant 1: four independent loads in flight per thread before the four stores
// variant 2: as 1 with nontemporal loads and stores (streaming data is read and written once)
// variant 3: as 2, a workgroup owns contiguous 16-KB chunks (4 consecutive float4 per lane and pass)
// variant 4 / 5 / 6 (r06): as 3 but WRITE-only (4 n bytes) / READ-only (4 n bytes) / the first half of src read and all of dst written
//   (6 n bytes, 1 : 2 -- the direction of the Winograd input transforms, which write 1.78-2.25 bytes per byte read)
template <int VARIANT>
__global__ __launch_bounds__(256) void peak_stream_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                          long long n4) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (VARIANT == 0) {
    for (; i < n4; i += stride) dst[i] = src[i];
    return;
  }
  if (VARIANT == 3) {
    const long long chunk = 256 * 4;
    for (long long c = (long long)blockIdx.x * chunk; c + chunk <= n4; c += (long long)gridDim.x * chunk) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(&src[c + u * 256 + threadIdx.x]);
#pragma unroll
      for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], &dst[c + u * 256 + threadIdx.x]);
    }
    const long long done = n4 / chunk * chunk;
    for (i = done + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
    return;
  }
  if (VARIANT >= 4) {          // r06: the two directions apart, and the Winograd input transforms' mix (contiguous 16-KB chunks as 3)
    const long long chunk = 256 * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const long long lim = VARIANT == 6 ? n4 / 2 : n4;       // 6: the first half of src is read, all of dst written (1 : 2)
    for (long long c = (long long)blockIdx.x * chunk; c + chunk <= lim; c += (long long)gridDim.x * chunk) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = VARIANT == 4 ? f32x4{1.f, 2.f, 3.f, (float)c} : __builtin_nontemporal_load(&src[c + u * 256 + threadIdx.x]);
      if (VARIANT == 5) {
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += v[u];
        continue;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], &dst[c + u * 256 + threadIdx.x]);
      if (VARIANT == 6) {
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], &dst[n4 / 2 + c + u * 256 + threadIdx.x]);
      }
    }
    if (VARIANT == 5 && sum.x + sum.y + sum.z + sum.w == 12345.678f) dst[0] = sum;     // (keeps the loads alive)
    return;
  }
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = VARIANT == 2 ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (VARIANT == 2) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

// Layout experiment for the Winograd transforms: every thread reads ONE float4 (coalesced) and writes NS float4, one
// into each of NS "position slabs". blocked = 0: slab s of item i at dst[s * n4 + i] (the [P][T][C] layout: NS write
// streams tens of MB apart per wave); blocked = 1: items are grouped in blocks of `blk` float4 and the NS slabs of a
// block are adjacent, dst[(i / blk) * NS * blk + s * blk + i % blk] (the [T/128][P][128][C] layout: all stores of a
// workgroup land within NS * blk * 16 bytes).
template <int NS>
__global__ __launch_bounds__(256) void peak_scatter_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                           long long n4, int blocked, long long blk) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = src[i];
    const long long base = blocked ? (i / blk) * NS * blk + (i % blk) : i;
    const long long step = blocked ? blk : n4;
#pragma unroll
    for (int s = 0; s < NS; ++s) dst[base + s * step] = v * (float)(s + 1);
  }
}

}  // namespace

extern "C" {

// reads n floats, writes ns * n floats (ns = 16 or 36); bytes moved = 4 * n * (1 + ns). dst: ns * n floats.
int mss_peak_scatter_f32(const float* src, float* dst, long long n, int ns, int blocked, long long blk_floats,
                         void* stream) {
  if (!src || !dst || n <= 0 || n % 4 || (ns != 16 && ns != 36) || blk_floats % 4 || blk_floats <= 0) return MSS_ERR_BAD_ARG;
  if (blocked && n % blk_floats) return MSS_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const f32x4* a = reinterpret_cast<const f32x4*>(src);
  f32x4* b = reinterpret_cast<f32x4*>(dst);
  if (ns == 16) hipLaunchKernelGGL(peak_scatter_kernel<16>, dim3(2048), dim3(256), 0, s, a, b, n / 4, blocked, blk_floats / 4);
  else hipLaunchKernelGGL(peak_scatter_kernel<36>, dim3(2048), dim3(256), 0, s, a, b, n / 4, blocked, blk_floats / 4);
  return mss_launch_status();
}

// out: at least blocks*256 floats. FLOPs performed = blocks * 4 waves * iters * 16 MFMAs * 4096.
int mss_peak_mfma_f32(float* out, int blocks, int iters, void* stream) {
  if (!out || blocks <= 0 || iters <= 0) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(peak_mfma_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, iters, 1.0f);
  return mss_launch_status();
}

// Clock calibration: one wave spins until s_memrealtime has advanced by `ticks`; out[0] = ticks seen, out[1] = s_memtime ticks in the
// same span. With the launch timed from the host this gives the frequency of BOTH counters (tools/peaks.py: realtime_MHz, and what
// s_memtime counts on an otherwise idle chip).
__global__ void peak_clock_kernel(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r = r0;
  while (r - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r = __builtin_amdgcn_s_memrealtime(); }
  if (threadIdx.x == 0) { out[0] = r - r0; out[1] = __builtin_amdgcn_s_memtime() - t0; }
}
int mss_peak_clock(unsigned long long* out, unsigned long long ticks, void* stream) {
  if (!out || ticks == 0) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(peak_clock_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, ticks);
  return mss_launch_status();
}

// out: at least blocks*256 floats. bf16 FLOPs performed = blocks * 4 waves * iters * 48 * 32768 (shape 0: 32x32x16) or
// blocks * 4 * iters * 96 * 16384 (shape 1: 16x16x32) -- the same per iteration.
int mss_peak_mfma_bf16(float* out, int blocks, int iters, int shape, void* stream) {
  if (!out || blocks <= 0 || iters <= 0 || shape < 0 || shape > 1) return MSS_ERR_BAD_ARG;
  if (shape == 0) hipLaunchKernelGGL(peak_mfma_bf16_kernel<0>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, iters, 12345u);
  else hipLaunchKernelGGL(peak_mfma_bf16_kernel<1>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, iters, 12345u);
  return mss_launch_status();
}

// copies n floats (n % 8 == 0); bytes moved = 8 * n for variants 0-3, 4 n / 4 n / 6 n for 4 / 5 / 6: see peak_stream_kernel (tools/peaks.py).
int mss_peak_stream_f32(const float* src, float* dst, long long n, int variant, void* stream) {
  if (!src || !dst || n <= 0 || n % 8 || variant < 0 || variant > 6) return MSS_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const f32x4* a = reinterpret_cast<const f32x4*>(src);
  f32x4* b = reinterpret_cast<f32x4*>(dst);
  const dim3 grid(256 * 8), block(256);
  switch (variant) {
    case 0: hipLaunchKernelGGL(peak_stream_kernel<0>, grid, block, 0, s, a, b, n / 4); break;
    case 1: hipLaunchKernelGGL(peak_stream_kernel<1>, grid, block, 0, s, a, b, n / 4); break;
    case 2: hipLaunchKernelGGL(peak_stream_kernel<2>, grid, block, 0, s, a, b, n / 4); break;
    case 3: hipLaunchKernelGGL(peak_stream_kernel<3>, grid, block, 0, s, a, b, n / 4); break;
    case 4: hipLaunchKernelGGL(peak_stream_kernel<4>, grid, block, 0, s, a, b, n / 4); break;
    case 5: hipLaunchKernelGGL(peak_stream_kernel<5>, grid, block, 0, s, a, b, n / 4); break;
    default: hipLaunchKernelGGL(peak_stream_kernel<6>, grid, block, 0, s, a, b, n / 4); break;
  }
  return mss_launch_status();
}

}  // extern "C"
