// Multi-scale deformable attention (MSDeformAttn) forward / backward for MI355X, wave64.
//
// Math spec: SURVEY.md Appendix B; reference kernels
//   lib/network/mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh
//     :38-89   bilinear gather        :242-304 forward kernel
//     :92-164  bilinear scatter/grads :306-925 backward kernel family
//
// Design (not a translation of the reference's 1-thread-per-element / 32-thread-block kernels):
//  forward fast path (fp32, D = 4*LPH): LPH lanes own one (n,q,m) pair with float4 channels each,
//    so one wavefront covers 64/LPH consecutive pairs (= all 8 heads of one query at D=32): the
//    pair's 2*L*P locations + L*P weights are staged once per wave through LDS (coalesced 1-KB
//    reads) instead of being re-read by every channel thread; each corner gather is one 16-B load
//    per lane (a full 128-B value row per 8 lanes); the output store is a contiguous 1 KB.
//  backward: LPP (32 or 64) lanes own one pair, lane = channel, so each grad_value atomic
//    wave-instruction covers whole 128-B rows (the shape the memory-side atomic units run at full
//    rate); grad_loc / grad_attn are reduced over the pair's lanes with DPP/shuffles and written
//    with plain stores (no LDS round trip, no serial thread-0 loop, no zero-init needed).
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };

// ------------------------------------------------------------------------------------------
// forward, fast path
template <int LPH>
__global__ __launch_bounds__(256) void msda_fwd_fast_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ attn, long long npairs, int S, int M, int L, int Lq,
    int P, float* __restrict__ out) {
  constexpr int D = 4 * LPH;
  constexpr int HPW = 64 / LPH;  // pairs per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* sloc = smem + wave * (HPW * LP * 3);  // [HPW][LP][2]
  float* sattn = sloc + HPW * LP * 2;          // [HPW][LP]

  const long long pair0 = ((long long)blockIdx.x * 4 + wave) * HPW;
  if (pair0 >= npairs) return;  // whole wave leaves together (no block barrier below)
  const int npw = (int)min((long long)HPW, npairs - pair0);

  // stage loc/attn of the wave's pairs (contiguous in memory)
  {
    const float* gl = loc + pair0 * LP * 2;
    const float* ga = attn + pair0 * LP;
    for (int i = lane; i < npw * LP * 2; i += 64) sloc[i] = gl[i];
    for (int i = lane; i < npw * LP; i += 64) sattn[i] = ga[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int g = lane / LPH, j = lane % LPH;
  if (g >= npw) return;
  const long long pair = pair0 + g;
  const int m = (int)(pair % M);
  const long long nq = pair / M;
  const int n = (int)(nq / Lq);
  const float* myloc = sloc + g * LP * 2;
  const float* myattn = sattn + g * LP;
  const size_t row_stride = (size_t)M * D;  // floats between consecutive spatial positions
  const float* vbase = value + (size_t)n * S * row_stride + (size_t)m * D + 4 * j;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float* vl = vbase + (size_t)starts[l] * row_stride;
#pragma unroll 4
    for (int pt = 0; pt < P; ++pt) {
      const float lx = myloc[(l * P + pt) * 2], ly = myloc[(l * P + pt) * 2 + 1];
      const float aw = myattn[l * P + pt];
      const float w_im = lx * W - 0.5f, h_im = ly * H - 0.5f;
      const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool okh0 = inside && h0 >= 0, okh1 = inside && h0 + 1 <= H - 1;
      const bool okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
      const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
      const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 v1 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h0c * W + w0c) * row_stride);
      f32x4 v2 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h0c * W + w1c) * row_stride);
      f32x4 v3 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h1c * W + w0c) * row_stride);
      f32x4 v4 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h1c * W + w1c) * row_stride);
      v1 = (okh0 && okw0) ? v1 : z;
      v2 = (okh0 && okw1) ? v2 : z;
      v3 = (okh1 && okw0) ? v3 : z;
      v4 = (okh1 && okw1) ? v4 : z;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const f32x4 val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
      acc += aw * val;
    }
  }
  *reinterpret_cast<f32x4*>(out + pair * D + 4 * j) = acc;
}

// ------------------------------------------------------------------------------------------
// forward, generic (any D, fp32/fp64): one thread per output element
template <typename T>
__global__ void msda_fwd_generic_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                        const int64_t* __restrict__ starts, const T* __restrict__ loc,
                                        const T* __restrict__ attn, long long total, int S, int M, int D, int L,
                                        int Lq, int P, T* __restrict__ out) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D);
    const long long pair = idx / D;
    const int m = (int)(pair % M);
    const int n = (int)(pair / M / Lq);
    const T* ploc = loc + pair * L * P * 2;
    const T* pattn = attn + pair * L * P;
    const size_t rs = (size_t)M * D;
    const T* vb = value + (size_t)n * S * rs + (size_t)m * D + d;
    T acc = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* vl = vb + (size_t)starts[l] * rs;
      for (int pt = 0; pt < P; ++pt) {
        const T lx = ploc[(l * P + pt) * 2], ly = ploc[(l * P + pt) * 2 + 1];
        const T aw = pattn[l * P + pt];
        const T w_im = lx * W - (T)0.5, h_im = ly * H - (T)0.5;
        if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
          const T hf = floor(h_im), wf = floor(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const T lh = h_im - hf, lw = w_im - wf, hh = (T)1 - lh, hw = (T)1 - lw;
          T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          if (h0 >= 0 && w0 >= 0) v1 = vl[(size_t)(h0 * W + w0) * rs];
          if (h0 >= 0 && w0 + 1 <= W - 1) v2 = vl[(size_t)(h0 * W + w0 + 1) * rs];
          if (h0 + 1 <= H - 1 && w0 >= 0) v3 = vl[(size_t)((h0 + 1) * W + w0) * rs];
          if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) v4 = vl[(size_t)((h0 + 1) * W + w0 + 1) * rs];
          const T val = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
          acc += aw * val;
        }
      }
    }
    out[idx] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// backward: LPP lanes per (n,q,m) pair, lane = channel (looping when D > LPP)
template <typename T, int LPP>
__device__ __forceinline__ T pair_reduce(T v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int LPP>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const T* __restrict__ loc, const T* __restrict__ attn, const T* __restrict__ gout, long long npairs, int S,
    int M, int D, int L, int Lq, int P, T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gattn) {
  constexpr int PPW = 64 / LPP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane / LPP, j = lane % LPP;
  const long long pair = ((long long)blockIdx.x * 4 + wave) * PPW + g;
  if (pair >= npairs) return;  // LPP-lane groups leave together; shuffles stay inside a group
  const int m = (int)(pair % M);
  const int n = (int)(pair / M / Lq);
  const T* ploc = loc + pair * L * P * 2;
  const T* pattn = attn + pair * L * P;
  const size_t rs = (size_t)M * D;
  const size_t voff = (size_t)n * S * rs + (size_t)m * D;
  const T* go = gout + pair * D;

  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const size_t loff = voff + (size_t)starts[l] * rs;
    for (int pt = 0; pt < P; ++pt) {
      const int sidx = l * P + pt;
      const T lx = ploc[sidx * 2], ly = ploc[sidx * 2 + 1];
      const T aw = pattn[sidx];
      const T w_im = lx * W - (T)0.5, h_im = ly * H - (T)0.5;
      T s_attn = 0, s_w = 0, s_h = 0;
      if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {  // uniform within the pair group
        const T hf = floor(h_im), wf = floor(w_im);
        const int h0 = (int)hf, w0 = (int)wf;
        const T lh = h_im - hf, lw = w_im - wf, hh = (T)1 - lh, hw = (T)1 - lw;
        const bool ok1 = h0 >= 0 && w0 >= 0, ok2 = h0 >= 0 && w0 + 1 <= W - 1;
        const bool ok3 = h0 + 1 <= H - 1 && w0 >= 0, ok4 = h0 + 1 <= H - 1 && w0 + 1 <= W - 1;
        const size_t o1 = loff + (size_t)(h0 * W + w0) * rs, o2 = o1 + rs;
        const size_t o3 = o1 + (size_t)W * rs, o4 = o3 + rs;
        const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        for (int d = j; d < D; d += LPP) {
          const T tg = go[d];
          const T tgv = aw * tg;
          T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          if (ok1) { v1 = value[o1 + d]; atomicAdd(gvalue + o1 + d, w1 * tgv); }
          if (ok2) { v2 = value[o2 + d]; atomicAdd(gvalue + o2 + d, w2 * tgv); }
          if (ok3) { v3 = value[o3 + d]; atomicAdd(gvalue + o3 + d, w3 * tgv); }
          if (ok4) { v4 = value[o4 + d]; atomicAdd(gvalue + o4 + d, w4 * tgv); }
          const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
          const T gw = -hh * v1 + hh * v2 - lh * v3 + lh * v4;
          const T gh = -hw * v1 - lw * v2 + hw * v3 + lw * v4;
          s_attn += tg * val;
          s_w += gw * tgv;
          s_h += gh * tgv;
        }
        s_w *= (T)W;
        s_h *= (T)H;
      }
      s_attn = pair_reduce<T, LPP>(s_attn);
      s_w = pair_reduce<T, LPP>(s_w);
      s_h = pair_reduce<T, LPP>(s_h);
      if (j == 0) {
        gattn[pair * L * P + sidx] = s_attn;
        gloc[(pair * L * P + sidx) * 2] = s_w;
        gloc[(pair * L * P + sidx) * 2 + 1] = s_h;
      }
    }
  }
}

template <typename T>
int msda_check(const T* value, const int64_t* shapes, const int64_t* starts, const T* loc, const T* attn, int N,
               int S, int M, int D, int L, int Lq, int P) {
  if (N < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return MSS_ERR_BAD_ARG;
  if ((long long)N * Lq == 0) return MSS_OK;  // empty query set: nothing is dereferenced
  if (!value || !shapes || !starts || !loc || !attn) return MSS_ERR_BAD_ARG;
  return MSS_OK;
}

template <typename T>
int msda_forward(const T* value, const int64_t* shapes, const int64_t* starts, const T* loc, const T* attn,
                 int N, int S, int M, int D, int L, int Lq, int P, T* out, hipStream_t stream) {
  int rc = msda_check(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  const long long npairs = (long long)N * Lq * M;
  if (npairs == 0) return MSS_OK;
  if (!out) return MSS_ERR_BAD_ARG;
  const long long total = npairs * D;
  int blocks = (int)min((total + 255) / 256, (long long)256 * 32);
  hipLaunchKernelGGL(msda_fwd_generic_kernel<T>, dim3(blocks), dim3(256), 0, stream, value, shapes, starts, loc,
                     attn, total, S, M, D, L, Lq, P, out);
  return mss_launch_status();
}

template <int LPH>
int msda_forward_fast(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc,
                      const float* attn, int N, int S, int M, int L, int Lq, int P, float* out,
                      hipStream_t stream) {
  constexpr int HPW = 64 / LPH;
  const long long npairs = (long long)N * Lq * M;
  const long long nblocks = (npairs + 4 * HPW - 1) / (4 * HPW);
  const size_t smem = (size_t)4 * HPW * L * P * 3 * sizeof(float);
  hipLaunchKernelGGL(msda_fwd_fast_kernel<LPH>, dim3((unsigned)nblocks), dim3(256), smem, stream, value, shapes,
                     starts, loc, attn, npairs, S, M, L, Lq, P, out);
  return mss_launch_status();
}

template <typename T>
int msda_backward(const T* value, const int64_t* shapes, const int64_t* starts, const T* loc, const T* attn,
                  const T* gout, int N, int S, int M, int D, int L, int Lq, int P, T* gvalue, T* gloc, T* gattn,
                  hipStream_t stream) {
  int rc = msda_check(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  if (!gvalue && N > 0) return MSS_ERR_BAD_ARG;
  if (N > 0) {
    hipError_t e = hipMemsetAsync(gvalue, 0, (size_t)N * S * M * D * sizeof(T), stream);
    if (e != hipSuccess) return (int)e;
  }
  const long long npairs = (long long)N * Lq * M;
  if (npairs == 0) return MSS_OK;
  if (!gout || !gloc || !gattn) return MSS_ERR_BAD_ARG;
  if (D <= 32) {
    const long long nblocks = (npairs + 7) / 8;
    hipLaunchKernelGGL((msda_bwd_kernel<T, 32>), dim3((unsigned)nblocks), dim3(256), 0, stream, value, shapes,
                       starts, loc, attn, gout, npairs, S, M, D, L, Lq, P, gvalue, gloc, gattn);
  } else {
    const long long nblocks = (npairs + 3) / 4;
    hipLaunchKernelGGL((msda_bwd_kernel<T, 64>), dim3((unsigned)nblocks), dim3(256), 0, stream, value, shapes,
                       starts, loc, attn, gout, npairs, S, M, D, L, Lq, P, gvalue, gloc, gattn);
  }
  return mss_launch_status();
}

}  // namespace

extern "C" {

int mss_abi_version(void) { return MSS_ABI_VERSION; }

int mss_msda_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const float* sampling_loc, const float* attn_weight, int N, int S, int M, int D, int L,
                         int Lq, int P, float* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = msda_check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  if ((long long)N * Lq * M == 0) return MSS_OK;
  if (!out) return MSS_ERR_BAD_ARG;
  const bool aligned = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const size_t smem_per_lp = (size_t)4 * 3 * sizeof(float) * L * P;
  if (aligned && D == 32 && smem_per_lp * 8 <= 65536)
    return msda_forward_fast<8>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, L,
                                Lq, P, out, s);
  if (aligned && D == 16 && smem_per_lp * 16 <= 65536)
    return msda_forward_fast<4>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, L,
                                Lq, P, out, s);
  if (aligned && D == 64 && smem_per_lp * 4 <= 65536)
    return msda_forward_fast<16>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, L,
                                 Lq, P, out, s);
  return msda_forward<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L,
                             Lq, P, out, s);
}

int mss_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const double* sampling_loc, const double* attn_weight, int N, int S, int M, int D, int L,
                         int Lq, int P, double* out, void* stream) {
  return msda_forward<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L,
                              Lq, P, out, static_cast<hipStream_t>(stream));
}

int mss_msda_backward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, const float* grad_out, int N,
                          int S, int M, int D, int L, int Lq, int P, float* grad_value, float* grad_loc,
                          float* grad_attn, void* stream) {
  return msda_backward<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, N,
                              S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn,
                              static_cast<hipStream_t>(stream));
}

int mss_msda_backward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const double* sampling_loc, const double* attn_weight, const double* grad_out, int N,
                          int S, int M, int D, int L, int Lq, int P, double* grad_value, double* grad_loc,
                          double* grad_attn, void* stream) {
  return msda_backward<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, N,
                               S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn,
                               static_cast<hipStream_t>(stream));
}

}  // extern "C"
