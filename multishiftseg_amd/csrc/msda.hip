// Multi-scale deformable attention (MSDeformAttn) forward / backward for MI355X, wave64.
//
// Math spec: SURVEY.md Appendix B; reference kernels
//   lib/network/mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh
//     :38-89   bilinear gather        :242-304 forward kernel
//     :92-164  bilinear scatter/grads :306-925 backward kernel family
//
// Design (not a translation of the reference's 1-thread-per-element / 32-thread-block kernels). ONE forward and TWO backward
// formulations ship (round 6; the measured losers are in the git history):
//  forward (fp32, D = 4*LPH = 16 / 32 / 64): msda_fwd_rec_kernel, per-sample records (below); anything else: msda_fwd_generic_kernel.
//  backward, default (fp32, D = 32, host copy of the level shapes at hand): grad_value on the BINNED owner-computes path (counting
//    sort of the samples by home tile, 64-bit fixed-point LDS tiles, halo merge: no float atomic, bit-reproducible) + one gather
//    pass for grad_loc / grad_attn (msda_bwd_gather_fast_kernel).
//  backward, generic (fp64, other head dimensions, no host shapes): msda_bwd_kernel -- LPP (32 or 64) lanes own one pair, lane =
//    channel, so each grad_value atomic wave-instruction covers whole 128-B rows (the shape the memory-side atomic units run at full
//    rate); grad_loc / grad_attn are reduced over the pair's lanes with DPP/shuffles and written with plain stores.
#include "mss_common.h"
#include <stdlib.h>
#include "../../include/mss_hip.h"

namespace {

constexpr int MSDA_MAX_LP = 20;       // prepare kernels: 256 pairs x 3 x L*P floats of LDS per workgroup (60 KB at 20)

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };

template <typename T, int LPP>
__device__ __forceinline__ T pair_reduce(T v);     // sum over the LPP lanes of a pair's group (defined with the backward)

// ------------------------------------------------------------------------------------------
// forward (fp32, D = 4 * LPH): per-sample RECORDS. LPH lanes own one (n, q, m) pair with float4 channels each, so one wavefront
// covers 64 / LPH consecutive pairs (all 8 heads of one query at D = 32); each corner gather is one 16-byte load per lane (a full
// 128-byte value row per 8 lanes) through ONE buffer resource (32-bit offsets; an out-of-image corner is an out-of-range offset the
// hardware answers with zeros: the reference's zero padding, no selects); the output store is a contiguous 1 KB.
// The lanes of a group split the L*P samples between them ONCE: lane j prepares samples j, j + LPH, ... -- it loads their raw values
// itself, for the FUSED form (`loc` / `attn` are the raw sampling offsets / attention logits of the module's two Linears and `ref` the
// reference points [N,Lq,L,2], ops/modules/ms_deform_attn.py:100-109) takes part in the group's softmax through shuffles and does
// loc = ref + offset / (W_l, H_l), and leaves a record in LDS: four byte offsets into the value tensor, the four bilinear weights,
// the attention weight. The sampling loop then costs, per sample and lane, three LDS reads (group-uniform addresses: broadcasts),
// four address adds and the 20 multiply-adds of the accumulation: 16 VALU per sample where round 2's kernel (every lane repeating
// the location arithmetic: ~40) was VALU-issue-bound next to the L2 gather ceiling. That kernel and round 2's LDS-window forward
// (0.7x the gather's speed) left the product in round 6 (git history).
template <int LPH, bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_rec_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ ref, long long npairs, int S,
    int M, int L, int Lq, int P, float* __restrict__ out, long long ldo, long long ldl, int LPpad,
    float* __restrict__ loc_out = nullptr, float* __restrict__ attn_out = nullptr) {
  constexpr int D = 4 * LPH;
  constexpr int HPW = 64 / LPH;  // pairs per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // per wave: offsets [HPW][LPpad] x 4 words, weights [HPW][LPpad] x 4 words, attention [HPW][LPpad]; LPpad = L*P + 1 makes the
  // pair stride 4 * LPpad words, which spreads the eight groups' 16-byte reads over distinct banks for L*P = 12
  const int pstride = 4 * LPpad;
  unsigned* soff = reinterpret_cast<unsigned*>(smem) + wave * (HPW * LPpad * 9);
  float* swgt = reinterpret_cast<float*>(soff) + HPW * pstride;
  float* satt = swgt + HPW * pstride;

  const long long pair0 = ((long long)blockIdx.x * 4 + wave) * HPW;
  if (pair0 >= npairs) return;  // whole wave leaves together (no block barrier below)
  const int npw = (int)min((long long)HPW, npairs - pair0);
  const int g = lane / LPH, j = lane % LPH;
  const bool mine = g < npw;
  const long long pair = pair0 + (mine ? g : 0);
  const int m = (int)(pair % M);
  const long long nq = pair / M;
  const int n = (int)(nq / Lq);
  const size_t row_stride = (size_t)M * D;
  const unsigned rs4 = (unsigned)(row_stride * sizeof(float));
  const unsigned pair_off = (unsigned)(((size_t)n * S * row_stride + (size_t)m * D) * sizeof(float));
  const float* gl = loc + nq * ldo + (size_t)m * (LP * 2);
  const float* ga = attn + nq * ldl + (size_t)m * LP;

  // ---- softmax of the group's logits (fused form): lane j holds logits j, j + LPH, ...
  constexpr int MAXI = (20 + LPH - 1) / LPH;          // L*P <= 20 in the fused form (checked by the launcher)
  float inv = 1.f, mx = 0.f;
  if (FUSED) {
    mx = -__builtin_huge_valf();
    float lg[MAXI];
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
      const int i = j + t * LPH;
      lg[t] = (mine && i < LP) ? ga[i] : -__builtin_huge_valf();
      mx = fmaxf(mx, lg[t]);
    }
#pragma unroll
    for (int o = LPH / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
      const int i = j + t * LPH;
      part += (mine && i < LP) ? expf(lg[t] - mx) : 0.f;
    }
    inv = 1.f / pair_reduce<float, LPH>(part);
  }
  // ---- records of this lane's samples
  if (mine) {
    for (int i = j; i < LP; i += LPH) {
      const int l = i / P;
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      float lx = gl[2 * i], ly = gl[2 * i + 1], aw = ga[i];
      if (FUSED) {
        aw = expf(aw - mx) * inv;
        lx = ref[(nq * L + l) * 2] + lx / (float)W;
        ly = ref[(nq * L + l) * 2 + 1] + ly / (float)H;
        if (loc_out) {        // training: the backward wants exactly these locations / weights (dense [N,Lq,M,L,P,2] / [N,Lq,M,L,P])
          loc_out[(pair * LP + i) * 2] = lx;
          loc_out[(pair * LP + i) * 2 + 1] = ly;
          attn_out[pair * LP + i] = aw;
        }
      }
      const float w_im = lx * W - 0.5f, h_im = ly * H - 0.5f;
      const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool okh0 = inside && h0 >= 0, okh1 = inside && h0 + 1 <= H - 1;
      const bool okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
      const unsigned o00 = pair_off + (unsigned)starts[l] * rs4 + (unsigned)(h0 * W + w0) * rs4;
      const unsigned oob = 0xffffff00u;                    // + 16 * j stays out of range (host: the tensor is below that)
      uint4 o;
      o.x = (okh0 && okw0) ? o00 : oob;
      o.y = (okh0 && okw1) ? o00 + rs4 : oob;
      o.z = (okh1 && okw0) ? o00 + W * rs4 : oob;
      o.w = (okh1 && okw1) ? o00 + W * rs4 + rs4 : oob;
      *reinterpret_cast<uint4*>(soff + g * pstride + 4 * i) = o;
      const f32x4 w = {hh * hw, hh * lw, lh * hw, lh * lw};
      *reinterpret_cast<f32x4*>(swgt + g * pstride + 4 * i) = w;
      satt[g * LPpad + i] = aw;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (!mine) return;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value), 0, (int)(unsigned)min((unsigned long long)npairs / M / Lq * S * row_stride * 4ull, 0xffffffffull),
      0x00020000);
  const unsigned j16 = 16u * (unsigned)j;
  const unsigned* mo = soff + g * pstride;
  const float* mw = swgt + g * pstride;
  const float* ma = satt + g * LPpad;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = 0; i < LP; ++i) {
    const uint4 o = *reinterpret_cast<const uint4*>(mo + 4 * i);
    const f32x4 w = *reinterpret_cast<const f32x4*>(mw + 4 * i);
    const float aw = ma[i];
    const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + j16, 0, 0));
    const f32x4 v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + j16, 0, 0));
    const f32x4 v3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + j16, 0, 0));
    const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + j16, 0, 0));
    const f32x4 val = w.x * v1 + w.y * v2 + w.z * v3 + w.w * v4;
    acc += aw * val;
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

template <typename T, int LPP, bool VALUE_GRAD>
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
          if (ok1) { v1 = value[o1 + d]; if (VALUE_GRAD) atomicAdd(gvalue + o1 + d, w1 * tgv); }
          if (ok2) { v2 = value[o2 + d]; if (VALUE_GRAD) atomicAdd(gvalue + o2 + d, w2 * tgv); }
          if (ok3) { v3 = value[o3 + d]; if (VALUE_GRAD) atomicAdd(gvalue + o3 + d, w3 * tgv); }
          if (ok4) { v4 = value[o4 + d]; if (VALUE_GRAD) atomicAdd(gvalue + o4 + d, w4 * tgv); }
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

// grad_sampling_loc / grad_attn_weight without touching grad_value (fp32, D = 32): the forward kernel's layout -- a wave
// holds 8 (query, head) pairs x 8 channel quads, loc/attn staged in LDS -- so the four corner rows of a sample are
// float4 gathers and the per-sample work is 4 dot products over 32 channels (4 FMAs + a 3-step DPP sum in each 8-lane
// group): everything the two gradients need is linear in p_c = <value_corner_c, grad_out>. Results overwrite the
// staged loc/attn in LDS and leave with coalesced stores.
template <bool BUF>
__global__ __launch_bounds__(256) void msda_bwd_gather_fast_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ gout, long long npairs, int S,
    int M, int L, int Lq, int P, float* __restrict__ gloc, float* __restrict__ gattn, float* __restrict__ goff = nullptr,
    long long ldo = 0, float* __restrict__ glog = nullptr, long long ldl = 0) {
  constexpr int LPH = 8, D = 32, HPW = 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // PROJ (goff != NULL, r04): the module's backward through the softmax and the location arithmetic
  // (ops/modules/ms_deform_attn.py:100-109) happens here, on the gradients this wave holds in LDS anyway -- what leaves is
  // d(offsets) = g_loc / (W_l, H_l) and d(logits) = attn * (g_attn - sum(attn * g_attn)), written straight into the (strided)
  // buffer of the two projections' output gradients; grad_loc / grad_attn are never materialised and msda_prepare_bwd_kernel
  // does not run. Same expressions in the same order as that kernel: the same bits.
  const bool proj = goff != nullptr;
  float* sloc = smem + wave * (HPW * LP * 4);  // [HPW][LP][2]
  float* sattn = sloc + HPW * LP * 2;          // [HPW][LP]
  float* skeep = sattn + HPW * LP;             // [HPW][LP]: the attention weights (sattn is overwritten with their gradients)
  const long long pair0 = ((long long)blockIdx.x * 4 + wave) * HPW;
  if (pair0 >= npairs) return;  // whole wave leaves together (no block barrier below)
  const int npw = (int)min((long long)HPW, npairs - pair0);
  {
    const float* gl = loc + pair0 * LP * 2;
    const float* ga = attn + pair0 * LP;
    for (int i = lane; i < npw * LP * 2; i += 64) sloc[i] = gl[i];
    for (int i = lane; i < npw * LP; i += 64) { const float a = ga[i]; sattn[i] = a; skeep[i] = a; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int g = lane / LPH, j = lane % LPH;
  const bool live = g < npw;                                   // idle groups run along on pair0 (DPP sums stay in-group)
  const long long pair = pair0 + (live ? g : 0);
  const int m = (int)(pair % M);
  const int n = (int)(pair / M / Lq);
  float* myloc = sloc + (live ? g : 0) * LP * 2;
  float* myattn = sattn + (live ? g : 0) * LP;
  const size_t row_stride = (size_t)M * D;
  const float* vbase = value + (size_t)n * S * row_stride + (size_t)m * D + 4 * j;
  const f32x4 go4 = *reinterpret_cast<const f32x4*>(gout + pair * D + 4 * j);
  auto dot8 = [&](f32x4 v) { return mss_sum8(v.x * go4.x + v.y * go4.y + v.z * go4.z + v.w * go4.w); };
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value), 0, (int)(unsigned)min((unsigned long long)npairs / M / Lq * S * row_stride * 4ull, 0xffffffffull), 0x00020000);
  const unsigned rs4 = (unsigned)(row_stride * sizeof(float));
  const unsigned lane_off = (unsigned)(((size_t)n * S * row_stride + (size_t)m * D + 4 * j) * sizeof(float));
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float* vl = vbase + (size_t)starts[l] * row_stride;
    for (int pt = 0; pt < P; ++pt) {
      const int sidx = l * P + pt;
      const float lx = myloc[sidx * 2], ly = myloc[sidx * 2 + 1];
      const float aw = myattn[sidx];
      const float w_im = lx * W - 0.5f, h_im = ly * H - 0.5f;
      const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool okh0 = inside && h0 >= 0, okh1 = inside && h0 + 1 <= H - 1;
      const bool okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
      f32x4 v1, v2, v3, v4;
      if (BUF) {          // one buffer resource for the value tensor: 32-bit corner offsets, zeros for out-of-image corners
        const unsigned o00 = lane_off + (unsigned)starts[l] * rs4 + (unsigned)(h0 * W + w0) * rs4, oob = 0xffffffffu;
        v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh0 && okw0) ? o00 : oob, 0, 0));
        v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh0 && okw1) ? o00 + rs4 : oob, 0, 0));
        v3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh1 && okw0) ? o00 + W * rs4 : oob, 0, 0));
        v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh1 && okw1) ? o00 + W * rs4 + rs4 : oob, 0, 0));
      } else {
        const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
        const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v1 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h0c * W + w0c) * row_stride);
        v2 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h0c * W + w1c) * row_stride);
        v3 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h1c * W + w0c) * row_stride);
        v4 = *reinterpret_cast<const f32x4*>(vl + (size_t)(h1c * W + w1c) * row_stride);
        v1 = (okh0 && okw0) ? v1 : z;
        v2 = (okh0 && okw1) ? v2 : z;
        v3 = (okh1 && okw0) ? v3 : z;
        v4 = (okh1 && okw1) ? v4 : z;
      }
      const float p1 = dot8(v1), p2 = dot8(v2), p3 = dot8(v3), p4 = dot8(v4);
      const float s_attn = hh * hw * p1 + hh * lw * p2 + lh * hw * p3 + lh * lw * p4;
      const float s_w = aw * (float)W * (-hh * p1 + hh * p2 - lh * p3 + lh * p4);
      const float s_h = aw * (float)H * (-hw * p1 - lw * p2 + hw * p3 + lw * p4);
      if (live && j == 0) {
        // PROJ: d(offset) = g_loc / (W_l, H_l) right here, where the level is known (the same division the prepare-backward kernel does)
        myloc[sidx * 2] = proj ? s_w / (float)W : s_w;
        myloc[sidx * 2 + 1] = proj ? s_h / (float)H : s_h;
        myattn[sidx] = s_attn;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (proj) {
    if (live) {
      const float* a = skeep + g * LP;
      float* ga = sattn + g * LP;
      float dot = 0.f;
      for (int i = 0; i < LP; ++i) dot = __builtin_fmaf(a[i], ga[i], dot);     // every lane of the group: the same ascending sum
      float d[3];                                                              // L*P <= 20 (host check): <= 3 samples per lane
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int i = j + t * LPH;
        d[t] = i < LP ? a[i] * (ga[i] - dot) : 0.f;
      }
      // every lane of the group has read all of ga[] (for `dot`) before any lane overwrites its entries: the same
      // release / wave barrier / acquire the other phases of this kernel put between LDS reads and writes (lockstep
      // execution and LDS op order must not be what makes this correct)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int i = j + t * LPH;
        if (i < LP) ga[i] = d[t];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (M == HPW) {            // the wave's pairs are the M heads of ONE query: its row of the buffer is contiguous
      float* ro = goff + pair0 / M * ldo;
      float* rl = glog + pair0 / M * ldl;
      for (int i = lane; i < npw * LP * 2; i += 64) ro[i] = sloc[i];
      for (int i = lane; i < npw * LP; i += 64) rl[i] = sattn[i];
      return;
    }
    for (int i = lane; i < npw * LP * 2; i += 64) {
      const long long pr = pair0 + i / (2 * LP);
      goff[pr / M * ldo + (pr % M) * (2 * LP) + i % (2 * LP)] = sloc[i];
    }
    for (int i = lane; i < npw * LP; i += 64) {
      const long long pr = pair0 + i / LP;
      glog[pr / M * ldl + (pr % M) * LP + i % LP] = sattn[i];
    }
    return;
  }
  {
    float* gl = gloc + pair0 * LP * 2;
    float* ga = gattn + pair0 * LP;
    for (int i = lane; i < npw * LP * 2; i += 64) gl[i] = sloc[i];
    for (int i = lane; i < npw * LP; i += 64) ga[i] = sattn[i];
  }
}

// The owner-computes formulations of grad_value accumulate in LDS in 64-bit FIXED POINT: ds_add_f32 runs at 0.33 lane-operations
// per clock per CU on gfx950 (measured, tools/hipbench/lds_atomic_rate.hip), ds_add_u64 at 9.2. With 2^e >= max|grad_out| *
// max|attn| every contribution is scaled by 2^(40-e), so |c| <= 2^40, 2^22 of them cannot overflow and the resolution is 2^-40
// of the largest possible contribution (fp32 atomics resolve 2^-24 of each partial sum). The result does not depend on the
// order of the additions. (Round 2's first form -- every tile re-scanning its level's samples -- and round 5's cell-sorted
// rows were measured slower than the binned path below at every size and left the product in round 6: git history,
// profiles/r05/msda_bwd_rows.md.)
constexpr int MSDA_FIXED_BITS = 40;

// 95 us here). `red`: >= 16 floats of LDS; every thread of the (<= 1024-thread) workgroup calls it.
__device__ __forceinline__ void msda_block_atomic_max(float m, unsigned* __restrict__ out, float* red) {
  m = mss_wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mm = fmaxf(mm, red[i]);
    const unsigned bits = __float_as_uint(mm);          // non-negative floats order like their bit patterns
    if (bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
  }
}

// ------------------------------------------------------------------------------------------
// grad_value, BINNED owner-computes path (round 3; replaces the re-scanning kernel above whenever the caller can hand over
// a HOST copy of spatial_shapes). The re-scanning kernel reads every sampling location of a level once per tile of that
// level (4 / 9 / 36 times at 704^2: 2.4 GB per launch at N = 16, 40 wave-instructions per hit). Here every in-image sample
// is filed ONCE, as a 16-byte record (query, h_im, w_im, attention weight), under the tile that holds its top-left
// corner ("home" tile; key = (image, head, tile)) by a counting sort:
//   msda_bin_count_kernel    per-key record counts (LDS histogram per 4096-sample chunk, one global add per key and chunk)
//   msda_bin_scan_kernel     exclusive scan -> first record of every key
//   msda_bin_scatter_kernel  the records, each chunk reserving its run of a key with one global add
//   msda_bwd_value_binned_kernel  a workgroup takes tiles off a ticket counter (heaviest levels first), accumulates the
//                            records of its tile in a (BH+1) x (BW+1) LDS window of 64-bit fixed-point words -- the extra
//                            row / column receives the corners that reach into the next tile -- stores the interior
//                            with plain stores and the halo row / column / corner into a side buffer
//   msda_bin_merge_kernel    adds each tile's three incoming halos to its first row / column in a fixed order
// No floating-point atomic, no memset of grad_value, bit-reproducible (integer sums do not depend on the order the
// records arrive in, the halo additions have a fixed order). Tile edge per level from the expected records per tile
// (8 x 8 where a level receives > 94 samples per position ... 16 x 16), so the 22^2 / 44^2 / 88^2 levels of a 704^2 crop
// give 9 + 9 + 36 tiles of 5.4 k / 5.4 k / 1.3 k records instead of 4 + 9 + 36 of 10 k / 4.5 k / 1.1 k.
constexpr int MSDA_BIN_MAXL = 8;
constexpr int MSDA_BIN_WIN = 289;          // window cells: (16 + 1) x (16 + 1)
constexpr int MSDA_BIN_TARGET = 6000;      // records per tile the tile-size rule aims at
constexpr int MSDA_BIN_CHUNK = 4096;       // samples per workgroup in the count / scatter kernels
constexpr int MSDA_BIN_NT = 1024;

struct MsdaBinLevel { int H, W, BH, BW, nr, nc, tile0, halo0; float invBH, invBW; };
struct MsdaBinGeom { MsdaBinLevel lv[MSDA_BIN_MAXL]; int L, ntiles, halo_cells; };

// host: tile geometry of every level. false: shapes this path does not take.
static bool msda_bin_geom(const int64_t* hs, int L, int Lq, int P, long long samples, MsdaBinGeom& g) {
  if (L < 1 || L > MSDA_BIN_MAXL) return false;
  // records per tile to aim at: enough tiles for ~6 workgroups per slot (two slots per CU) on small calls
  const long long target = samples / 3072 < 750 ? 750 : (samples / 3072 > MSDA_BIN_TARGET ? MSDA_BIN_TARGET : samples / 3072);
  g.L = L;
  int tile0 = 0, halo0 = 0;
  for (int l = 0; l < L; ++l) {
    const long long H = hs[2 * l], W = hs[2 * l + 1];
    if (H < 1 || W < 1 || H > 32767 || W > 32767) return false;
    const double density = (double)Lq * P / ((double)H * W);
    int cells = 16;                                        // 4x4, 4x8, 8x8, 8x16 or 16x16 positions
    while (cells < 256 && density * cells < target) cells *= 2;
    const int bh0 = cells == 256 ? 16 : (cells >= 64 ? 8 : 4);
    int BW = (int)(W < cells / bh0 ? W : cells / bh0);
    int BH = (int)(H < cells / BW ? H : cells / BW);
    BW = (int)(W < cells / BH ? W : cells / BH);
    if (BH >= BW) { const int cap = MSDA_BIN_WIN / (BW + 1) - 1; if (BH > cap) BH = cap; }
    else          { const int cap = MSDA_BIN_WIN / (BH + 1) - 1; if (BW > cap) BW = cap; }
    MsdaBinLevel& v = g.lv[l];
    v.H = (int)H; v.W = (int)W; v.BH = BH; v.BW = BW;
    v.nr = (int)((H + BH - 1) / BH); v.nc = (int)((W + BW - 1) / BW);
    v.tile0 = tile0; v.halo0 = halo0;
    v.invBH = 1.0f / (float)BH; v.invBW = 1.0f / (float)BW;
    const long long nt = (long long)v.nr * v.nc;
    if (tile0 + nt > (1 << 20)) return false;
    tile0 += (int)nt;
    halo0 += (int)nt * (BH + BW + 1);
  }
  for (int l = L; l < MSDA_BIN_MAXL; ++l) g.lv[l] = g.lv[L - 1];
  g.ntiles = tile0;
  g.halo_cells = halo0;
  return true;
}

// home tile of a sample inside level `v` (h_im / w_im already known to be inside (-1, H) x (-1, W))
__device__ __forceinline__ int msda_bin_home(const MsdaBinLevel& v, float h_im, float w_im) {
  const int hc = max((int)floorf(h_im), 0), wc = max((int)floorf(w_im), 0);
  const int br = (int)(((float)hc + 0.5f) * v.invBH), bc = (int)(((float)wc + 0.5f) * v.invBW);   // exact: hc, wc < 2^15
  return v.tile0 + br * v.nc + bc;
}

// A record is 4 dwords: header = kill << 31 | query << 9 | window cell of the top-left corner, then A = attn x (row weight of
// the corner's row), B = attn x (row weight of the row below) and the column fraction lw: the four contributions are
// (A, B) x (1 - lw, kill ? 0 : lw). Corners outside the image are folded in here -- a top-left corner in row / column -1
// moves to row / column 0 of the window and takes the other row's / column's weight, a row or column past the last one
// gets weight 0 -- so the accumulating kernel adds all four unconditionally.
// MODE 0: count. MODE 1: scatter (cursor[] holds the scan, advanced by every chunk's reservation).
template <int MODE>
__global__ __launch_bounds__(MSDA_BIN_NT) void msda_bin_kernel(MsdaBinGeom g, const float* __restrict__ loc,
                                                              const float* __restrict__ attn, int M, int Lq, int P,
                                                              int* __restrict__ counts_or_cursor, f32x4* __restrict__ records,
                                                              unsigned* __restrict__ absmax_attn, const float* __restrict__ gout,
                                                              long long ngout) {
  extern __shared__ int hist[];                              // [M][ntiles] + 16 floats for the block maximum
  __shared__ MsdaBinLevel lv[MSDA_BIN_MAXL];
  const int tid = threadIdx.x, n = blockIdx.y;
  if (tid < g.L) lv[tid] = g.lv[tid];
  const int nkeys = M * g.ntiles;
  for (int i = tid; i < nkeys; i += MSDA_BIN_NT) hist[i] = 0;
  __syncthreads();
  const int L = g.L;
  const long long per_image = (long long)Lq * M * L * P;
  const long long s0 = (long long)blockIdx.x * MSDA_BIN_CHUNK;
  constexpr int IT = MSDA_BIN_CHUNK / MSDA_BIN_NT;
  int key[IT], rank[IT], rq[IT];
  float rh[IT], rw[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const long long s = s0 + it * MSDA_BIN_NT + tid;
    key[it] = -1;
    if (s < per_image) {
      const long long gs = (long long)n * per_image + s;
      const float lx = loc[gs * 2], ly = loc[gs * 2 + 1];
      const unsigned su = (unsigned)s;                       // per_image < 2^31 (host-checked)
      const unsigned sp = su / (unsigned)P;
      const unsigned l = sp % (unsigned)L;
      const unsigned sm = sp / (unsigned)L;
      const unsigned m = sm % (unsigned)M;
      const MsdaBinLevel v = lv[l];
      const float w_im = __fmaf_rn(lx, (float)v.W, -0.5f), h_im = __fmaf_rn(ly, (float)v.H, -0.5f);   // one rounding in both modes
      if (h_im > -1.f && w_im > -1.f && h_im < (float)v.H && w_im < (float)v.W) {
        const int tile = msda_bin_home(v, h_im, w_im);
        key[it] = (int)m * g.ntiles + tile;
        rank[it] = atomicAdd(&hist[key[it]], 1);
        if (MODE == 1) { rq[it] = (int)((sm / (unsigned)M) << 4) | (int)l; rh[it] = h_im; rw[it] = w_im; }
      }
    }
  }
  __syncthreads();
  int* gk = counts_or_cursor + (size_t)n * nkeys;
  if (MODE == 0) {
    for (int i = tid; i < nkeys; i += MSDA_BIN_NT) {
      const int c = hist[i];
      if (c) atomicAdd(gk + i, c);
    }
    // ... and this workgroup's share of max|grad_out| (absmax_attn[-1]; 16-byte aligned, float4 per lane): saves a launch
    const long long n4 = ngout >> 2, nwg = (long long)gridDim.x * gridDim.y, wg = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    const long long per = (n4 + nwg - 1) / nwg, b4 = wg * per, e4 = min(n4, b4 + per);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(gout);
    float mx = 0.f;
    bool bad = false;
    for (long long i = b4 + tid; i < e4; i += MSDA_BIN_NT) {
      const f32x4 x = a4[i];
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
      bad |= !(x.x - x.x == 0.f) || !(x.y - x.y == 0.f) || !(x.z - x.z == 0.f) || !(x.w - x.w == 0.f);
    }
    if (wg == 0 && tid < (int)(ngout & 3)) {
      const float x = gout[(n4 << 2) + tid];
      mx = fmaxf(mx, fabsf(x));
      bad |= !(x - x == 0.f);
    }
    if (bad) mx = __builtin_huge_valf();
    msda_block_atomic_max(mx, absmax_attn - 1, reinterpret_cast<float*>(hist + nkeys));
    return;
  }
  for (int i = tid; i < nkeys; i += MSDA_BIN_NT) {
    const int c = hist[i];
    if (c) hist[i] = atomicAdd(gk + i, c);                   // first record of this chunk's run under key i
  }
  __syncthreads();
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    if (key[it] >= 0) {
      const long long gs = (long long)n * per_image + s0 + it * MSDA_BIN_NT + tid;
      const float aw = attn[gs];
      amax = (aw - aw == 0.f) ? fmaxf(amax, fabsf(aw)) : __builtin_huge_valf();     // non-finite poisons the gradient
      const MsdaBinLevel v = lv[rq[it] & 15];
      const int q = rq[it] >> 4;
      const float hf = floorf(rh[it]), wf = floorf(rw[it]);
      int h0 = (int)hf, w0 = (int)wf;
      const float lh = rh[it] - hf, lw = rw[it] - wf, hh = 1.f - lh, hw = 1.f - lw;
      float A, B, lwq;
      unsigned kill;
      if (h0 < 0) { A = aw * lh; B = 0.f; h0 = 0; }                       // row -1 is outside: row 0 moves up
      else { A = aw * hh; B = (h0 + 1 <= v.H - 1) ? aw * lh : 0.f; }
      if (w0 < 0) { lwq = hw; kill = 1u; w0 = 0; }                        // column 0 takes 1 - (1 - lw)
      else { lwq = lw; kill = (w0 + 1 <= v.W - 1) ? 0u : 1u; }
      const int tile = key[it] - (key[it] / g.ntiles) * g.ntiles - v.tile0;
      const int br = tile / v.nc, bc = tile - br * v.nc;
      const int r0 = br * v.BH, c0 = bc * v.BW;
      const int tw = min(v.W, c0 + v.BW) - c0;
      const int cell = (h0 - r0) * (tw + 1) + (w0 - c0);
      f32x4 r;
      r.x = __uint_as_float((kill << 31) | ((unsigned)q << 9) | (unsigned)cell);
      r.y = A; r.z = B; r.w = lwq;
      records[(size_t)hist[key[it]] + rank[it]] = r;
    }
  }
  msda_block_atomic_max(amax, absmax_attn, reinterpret_cast<float*>(hist + nkeys));   // NaN / inf were turned into +inf above
}

// exclusive scan of counts[nkeys] -> offsets[nkeys + 1] and cursor[nkeys] (one workgroup)
__global__ __launch_bounds__(1024) void msda_bin_scan_kernel(const int* __restrict__ counts, int nkeys, int* __restrict__ offsets,
                                                             int* __restrict__ cursor) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (nkeys + 1023) / 1024;
  const int b = tid * per, e = min(nkeys, b + per);
  int sum = 0;
  for (int i = b; i < e; ++i) sum += counts[i];
  part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int i = b; i < e; ++i) {
    offsets[i] = run;
    cursor[i] = run;
    run += counts[i];
  }
  if (tid == 1023) offsets[nkeys] = part[1023];
}

template <int UN>
__global__ __launch_bounds__(MSDA_BIN_NT, 8) void msda_bwd_value_binned_kernel(        // 8 waves / SIMD: two workgroups per CU

    MsdaBinGeom g, const int64_t* __restrict__ starts, const float* __restrict__ gout, const unsigned* __restrict__ absmax,
    const f32x4* __restrict__ records, const int* __restrict__ offsets, int* __restrict__ ticket, int S, int M, int Lq, int N,
    float* __restrict__ gvalue, float* __restrict__ halo) {
  constexpr int D = 32, NT = MSDA_BIN_NT;
  extern __shared__ unsigned long long win[];              // [(th + 1) * (tw + 1)][32]
  __shared__ int s_item;
  __shared__ MsdaBinLevel slv[MSDA_BIN_MAXL];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, d = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < g.L) slv[tid] = g.lv[tid];
  const int NM = N * M, nitems = g.ntiles * NM;
  const float bound = __uint_as_float(absmax[0]) * __uint_as_float(absmax[1]);
  int e = 0;
  if (bound > 0.f && bound < __builtin_huge_valf()) (void)frexpf(bound, &e);
  const bool finite = bound < __builtin_huge_valf();
  const double to_fixed = ldexp(1.0, MSDA_FIXED_BITS - e);
  const double from_fixed = finite ? ldexp(1.0, e - MSDA_FIXED_BITS) : (double)__builtin_nanf("");
  const size_t rs = (size_t)M * D;
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(gout), 0, (int)(unsigned)min((unsigned long long)N * Lq * rs * 4ull, 0xffffffffull), 0x00020000);
  for (;;) {
    if (tid == 0) s_item = atomicAdd(ticket, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= nitems) break;
    const int t = item / NM, nm = item - t * NM;            // tile-major: the heaviest (coarsest) levels go first
    const int n = nm / M, m = nm - n * M;
    int l = 0;
    while (l + 1 < g.L && t >= slv[l + 1].tile0) ++l;
    const MsdaBinLevel v = slv[l];
    const int tt = t - v.tile0, br = tt / v.nc, bc = tt - br * v.nc;
    const int r0 = br * v.BH, r1 = min(v.H, r0 + v.BH), c0 = bc * v.BW, c1 = min(v.W, c0 + v.BW);
    const int th = r1 - r0, tw = c1 - c0, ww = tw + 1;
    const int wcells = (th + 1) * ww;
    for (int i = tid; i < wcells * D; i += NT) win[i] = 0ull;
    __syncthreads();
    const int key = nm * g.ntiles + t;
    const int beg = __builtin_amdgcn_readfirstlane(offsets[key]), end = __builtin_amdgcn_readfirstlane(offsets[key + 1]);
    auto fx = [&](float c) {
      const double tq = fma((double)c, to_fixed, 6755399441055744.0);
      return (unsigned long long)(__double_as_longlong(tq) - 0x4338000000000000ll);
    };
    // a wave takes UN records at a time, ONE record per wave-instruction: the record words come through the scalar cache
    // (wave-uniform address) and are used as scalar operands; lane = column * 32 + channel, so lanes 0-31 add the two
    // corners of the left column, lanes 32-63 those of the right column, and a wave's 64 LDS words are consecutive
    unsigned long long* wl = win + lane;
    unsigned long long* wl2 = wl + ww * D;
    // grad_out as ONE buffer resource (host-checked: < 4 GB): a row is a scalar byte offset (SGPR), the channel a constant
    // per-lane offset -- no 64-bit address arithmetic and no address registers per load in flight
    const unsigned go_nm = (unsigned)(((size_t)n * Lq * M + m) * D * sizeof(float));
    const unsigned rs4 = (unsigned)(rs * sizeof(float));
    const unsigned d4 = (unsigned)d * 4u;
    // A wave owns blocks of 64 consecutive records: lane k fetches record k of the block (one coalesced 1-KB load, the next
    // block's requested a block ahead), then the block is walked UN records at a time, a record's four words broadcast
    // with v_readlane (no memory round trip, no LDS) and its grad_out row requested one sub-round ahead. (Fetching the
    // records through the scalar cache instead serialised on SGPR pressure: 4 us per round of 8 records.)
    const f32x4 zero_rec = {0.f, 0.f, 0.f, 0.f};               // A = B = 0: adds zeros to cell 0 of query 0
    auto load_g = [&](const f32x4& rv, int k0, float* gq) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const unsigned hd = (unsigned)__builtin_amdgcn_readlane(__float_as_int(rv.x), k0 + u);
        gq[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, d4, go_nm + ((hd >> 9) & 0x3fffffu) * rs4, 0));
      }
    };
    int blk = beg + wave * 64;
    f32x4 rv = blk + lane < end ? records[blk + lane] : zero_rec;
    for (; blk < end; blk += (NT / 64) * 64) {
      const int nb = blk + (NT / 64) * 64;
      const f32x4 rvn = nb + lane < end ? records[nb + lane] : zero_rec;
      const int cnt = min(64, end - blk);
      auto process = [&](int k0, const float* gq) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const unsigned hd = (unsigned)__builtin_amdgcn_readlane(__float_as_int(rv.x), k0 + u);
          const float A = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv.y), k0 + u));
          const float B = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv.z), k0 + u));
          const float lwv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv.w), k0 + u));
          const float lwk = (hd >> 31) ? 0.f : lwv;            // scalar select
          const float cw = half ? lwk : 1.f - lwv;
          const float tc = cw * gq[u];
          const int cell = (int)(hd & 511u) * D;
          atomicAdd(wl + cell, fx(tc * A));
          atomicAdd(wl2 + cell, fx(tc * B));
        }
      };
      // two register sets in ping-pong (no copies: a copy would wait for the rows just requested); lanes past the end of
      // the tile hold zero records, so a sub-round past `cnt` only adds zeros
      float ga[UN], gb[UN];
      load_g(rv, 0, ga);
      for (int k0 = 0; k0 < cnt; k0 += 2 * UN) {
        load_g(rv, k0 + UN, gb);
        process(k0, ga);
        if (k0 + 2 * UN < 64) load_g(rv, k0 + 2 * UN, ga);
        process(k0 + UN, gb);
      }
      rv = rvn;
    }
    __syncthreads();
    float* gv = gvalue + ((size_t)n * S + (size_t)starts[l]) * rs + (size_t)m * D;
    for (int i = tid; i < th * tw * D; i += NT) {
      const int cell = i >> 5, ch = i & 31;
      const int r = cell / tw, c = cell - r * tw;
      gv[(size_t)((r0 + r) * v.W + c0 + c) * rs + ch] = (float)((double)(long long)win[(r * ww + c) * D + ch] * from_fixed);
    }
    // halo: slots [0, tw] = window row th (the corner last), slots [BW + 1, BW + 1 + th) = window column tw
    float* hb = halo + ((size_t)nm * g.halo_cells + v.halo0 + (size_t)tt * (v.BH + v.BW + 1)) * D;
    if (r1 < v.H) {
      const int ncol = c1 < v.W ? tw + 1 : tw;
      for (int i = tid; i < ncol * D; i += NT)
        hb[i] = (float)((double)(long long)win[(th * ww) * D + i] * from_fixed);
    }
    if (c1 < v.W) {
      for (int i = tid; i < th * D; i += NT) {
        const int r = i >> 5, ch = i & 31;
        hb[(size_t)(v.BW + 1 + r) * D + ch] = (float)((double)(long long)win[(r * ww + tw) * D + ch] * from_fixed);
      }
    }
  }
}

// grad_value[first row / first column of every tile] += the halos its upper / left / upper-left neighbours left behind
__global__ __launch_bounds__(256) void msda_bin_merge_kernel(MsdaBinGeom g, const int64_t* __restrict__ starts, int S, int M,
                                                            int N, const float* __restrict__ halo, float* __restrict__ gvalue) {
  constexpr int D = 32;
  const int item = blockIdx.x, nm = blockIdx.y;
  int l = 0;
  while (l + 1 < g.L && item >= g.lv[l + 1].tile0) ++l;
  const MsdaBinLevel v = g.lv[l];
  const int tt = item - v.tile0, br = tt / v.nc, bc = tt - br * v.nc;
  if (br == 0 && bc == 0) return;
  const int n = nm / M, m = nm - n * M;
  const int r0 = br * v.BH, r1 = min(v.H, r0 + v.BH), c0 = bc * v.BW, c1 = min(v.W, c0 + v.BW);
  const int th = r1 - r0, tw = c1 - c0;
  const int slots = v.BH + v.BW + 1;
  const size_t rs = (size_t)M * D;
  const float* hl = halo + ((size_t)nm * g.halo_cells + v.halo0) * D;
  float* gv = gvalue + ((size_t)n * S + (size_t)starts[l]) * rs + (size_t)m * D;
  const int ch = threadIdx.x & 31;
  // cells of the first row (j = 0 .. tw-1), then of the first column below it (j = tw .. tw+th-2)
  for (int j = threadIdx.x >> 5; j < tw + th - 1; j += 8) {
    const int r = j < tw ? 0 : j - tw + 1, c = j < tw ? j : 0;
    float add = 0.f;
    if (r == 0 && br > 0) add += hl[((size_t)(tt - v.nc) * slots + c) * D + ch];                      // upper tile's row
    if (c == 0 && bc > 0) add += hl[((size_t)(tt - 1) * slots + v.BW + 1 + r) * D + ch];             // left tile's column
    if (r == 0 && c == 0 && br > 0 && bc > 0) {
      // upper-left tile's corner: slot (its width) = BW, it is never a ragged tile
      add += hl[((size_t)(tt - v.nc - 1) * slots + v.BW) * D + ch];
    }
    if ((r == 0 && br > 0) || (c == 0 && bc > 0)) gv[(size_t)((r0 + r) * v.W + c0 + c) * rs + ch] += add;
  }
}

// ------------------------------------------------------------------------------------------
// Operand preparation of the MSDeformAttn module (ops/modules/ms_deform_attn.py:100-109) in one pass:
//   attn = softmax over the L*P logits of a (query, head);  loc = reference_point[l] + offset / (W_l, H_l)
// instead of softmax + view + stack + div + add as five elementwise library kernels over the 12-36 values per (q, m).
// One thread per (n, q, m). Backward: d_logit = attn * (g_attn - sum(attn * g_attn)), d_offset = g_loc / (W_l, H_l).
// A workgroup stages the contiguous logits / offsets of its 256 pairs through LDS (coalesced loads and stores; a
// thread then owns one pair's row: stride L*P and 2*L*P floats, odd multiples of 4 banks for the usual 12 / 24).
__global__ __launch_bounds__(256) void msda_prepare_kernel(const float* __restrict__ offsets, const float* __restrict__ logits,
                                                           const float* __restrict__ ref, const int64_t* __restrict__ shapes,
                                                           long long npairs, int M, int L, int P, float* __restrict__ loc,
                                                           float* __restrict__ attn, long long ldo, long long ldl) {
  extern __shared__ float sm[];
  const int LP = L * P;
  float* sl = sm;                 // [256][LP]      logits -> attention weights
  float* so = sm + 256 * LP;      // [256][2 LP]    offsets -> locations
  const long long pair0 = (long long)blockIdx.x * 256;
  const int np = (int)min((long long)256, npairs - pair0);
  if (ldo == (long long)M * LP * 2 && ldl == (long long)M * LP) {
    for (int i = threadIdx.x; i < np * LP; i += 256) sl[i] = logits[pair0 * LP + i];
    for (int i = threadIdx.x; i < np * LP * 2; i += 256) so[i] = offsets[pair0 * LP * 2 + i];
  } else {          // inputs are column ranges of a wider buffer (ldo / ldl floats per (n, q) row); outputs stay dense
    for (int i = threadIdx.x; i < np * LP; i += 256) {
      const long long pr = pair0 + i / LP;
      sl[i] = logits[pr / M * ldl + (pr % M) * LP + i % LP];
    }
    for (int i = threadIdx.x; i < np * LP * 2; i += 256) {
      const long long pr = pair0 + i / (LP * 2);
      so[i] = offsets[pr / M * ldo + (pr % M) * (LP * 2) + i % (LP * 2)];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < np) {
    const long long nq = (pair0 + threadIdx.x) / M;
    float* lg = sl + threadIdx.x * LP;
    float* off = so + threadIdx.x * LP * 2;
    float mx = -__builtin_huge_valf();
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
    for (int i = 0; i < LP; ++i) { const float e = expf(lg[i] - mx); lg[i] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int l = 0; l < L; ++l) {
      const float fw = (float)shapes[2 * l + 1], fh = (float)shapes[2 * l];
      const float rx = ref[(nq * L + l) * 2], ry = ref[(nq * L + l) * 2 + 1];
      for (int pt = 0; pt < P; ++pt) {
        const int i = l * P + pt;
        lg[i] *= inv;
        off[i * 2] = rx + off[i * 2] / fw;
        off[i * 2 + 1] = ry + off[i * 2 + 1] / fh;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < np * LP; i += 256) attn[pair0 * LP + i] = sl[i];
  for (int i = threadIdx.x; i < np * LP * 2; i += 256) loc[pair0 * LP * 2 + i] = so[i];
}

// ldo / ldl: row strides (floats) of goffsets / glogits per (n, q) -- M*2LP / M*LP when dense; larger when both live in ONE
// [N*Lq, M*3LP] buffer (offsets | logits), so that the two projections' weight and data gradients are one GEMM each (r04)
__global__ __launch_bounds__(256) void msda_prepare_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ gattn,
                                                               const float* __restrict__ gloc, const int64_t* __restrict__ shapes,
                                                               long long npairs, int M, int L, int P, float* __restrict__ goffsets,
                                                               long long ldo, float* __restrict__ glogits, long long ldl) {
  extern __shared__ float sm[];
  const int LP = L * P;
  float* sa = sm;                 // [256][LP]   attn
  float* sg = sm + 256 * LP;      // [256][LP]   g_attn -> d_logits
  const long long pair0 = (long long)blockIdx.x * 256;
  const int np = (int)min((long long)256, npairs - pair0);
  for (int i = threadIdx.x; i < np * LP; i += 256) { sa[i] = attn[pair0 * LP + i]; sg[i] = gattn[pair0 * LP + i]; }
  // d_offsets is elementwise: coalesced straight through
  for (int i = threadIdx.x; i < np * LP * 2; i += 256) {
    const int l = (i / 2 % LP) / P;
    const long long pair = pair0 + i / (2 * LP);
    goffsets[pair / M * ldo + (pair % M) * (2 * LP) + i % (2 * LP)] = gloc[pair0 * LP * 2 + i] / (float)shapes[2 * l + ((i & 1) ? 0 : 1)];
  }
  __syncthreads();
  if ((int)threadIdx.x < np) {
    const float* a = sa + threadIdx.x * LP;
    float* ga = sg + threadIdx.x * LP;
    float dot = 0.f;
    for (int i = 0; i < LP; ++i) dot = __builtin_fmaf(a[i], ga[i], dot);     // (explicit: the gather kernel's PROJ epilogue must round alike)
    for (int i = 0; i < LP; ++i) ga[i] = a[i] * (ga[i] - dot);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < np * LP; i += 256) {
    const long long pair = pair0 + i / LP;
    glogits[pair / M * ldl + (pair % M) * LP + i % LP] = sg[i];
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
int msda_forward_rec(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc,
                      const float* attn, const float* ref, int N, int S, int M, int L, int Lq, int P, float* out,
                      hipStream_t stream, long long ldo = 0, long long ldl = 0, float* loc_out = nullptr, float* attn_out = nullptr) {
  constexpr int HPW = 64 / LPH;
  const long long npairs = (long long)N * Lq * M;
  const long long nblocks = (npairs + 4 * HPW - 1) / (4 * HPW);
  if (ldo <= 0) ldo = (long long)M * L * P * 2;
  if (ldl <= 0) ldl = (long long)M * L * P;
  // buffer-resource addressing (32-bit offsets, hardware zero fill) and the 0xffffff00 out-of-range marker need a value tensor
  // below that size; beyond it (and where the records do not fit in LDS) the caller falls back to the generic kernel
  const size_t smem_rec = (size_t)4 * HPW * (L * P + 1) * 9 * sizeof(float);
  if ((unsigned long long)N * S * M * (4 * LPH) * 4ull >= 0xffffff00ull || smem_rec > 65536) return MSS_ERR_UNSUPPORTED;
  if (ref)
    hipLaunchKernelGGL((msda_fwd_rec_kernel<LPH, true>), dim3((unsigned)nblocks), dim3(256), smem_rec, stream, value, shapes, starts, loc,
                       attn, ref, npairs, S, M, L, Lq, P, out, ldo, ldl, L * P + 1, loc_out, attn_out);
  else
    hipLaunchKernelGGL((msda_fwd_rec_kernel<LPH, false>), dim3((unsigned)nblocks), dim3(256), smem_rec, stream, value, shapes, starts, loc,
                       attn, ref, npairs, S, M, L, Lq, P, out, ldo, ldl, L * P + 1);
  return mss_launch_status();
}

template <typename T>
int msda_backward(const T* value, const int64_t* shapes, const int64_t* starts, const T* loc, const T* attn,
                  const T* gout, int N, int S, int M, int D, int L, int Lq, int P, T* gvalue, T* gloc, T* gattn,
                  hipStream_t stream) {
  int rc = msda_check(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  if (!gvalue && N > 0) return MSS_ERR_BAD_ARG;
  const long long npairs = (long long)N * Lq * M;
  // the generic formulation (any dtype / head dimension, no host copy of the level shapes needed): scatter-add with whole-row atomics
  if (N > 0) {
    hipError_t e = hipMemsetAsync(gvalue, 0, (size_t)N * S * M * D * sizeof(T), stream);
    if (e != hipSuccess) return (int)e;
  }
  if (npairs == 0) return MSS_OK;
  if (!gout || !gloc || !gattn) return MSS_ERR_BAD_ARG;
  if (D <= 32) {
    const long long nblocks = (npairs + 7) / 8;
    hipLaunchKernelGGL((msda_bwd_kernel<T, 32, true>), dim3((unsigned)nblocks), dim3(256), 0, stream, value, shapes,
                       starts, loc, attn, gout, npairs, S, M, D, L, Lq, P, gvalue, gloc, gattn);
  } else {
    const long long nblocks = (npairs + 3) / 4;
    hipLaunchKernelGGL((msda_bwd_kernel<T, 64, true>), dim3((unsigned)nblocks), dim3(256), 0, stream, value, shapes,
                       starts, loc, attn, gout, npairs, S, M, D, L, Lq, P, gvalue, gloc, gattn);
  }
  return mss_launch_status();
}

// grad_loc / grad_attn (and, with goff / glog, the module's backward): the gather pass (no atomics)
static int msda_launch_gather(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc, const float* attn,
                              const float* gout, int N, int S, int M, int L, int Lq, int P, float* gloc, float* gattn, float* goff,
                              long long ldo, float* glog, long long ldl, hipStream_t stream) {
  const long long npairs = (long long)N * Lq * M;
  const size_t smem_gather = (size_t)4 * 8 * L * P * 4 * sizeof(float);
  const long long nblocks = (npairs + 31) / 32;
  const bool buf = (unsigned long long)N * S * M * 32 * 4ull < 0xffffffffull;
  if (buf)
    hipLaunchKernelGGL(msda_bwd_gather_fast_kernel<true>, dim3((unsigned)nblocks), dim3(256), smem_gather, stream, value, shapes, starts,
                       loc, attn, gout, npairs, S, M, L, Lq, P, gloc, gattn, goff, ldo, glog, ldl);
  else
    hipLaunchKernelGGL(msda_bwd_gather_fast_kernel<false>, dim3((unsigned)nblocks), dim3(256), smem_gather, stream, value, shapes, starts,
                       loc, attn, gout, npairs, S, M, L, Lq, P, gloc, gattn, goff, ldo, glog, ldl);
  return mss_launch_status();
}

struct MsdaBinWs { size_t counts, offsets, cursor, records, halo, total; long long nkeys; };
static bool msda_bin_layout(const MsdaBinGeom& g, int N, int M, int L, int Lq, int P, MsdaBinWs& w) {
  const long long per_image = (long long)Lq * M * L * P;
  if (per_image >= (1ll << 31) || (long long)N * per_image >= (1ll << 31)) return false;
  w.nkeys = (long long)N * M * g.ntiles;
  if (w.nkeys >= (1ll << 28) || (long long)M * g.ntiles * 4 > 60 * 1024) return false;      // LDS histogram of a chunk
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  w.counts = 64;
  w.offsets = up(w.counts + (size_t)w.nkeys * 4);
  w.cursor = up(w.offsets + (size_t)(w.nkeys + 1) * 4);
  w.records = up(w.cursor + (size_t)w.nkeys * 4);
  w.halo = up(w.records + (size_t)N * per_image * 16);
  w.total = up(w.halo + (size_t)N * M * g.halo_cells * 32 * 4);
  return true;
}

static int msda_backward_binned(const float* value, const int64_t* shapes, const int64_t* starts, const int64_t* host_shapes,
                                const float* loc, const float* attn, const float* gout, int N, int S, int M, int D, int L,
                                int Lq, int P, float* gvalue, float* gloc, float* gattn, void* ws, size_t ws_bytes,
                                hipStream_t stream, float* goff = nullptr, long long ldo = 0, float* glog = nullptr, long long ldl = 0) {
  if (!host_shapes) return MSS_ERR_BAD_ARG;
  if (goff || glog) {          // the module's backward folded into the gather pass: both outputs, strides at least dense, L*P <= 20
    if (!goff || !glog || ldo < (long long)M * 2 * L * P || ldl < (long long)M * L * P) return MSS_ERR_BAD_ARG;
    if (L * P > 20) return MSS_ERR_UNSUPPORTED;
  }
  int rc = msda_check(value, host_shapes, starts, loc, attn, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  const long long npairs = (long long)N * Lq * M;
  if (npairs == 0 || (long long)N * S == 0) return MSS_ERR_UNSUPPORTED;
  if (D != 32 || M > 65535 || (long long)Lq * P > (1ll << 22)) return MSS_ERR_UNSUPPORTED;
  if ((unsigned long long)N * Lq * M * D * 4ull >= 0xffffffffull) return MSS_ERR_UNSUPPORTED;      // grad_out as one buffer resource
  if (!gvalue || !gout || !ws || (!goff && (!gloc || !gattn))) return MSS_ERR_BAD_ARG;
  const size_t smem_gather = (size_t)4 * 8 * L * P * 4 * sizeof(float);
  if (((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(gout)) & 15) != 0 || smem_gather > 65536) return MSS_ERR_UNSUPPORTED;
  long long cells = 0;
  for (int l = 0; l < L; ++l) cells += host_shapes[2 * l] * host_shapes[2 * l + 1];
  if (cells > S) return MSS_ERR_BAD_ARG;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return MSS_ERR_BAD_ARG;
  MsdaBinGeom g;
  MsdaBinWs w;
  if (!msda_bin_geom(host_shapes, L, Lq, P, (long long)N * Lq * M * L * P, g) || !msda_bin_layout(g, N, M, L, Lq, P, w)) return MSS_ERR_UNSUPPORTED;
  if (ws_bytes < w.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return MSS_ERR_BAD_ARG;
  if (cells < S) {
    // the tiles plain-store exactly the rows of the L levels; a value tensor with more rows than the levels cover
    // (padding behind the last level, or gaps between levels in level_start_index) must still come back zero there,
    // as the reference's zero-initialised output does (ms_deform_attn_cuda.cu:126)
    hipError_t ez = hipMemsetAsync(gvalue, 0, (size_t)N * S * M * D * sizeof(float), stream);
    if (ez != hipSuccess) return (int)ez;
  }
  char* base = static_cast<char*>(ws);
  int* ticket = reinterpret_cast<int*>(base);
  unsigned* absmax = reinterpret_cast<unsigned*>(base) + 1;
  int* counts = reinterpret_cast<int*>(base + w.counts);
  int* offsets = reinterpret_cast<int*>(base + w.offsets);
  int* cursor = reinterpret_cast<int*>(base + w.cursor);
  f32x4* records = reinterpret_cast<f32x4*>(base + w.records);
  float* halo = reinterpret_cast<float*>(base + w.halo);
  hipError_t e = hipMemsetAsync(base, 0, w.counts + (size_t)w.nkeys * 4, stream);
  if (e != hipSuccess) return (int)e;
  const long long per_image = (long long)Lq * M * L * P;
  const unsigned chunks = (unsigned)((per_image + MSDA_BIN_CHUNK - 1) / MSDA_BIN_CHUNK);
  const size_t hist_bytes = (size_t)M * g.ntiles * sizeof(int) + 16 * sizeof(float);
  hipLaunchKernelGGL(msda_bin_kernel<0>, dim3(chunks, (unsigned)N), dim3(MSDA_BIN_NT), hist_bytes, stream, g, loc, attn, M, Lq, P,
                     counts, records, absmax + 1, gout, npairs * D);          // + max|grad_out|
  hipLaunchKernelGGL(msda_bin_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, (int)w.nkeys, offsets, cursor);
  hipLaunchKernelGGL(msda_bin_kernel<1>, dim3(chunks, (unsigned)N), dim3(MSDA_BIN_NT), hist_bytes, stream, g, loc, attn, M, Lq, P,
                     cursor, records, absmax + 1, gout, 0ll);       // + max|attn| over the filed samples
  const size_t win_bytes = (size_t)MSDA_BIN_WIN * 32 * sizeof(unsigned long long);
  auto kern = msda_bwd_value_binned_kernel<8>;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes);
  if (e != hipSuccess) return (int)e;
  const long long nitems = (long long)g.ntiles * N * M;
  const unsigned nwg = (unsigned)(nitems < 512 ? nitems : 512);          // two 1024-thread workgroups per CU, 256 CUs
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(MSDA_BIN_NT), win_bytes, stream, g, starts, gout, absmax, records, offsets, ticket, S, M,
                     Lq, N, gvalue, halo);
  hipLaunchKernelGGL(msda_bin_merge_kernel, dim3((unsigned)g.ntiles, (unsigned)(N * M)), dim3(256), 0, stream, g, starts, S, M, N,
                     halo, gvalue);
  return msda_launch_gather(value, shapes, starts, loc, attn, gout, N, S, M, L, Lq, P, gloc, gattn, goff, ldo, glog, ldl, stream);
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
  rc = MSS_ERR_UNSUPPORTED;             // the record kernel: fp32, head dimension 16 / 32 / 64, value below 4 GB; else the generic one
  if (aligned && D == 32) rc = msda_forward_rec<8>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, N, S, M, L, Lq, P, out, s);
  if (aligned && D == 16) rc = msda_forward_rec<4>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, N, S, M, L, Lq, P, out, s);
  if (aligned && D == 64) rc = msda_forward_rec<16>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, N, S, M, L, Lq, P, out, s);
  if (rc != MSS_ERR_UNSUPPORTED) return rc;
  return msda_forward<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L,
                             Lq, P, out, s);
}

// forward straight from the module's raw projections (softmax + location arithmetic inside the sampling kernel); fp32,
// head dimension 16 / 32 / 64 only -- MSS_ERR_UNSUPPORTED otherwise (the caller then runs prepare + forward).
// ld_offsets / ld_logits: floats between the rows of consecutive (n, q) of `offsets` [N,Lq,M,L,P,2] / `logits` [N,Lq,M,L,P];
// 0 = dense. With both > dense the two tensors may be column ranges of ONE [N*Lq, M*3*L*P] buffer, i.e. the output of a single
// product q [Woff ; Watt]^T (ops/modules/ms_deform_attn.py:98-101 are two Linears on the same query).
// mss_msda_forward_fused_save_f32: the same, and the sampling locations [N,Lq,M,L,P,2] / attention weights [N,Lq,M,L,P] the kernel
// formed on the way are written out (training: the backward reads exactly what the forward used instead of re-deriving them with
// mss_msda_prepare_f32). MSS_ERR_UNSUPPORTED where the record kernel does not run (the caller then saves nothing and prepares).
int mss_msda_forward_fused_save_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* offsets, long long ld_offsets, const float* logits, long long ld_logits,
                                    const float* reference_points, int N, int S, int M, int D, int L, int Lq, int P, float* out,
                                    float* sampling_loc_out, float* attn_weight_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((sampling_loc_out == nullptr) != (attn_weight_out == nullptr)) return MSS_ERR_BAD_ARG;
  int rc = msda_check(value, spatial_shapes, level_start_index, offsets, logits, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  if ((long long)N * Lq * M == 0) return MSS_OK;
  if (!out || !reference_points) return MSS_ERR_BAD_ARG;
  if ((ld_offsets && ld_offsets < (long long)M * 2 * L * P) || (ld_logits && ld_logits < (long long)M * L * P)) return MSS_ERR_BAD_ARG;
  if (L * P > 20) return MSS_ERR_UNSUPPORTED;           // per-lane sample slots of the in-LDS softmax (as mss_msda_prepare_f32)
  const bool aligned = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (!aligned) return MSS_ERR_UNSUPPORTED;
  if (D == 32) return msda_forward_rec<8>(value, spatial_shapes, level_start_index, offsets, logits, reference_points, N, S, M, L, Lq, P, out, s,
                                          ld_offsets, ld_logits, sampling_loc_out, attn_weight_out);
  if (D == 16) return msda_forward_rec<4>(value, spatial_shapes, level_start_index, offsets, logits, reference_points, N, S, M, L, Lq, P, out, s,
                                          ld_offsets, ld_logits, sampling_loc_out, attn_weight_out);
  if (D == 64) return msda_forward_rec<16>(value, spatial_shapes, level_start_index, offsets, logits, reference_points, N, S, M, L, Lq, P, out, s,
                                           ld_offsets, ld_logits, sampling_loc_out, attn_weight_out);
  return MSS_ERR_UNSUPPORTED;
}

int mss_msda_forward_fused_ld_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                  const float* offsets, long long ld_offsets, const float* logits, long long ld_logits,
                                  const float* reference_points, int N, int S, int M, int D, int L, int Lq, int P, float* out,
                                  void* stream) {
  return mss_msda_forward_fused_save_f32(value, spatial_shapes, level_start_index, offsets, ld_offsets, logits, ld_logits,
                                         reference_points, N, S, M, D, L, Lq, P, out, nullptr, nullptr, stream);
}

int mss_msda_forward_fused_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* offsets, const float* logits, const float* reference_points, int N, int S,
                               int M, int D, int L, int Lq, int P, float* out, void* stream) {
  return mss_msda_forward_fused_ld_f32(value, spatial_shapes, level_start_index, offsets, 0, logits, 0, reference_points, N, S, M, D,
                                       L, Lq, P, out, stream);
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

// backward with grad_value on the binned owner-computes path (fp32, D = 32, L <= 8): `host_shapes` is a HOST copy of
// spatial_shapes, `workspace` a 256-byte aligned device buffer of at least mss_msda_backward_workspace_bytes(...) bytes
// (0: shapes this path does not take -> call mss_msda_backward_f32). MSS_ERR_UNSUPPORTED likewise.
long long mss_msda_backward_workspace_bytes(const int64_t* host_shapes, int N, int M, int D, int L, int Lq, int P) {
  MsdaBinGeom g;
  MsdaBinWs w;
  if (!host_shapes || D != 32 || N <= 0 || Lq <= 0 || M <= 0 || P <= 0 || M > 65535 || (long long)Lq * P > (1ll << 22)) return 0;
  if ((unsigned long long)N * Lq * M * D * 4ull >= 0xffffffffull) return 0;
  if ((size_t)4 * 8 * L * P * 4 * sizeof(float) > 65536) return 0;
  if (!msda_bin_geom(host_shapes, L, Lq, P, (long long)N * Lq * M * L * P, g) || !msda_bin_layout(g, N, M, L, Lq, P, w)) return 0;
  return (long long)w.total;
}

int mss_msda_backward_binned_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const int64_t* host_shapes, const float* sampling_loc, const float* attn_weight,
                                 const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_value,
                                 float* grad_loc, float* grad_attn, void* workspace, long long workspace_bytes, void* stream) {
  if (!spatial_shapes) return MSS_ERR_BAD_ARG;
  return msda_backward_binned(value, spatial_shapes, level_start_index, host_shapes, sampling_loc, attn_weight, grad_out, N, S, M, D,
                              L, Lq, P, grad_value, grad_loc, grad_attn, workspace, (size_t)(workspace_bytes < 0 ? 0 : workspace_bytes),
                              static_cast<hipStream_t>(stream));
}

// The same with the MODULE's backward folded in (ops/modules/ms_deform_attn.py:100-109: softmax over the L*P logits, loc = ref +
// offset / (W_l, H_l)): instead of grad_sampling_loc / grad_attn_weight it writes d(offsets) and d(logits), row (n, q) at
// + (n*Lq + q) * ld (both may be column ranges of one [N*Lq, M*3*L*P] buffer, the output gradient of the merged projection).
// Equal, bit for bit, to mss_msda_backward_binned_f32 followed by mss_msda_prepare_backward_ld_f32; L*P <= 20.
int mss_msda_backward_binned_proj_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                      const int64_t* host_shapes, const float* sampling_loc, const float* attn_weight,
                                      const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_value,
                                      float* grad_offsets, long long ld_offsets, float* grad_logits, long long ld_logits,
                                      void* workspace, long long workspace_bytes, void* stream) {
  if (!spatial_shapes || !grad_offsets || !grad_logits) return MSS_ERR_BAD_ARG;
  return msda_backward_binned(value, spatial_shapes, level_start_index, host_shapes, sampling_loc, attn_weight, grad_out, N, S, M, D,
                              L, Lq, P, grad_value, nullptr, nullptr, workspace, (size_t)(workspace_bytes < 0 ? 0 : workspace_bytes),
                              static_cast<hipStream_t>(stream), grad_offsets, ld_offsets, grad_logits, ld_logits);
}

int mss_msda_prepare_ld_f32(const float* offsets, long long ld_offsets, const float* logits, long long ld_logits,
                            const float* reference_points, const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P,
                            float* sampling_loc, float* attn_weight, void* stream) {
  if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0 || L * P > MSDA_MAX_LP) return MSS_ERR_UNSUPPORTED;
  const long long npairs = (long long)N * Lq * M;
  if (npairs == 0) return MSS_OK;
  if (!offsets || !logits || !reference_points || !spatial_shapes || !sampling_loc || !attn_weight) return MSS_ERR_BAD_ARG;
  if (ld_offsets <= 0) ld_offsets = (long long)M * 2 * L * P;
  if (ld_logits <= 0) ld_logits = (long long)M * L * P;
  if (ld_offsets < (long long)M * 2 * L * P || ld_logits < (long long)M * L * P) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(msda_prepare_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), (size_t)256 * 3 * L * P * sizeof(float),
                     static_cast<hipStream_t>(stream),
                     offsets, logits, reference_points, spatial_shapes, npairs, M, L, P, sampling_loc, attn_weight, ld_offsets,
                     ld_logits);
  return mss_launch_status();
}

int mss_msda_prepare_f32(const float* offsets, const float* logits, const float* reference_points,
                         const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P, float* sampling_loc,
                         float* attn_weight, void* stream) {
  return mss_msda_prepare_ld_f32(offsets, 0, logits, 0, reference_points, spatial_shapes, N, Lq, M, L, P, sampling_loc, attn_weight,
                                 stream);
}

int mss_msda_prepare_backward_ld_f32(const float* attn_weight, const float* grad_attn, const float* grad_loc,
                                     const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P, float* grad_offsets,
                                     long long ld_offsets, float* grad_logits, long long ld_logits, void* stream) {
  if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0 || L * P > MSDA_MAX_LP) return MSS_ERR_UNSUPPORTED;
  const long long npairs = (long long)N * Lq * M;
  if (npairs == 0) return MSS_OK;
  if (!attn_weight || !grad_attn || !grad_loc || !spatial_shapes || !grad_offsets || !grad_logits) return MSS_ERR_BAD_ARG;
  if (ld_offsets < (long long)M * 2 * L * P || ld_logits < (long long)M * L * P) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(msda_prepare_bwd_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256),
                     (size_t)256 * 2 * L * P * sizeof(float), static_cast<hipStream_t>(stream), attn_weight, grad_attn, grad_loc, spatial_shapes, npairs, M, L, P,
                     grad_offsets, ld_offsets, grad_logits, ld_logits);
  return mss_launch_status();
}

int mss_msda_prepare_backward_f32(const float* attn_weight, const float* grad_attn, const float* grad_loc,
                                  const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P, float* grad_offsets,
                                  float* grad_logits, void* stream) {
  return mss_msda_prepare_backward_ld_f32(attn_weight, grad_attn, grad_loc, spatial_shapes, N, Lq, M, L, P, grad_offsets,
                                          (long long)M * 2 * L * P, grad_logits, (long long)M * L * P, stream);
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
