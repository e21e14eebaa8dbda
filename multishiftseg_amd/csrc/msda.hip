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
#include <stdlib.h>
#include "../../include/mss_hip.h"

namespace {

constexpr int MSDA_MAX_LP = 20;       // prepare kernels: 256 pairs x 3 x L*P floats of LDS per workgroup (60 KB at 20)
constexpr int MSDA_WIN_MAXL = 8;      // window forward: levels carried by value in the kernel arguments
constexpr int MSDA_WIN_ROWS = 320;    // window forward: value rows (128 B each) of one level staged per workgroup (40 KB)

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };

template <typename T, int LPP>
__device__ __forceinline__ T pair_reduce(T v);     // sum over the LPP lanes of a pair's group (defined with the backward)

// ------------------------------------------------------------------------------------------
// forward, fast path
// FUSED: `loc` / `attn` are the raw sampling offsets and attention logits of the module's two Linears and `ref` the
// reference points [N,Lq,L,2] (ops/modules/ms_deform_attn.py:100-109): the softmax over the L*P logits and the
// location arithmetic loc = ref + offset / (W_l, H_l) happen here, on the values the wave has staged in LDS anyway, so
// the [N,Lq,M,L,P,2] / [N,Lq,M,L,P] tensors (11.7 MB per call at C4, read AND written by a separate kernel) never
// exist. Same arithmetic as msda_prepare_kernel except the order in which the L*P exponentials are added.
template <int LPH, bool FUSED, bool BUF>
__global__ __launch_bounds__(256) void msda_fwd_fast_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ ref, long long npairs, int S,
    int M, int L, int Lq, int P, float* __restrict__ out, long long ldo, long long ldl) {
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

  // stage loc/attn of the wave's pairs. ldo / ldl: floats between the rows of consecutive (n, q) -- M*2LP / M*LP when the two
  // tensors are dense; r04: both projections' outputs side by side in ONE [N*Lq, M*3LP] buffer (a single 288-wide product).
  // The wave's pairs are contiguous in memory when dense, and also when the wave holds exactly the M heads of one query.
  if (M == HPW || (ldo == (long long)M * LP * 2 && ldl == (long long)M * LP)) {
    const float* gl = M == HPW ? loc + pair0 / M * ldo : loc + pair0 * LP * 2;
    const float* ga = M == HPW ? attn + pair0 / M * ldl : attn + pair0 * LP;
    for (int i = lane; i < npw * LP * 2; i += 64) sloc[i] = gl[i];
    for (int i = lane; i < npw * LP; i += 64) sattn[i] = ga[i];
  } else {
    for (int i = lane; i < npw * LP * 2; i += 64) {
      const long long pr = pair0 + i / (LP * 2);
      sloc[i] = loc[pr / M * ldo + (pr % M) * (LP * 2) + i % (LP * 2)];
    }
    for (int i = lane; i < npw * LP; i += 64) {
      const long long pr = pair0 + i / LP;
      sattn[i] = attn[pr / M * ldl + (pr % M) * LP + i % LP];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int g = lane / LPH, j = lane % LPH;
  const bool mine = g < npw;                    // lanes without a pair stay until the wave-level barriers are done
  const long long pair = pair0 + (mine ? g : 0);
  const int m = (int)(pair % M);
  const long long nq = pair / M;
  const int n = (int)(nq / Lq);
  const float* myloc = sloc + g * LP * 2;
  const float* myattn = sattn + g * LP;
  const size_t row_stride = (size_t)M * D;  // floats between consecutive spatial positions
  const float* vbase = value + (size_t)n * S * row_stride + (size_t)m * D + 4 * j;
  if (FUSED && mine) {
    // The LPH lanes of a pair turn its staged raw values into attention weights and locations IN PLACE, each lane taking
    // every LPH-th sample (2 exps + 4 divisions per lane at L*P = 12, instead of every lane redoing all 12): the maximum
    // from LDS, the partial sums of exponentials combined by an LPH-lane shuffle tree. A wave executes in lockstep and
    // LDS operations of one wave complete in order, so the reads of the raw logits precede the overwriting stores.
    float* ml = sloc + g * LP * 2;
    float* ma = sattn + g * LP;
    float mx = -__builtin_huge_valf();
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, ma[i]);
    constexpr int MAXI = (20 + LPH - 1) / LPH;          // L*P <= 20 (checked by the launcher)
    float e[MAXI];
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
      const int i = j + t * LPH;
      e[t] = i < LP ? expf(ma[i] - mx) : 0.f;
      part += e[t];
    }
    const float inv = 1.f / pair_reduce<float, LPH>(part);
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
      const int i = j + t * LPH;
      if (i < LP) {
        const int l = i / P;
        ma[i] = e[t] * inv;
        ml[2 * i] = ref[(nq * L + l) * 2] + ml[2 * i] / (float)shapes[2 * l + 1];
        ml[2 * i + 1] = ref[(nq * L + l) * 2 + 1] + ml[2 * i + 1] / (float)shapes[2 * l];
      }
    }
  }
  if (FUSED) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (!mine) return;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (BUF) {
    // PMC showed this kernel issuing 22 VALU instructions per gather (69 % VALU-issue utilisation next to the L2 gather
    // ceiling; the fused form 89 %): 64-bit address arithmetic per corner and sixteen selects per sample that zero the
    // out-of-image corners. Here the whole value tensor is ONE buffer resource (host-checked: < 4 GB): a corner is a 32-bit
    // byte offset, and an out-of-image corner gets an out-of-range offset, for which the hardware returns zeros without a
    // memory access -- the reference's zero padding, exact even next to non-finite values.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value), 0, (int)(unsigned)min((unsigned long long)npairs / M / Lq * S * row_stride * 4ull, 0xffffffffull),
        0x00020000);
    const unsigned rs4 = (unsigned)(row_stride * sizeof(float));
    const unsigned lane_off = (unsigned)(((size_t)n * S * row_stride + (size_t)m * D + 4 * j) * sizeof(float));
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const unsigned lvl_off = lane_off + (unsigned)starts[l] * rs4;
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
        const unsigned o00 = lvl_off + (unsigned)(h0 * W + w0) * rs4;          // garbage when out of image: replaced below
        const unsigned oob = 0xffffffffu;
        const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh0 && okw0) ? o00 : oob, 0, 0));
        const f32x4 v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh0 && okw1) ? o00 + rs4 : oob, 0, 0));
        const f32x4 v3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh1 && okw0) ? o00 + W * rs4 : oob, 0, 0));
        const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (okh1 && okw1) ? o00 + W * rs4 + rs4 : oob, 0, 0));
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const f32x4 val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        acc += aw * val;
      }
    }
    *reinterpret_cast<f32x4*>(out + pair * D + 4 * j) = acc;
    return;
  }
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
// forward, r04: per-sample RECORDS. In msda_fwd_fast_kernel every one of the LPH lanes that share a (query, head) repeats the
// sample's location arithmetic (pixel coordinates, floor, four weights, four corner offsets with their in-image tests: ~20 of the
// ~40 VALU instructions per sample; the kernel is VALU-issue-bound next to the L2 gather ceiling). Here the
// lanes of a group split the L*P samples between them ONCE: lane j prepares samples j, j + LPH, ... -- it loads their raw values
// itself (no LDS staging of loc / attn), for the fused form takes part in the group's softmax through shuffles, and leaves a
// record in LDS: four byte offsets into the value tensor (out-of-image corners: an out-of-range offset, answered with zeros by
// the buffer hardware), the four bilinear weights, the attention weight. The sampling loop then costs, per sample and lane, three
// LDS reads (group-uniform addresses: broadcasts), four address adds and the 20 multiply-adds of the accumulation.
// Arithmetic per output identical to msda_fwd_fast_kernel's (same expressions in the same order): bit-identical results.
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
// forward through LDS windows (fp32, D = 32). The fast kernel above sits on the L2 row-gather rate (48 x 128-B rows per
// (query, head), 16-17 TB/s measured); in the encoder the queries are the pixels of the levels themselves and their
// samples lie a few pixels around their own position, so neighbouring queries gather the same rows again and again.
// One workgroup = 64 queries (an 8x8 pixel tile of one level when the queries are the pixel grid, else 64 consecutive
// queries) x one head. Per level: the bounding box of the tile's samples is reduced in LDS, a window of at most
// MSDA_WIN_ROWS value rows (the whole box if it fits, else a box of that area around the samples' mean) is staged once
// with coalesced 128-B loads, and the 4*P corner fetches of every query read LDS; a corner outside the window falls
// back to the global gather, so the result never depends on where the window lies. Same arithmetic per sample as the
// fast kernel. Level geometry comes BY VALUE from the host (the grid size depends on it).
// MEASURED (profiles/r02/m2f/bench_msda_window.jsonl): correct but SLOWER than the gather kernel on MI355X -- 0.66-0.81 ms
// against 0.46 ms at N = 16 (C4), 0.18-0.22 against 0.107 at C5 -- so it is opt-in (MSS_MSDA_WINDOW=1) and kept as the
// record of the experiment: 48 LDS b128 reads per (query, head) cost 8 clocks each before bank conflicts (rows of the
// same parity collide), the per-level barriers and box reductions add to that, and the L2 gather already delivers
// 17 TB/s = 43 % of the L1 data path; the ceiling of the LDS route is ~2x, the first implementation is 0.7x.
struct MsdaLevels {
  int L;
  int H[MSDA_WIN_MAXL], W[MSDA_WIN_MAXL];
  int qstart[MSDA_WIN_MAXL];         // first query of the level (grid mode)
  int tiles_x[MSDA_WIN_MAXL];        // 8x8 tiles per row of the level
  int tile_start[MSDA_WIN_MAXL + 1]; // prefix sum of the tile counts
};

template <bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_window_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ starts, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ ref, const MsdaLevels lv, int grid_mode, int ntiles, int S, int M,
    int Lq, int P, float* __restrict__ out) {
  constexpr int D = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int sq[64];
  __shared__ int sstat[MSDA_WIN_MAXL][8];
  const int L = lv.L, LP = L * P;
  float* swin = smem;                       // [MSDA_WIN_ROWS][32]
  float* sx = swin + MSDA_WIN_ROWS * D;     // [64][LP] w_im
  float* sy = sx + 64 * LP;                 // h_im
  float* sa = sy + 64 * LP;                 // attention weight
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int m = b % M;
  b /= M;
  const int tile = b % ntiles, n = b / ntiles;

  if (tid < 64) {
    int q = -1;
    if (grid_mode) {
      int lq = 0;
      while (lq + 1 < L && tile >= lv.tile_start[lq + 1]) ++lq;
      const int t = tile - lv.tile_start[lq];
      const int ty = t / lv.tiles_x[lq], tx = t - ty * lv.tiles_x[lq];
      const int y = ty * 8 + (tid >> 3), x = tx * 8 + (tid & 7);
      if (y < lv.H[lq] && x < lv.W[lq]) q = lv.qstart[lq] + y * lv.W[lq] + x;
    } else {
      q = tile * 64 + tid;
      if (q >= Lq) q = -1;
    }
    sq[tid] = q;
  }
  if (tid < L * 8) {
    const int k = tid & 7;
    sstat[tid >> 3][k] = (k == 0 || k == 2) ? 0x7fffffff : (k == 1 || k == 3) ? (int)0x80000000 : 0;
  }
  __syncthreads();

  // ---- phase 1: the 4 lanes of a query slot turn its L*P raw entries into (w_im, h_im, weight) in LDS
  {
    const int slot = tid >> 2, sub = tid & 3;
    const int q = sq[slot];
    const long long nq = (long long)n * Lq + max(q, 0);
    const long long pair = nq * M + m;
    const float* gl = loc + pair * LP * 2;
    const float* ga = attn + pair * LP;
    constexpr int MAXI = (MSDA_MAX_LP + 3) / 4;
    float lx[MAXI], ly[MAXI], aw[MAXI];
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      const int k = sub + 4 * u;
      if (k < LP) {
        lx[u] = gl[2 * k], ly[u] = gl[2 * k + 1], aw[u] = ga[k];
        mx = fmaxf(mx, aw[u]);
      }
    }
    if (FUSED) {
      mx = fmaxf(mx, __shfl_xor(mx, 1));
      mx = fmaxf(mx, __shfl_xor(mx, 2));
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < MAXI; ++u)
        if (sub + 4 * u < LP) {
          aw[u] = expf(aw[u] - mx);
          part += aw[u];
        }
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      const float inv = 1.f / part;
#pragma unroll
      for (int u = 0; u < MAXI; ++u) {
        const int k = sub + 4 * u;
        if (k < LP) {
          const int l = k / P;
          aw[u] *= inv;
          lx[u] = ref[(nq * L + l) * 2] + lx[u] / (float)lv.W[l];
          ly[u] = ref[(nq * L + l) * 2 + 1] + ly[u] / (float)lv.H[l];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      const int k = sub + 4 * u;
      if (k < LP) {
        const int l = k / P;
        sx[slot * LP + k] = q >= 0 ? lx[u] * lv.W[l] - 0.5f : -8.f;
        sy[slot * LP + k] = q >= 0 ? ly[u] * lv.H[l] - 0.5f : -8.f;
        sa[slot * LP + k] = q >= 0 ? aw[u] : 0.f;
      }
    }
  }
  __syncthreads();

  const int g = tid >> 3, j = tid & 7;
  const size_t row_stride = (size_t)M * D;
  const float* vbase = value + (size_t)n * S * row_stride + (size_t)m * D + 4 * j;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int q0 = sq[g], q1 = sq[g + 32];

  for (int l = 0; l < L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    // ---- bounding box and mean of the tile's samples on this level
    {
      int lo_h = 0x7fffffff, hi_h = (int)0x80000000, lo_w = 0x7fffffff, hi_w = (int)0x80000000, cnt = 0;
      float sh = 0.f, sw = 0.f;
      for (int i = tid; i < 64 * P; i += 256) {
        const int slot = i / P, pt = i - slot * P;
        const float w_im = sx[slot * LP + l * P + pt], h_im = sy[slot * LP + l * P + pt];
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
          lo_h = min(lo_h, max(h0, 0)), hi_h = max(hi_h, min(h0 + 1, H - 1));
          lo_w = min(lo_w, max(w0, 0)), hi_w = max(hi_w, min(w0 + 1, W - 1));
          sh += h_im, sw += w_im, ++cnt;
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        lo_h = min(lo_h, __shfl_xor(lo_h, o)), hi_h = max(hi_h, __shfl_xor(hi_h, o));
        lo_w = min(lo_w, __shfl_xor(lo_w, o)), hi_w = max(hi_w, __shfl_xor(hi_w, o));
        cnt += __shfl_xor(cnt, o), sh += __shfl_xor(sh, o), sw += __shfl_xor(sw, o);
      }
      if ((tid & 63) == 0 && cnt > 0) {
        atomicMin(&sstat[l][0], lo_h), atomicMax(&sstat[l][1], hi_h);
        atomicMin(&sstat[l][2], lo_w), atomicMax(&sstat[l][3], hi_w);
        atomicAdd(&sstat[l][4], cnt);
        atomicAdd(reinterpret_cast<float*>(&sstat[l][5]), sh), atomicAdd(reinterpret_cast<float*>(&sstat[l][6]), sw);
      }
    }
    __syncthreads();
    // ---- the window (block-uniform)
    int wh0 = 0, ww0 = 0, nh = 0, nw = 0;
    {
      const int cnt = sstat[l][4];
      if (cnt > 0) {
        const int lo_h = sstat[l][0], hi_h = sstat[l][1], lo_w = sstat[l][2], hi_w = sstat[l][3];
        const int bh = hi_h - lo_h + 1, bw = hi_w - lo_w + 1;
        nh = bh, nw = bw, wh0 = lo_h, ww0 = lo_w;
        if (bh * bw > MSDA_WIN_ROWS) {
          const float f = sqrtf((float)MSDA_WIN_ROWS / ((float)bh * (float)bw));
          nh = min(bh, max(2, (int)(bh * f)));
          nw = min(bw, MSDA_WIN_ROWS / nh);
          nh = min(bh, MSDA_WIN_ROWS / nw);
          const float mh = __int_as_float(sstat[l][5]) / cnt, mw = __int_as_float(sstat[l][6]) / cnt;
          wh0 = min(max((int)floorf(mh + 1.f) - nh / 2, lo_h), hi_h - nh + 1);
          ww0 = min(max((int)floorf(mw + 1.f) - nw / 2, lo_w), hi_w - nw + 1);
        }
      }
    }
    const float* vl = vbase + (size_t)starts[l] * row_stride;
    {
      const int rows = nh * nw;
      constexpr int UN = MSDA_WIN_ROWS / 32;
      f32x4 t[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int r = g + 32 * u;
        if (r < rows) {
          const int hy = r / nw, wx = r - hy * nw;
          t[u] = *reinterpret_cast<const f32x4*>(vl + (size_t)((wh0 + hy) * W + ww0 + wx) * row_stride);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int r = g + 32 * u;
        if (r < rows) *reinterpret_cast<f32x4*>(swin + r * D + 4 * j) = t[u];
      }
    }
    __syncthreads();
    // ---- sampling: the thread's two queries, P points each. (A branch-free variant -- LDS read with a clamped index plus
    // a buffer load pushed out of range for in-window corners -- was 3x slower: the out-of-range buffer loads still
    // occupy the address path, 2.04 ms against 0.66 ms at N = 16.)
    auto fetch = [&](int hc, int wc) -> f32x4 {
      const int dh = hc - wh0, dw = wc - ww0;
      if ((unsigned)dh < (unsigned)nh && (unsigned)dw < (unsigned)nw)
        return *reinterpret_cast<const f32x4*>(swin + (dh * nw + dw) * D + 4 * j);
      return *reinterpret_cast<const f32x4*>(vl + (size_t)(hc * W + wc) * row_stride);
    };
    auto sample = [&](int slot, f32x4& acc) {
      for (int pt = 0; pt < P; ++pt) {
        const float w_im = sx[slot * LP + l * P + pt], h_im = sy[slot * LP + l * P + pt];
        const float aw = sa[slot * LP + l * P + pt];
        const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        if (!inside) continue;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h0 = (int)hf, w0 = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool okh0 = h0 >= 0, okh1 = h0 + 1 <= H - 1, okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
        const int h0c = max(h0, 0), h1c = min(h0 + 1, H - 1), w0c = max(w0, 0), w1c = min(w0 + 1, W - 1);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = fetch(h0c, w0c), v2 = fetch(h0c, w1c), v3 = fetch(h1c, w0c), v4 = fetch(h1c, w1c);
        v1 = (okh0 && okw0) ? v1 : z;
        v2 = (okh0 && okw1) ? v2 : z;
        v3 = (okh1 && okw0) ? v3 : z;
        v4 = (okh1 && okw1) ? v4 : z;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const f32x4 val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        acc += aw * val;
      }
    };
    if (q0 >= 0) sample(g, acc0);
    if (q1 >= 0) sample(g + 32, acc1);
  }
  if (q0 >= 0) *reinterpret_cast<f32x4*>(out + (((long long)n * Lq + q0) * M + m) * D + 4 * j) = acc0;
  if (q1 >= 0) *reinterpret_cast<f32x4*>(out + (((long long)n * Lq + q1) * M + m) * D + 4 * j) = acc1;
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

// grad_value without global atomics (fp32, D = 32). The scatter-add formulation above is bound by memory-side
// atomics (~1.3 TB/s of added bytes: 6 ms for the 8 GB of a 16-image C4 call). Here a workgroup OWNS a tile of
// grad_value -- (image n, head m, level l, a band of rows x columns with at most 256 positions) -- scans the level's
// Lq*P sampling locations of (n, m) (L2-resident, a few hundred KB), and for the samples whose bilinear footprint
// touches its tile accumulates w_corner * attn * grad_out in LDS; the finished tile is written with plain stores.
// Every grad_value element belongs to exactly one tile, so there is no memset and no global atomic. Passing samples
// are compacted with a wave ballot and handled two at a time (32 channels each), four pairs in flight.
//
// The LDS accumulators are 64-bit FIXED POINT: ds_add_f32 runs at 0.33 lane-operations per clock per CU on gfx950
// (measured, tools/hipbench/lds_atomic_rate.hip), ds_add_u64 at 9.2. A pre-pass finds max|grad_out| and max|attn|;
// with 2^e >= their product every contribution is scaled by 2^(40-e), so |c| <= 2^40, 2^22 of them cannot overflow
// and the resolution is 2^-40 of the largest possible contribution (fp32 atomics resolve 2^-24 of each partial sum).
// The result does not depend on the order of the additions.
constexpr int MSDA_TILE_CELLS = 256;   // x 32 channels x 8 B = 64 KB -> two workgroups per CU
constexpr int MSDA_TILE_W = 16;
constexpr int MSDA_FIXED_BITS = 40;

__global__ __launch_bounds__(256) void msda_absmax_kernel(const float* __restrict__ a, long long na,
                                                          const float* __restrict__ b, long long nb,
                                                          unsigned* __restrict__ out) {
  float ma = 0.f, mb = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na; i += (long long)gridDim.x * 256) ma = fmaxf(ma, fabsf(a[i]));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (long long)gridDim.x * 256) mb = fmaxf(mb, fabsf(b[i]));
  ma = mss_wave_max(ma);
  mb = mss_wave_max(mb);
  if ((threadIdx.x & 63) == 0) {       // non-negative floats order like their bit patterns
    atomicMax(out, __float_as_uint(ma));
    atomicMax(out + 1, __float_as_uint(mb));
  }
}

// one atomicMax per WORKGROUP, and only when it can raise the value (same-address atomics serialise: 8192 of them cost
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

struct MsdaTile { int l, H, W, r0, r1, c0, c1; long long start; bool valid; };
__device__ __forceinline__ MsdaTile msda_find_tile(const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
                                                   int L, int t) {
  MsdaTile k;
  k.valid = false;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    // ~16 x 16 positions (a bilinear footprint straddles a tile edge with probability ~1/16 per axis; row bands would
    // duplicate half of the samples); thin levels get wider / taller tiles so that a tile still holds ~256 positions
    const int BH0 = H < MSDA_TILE_W ? H : MSDA_TILE_W;
    const int BW = min(W, MSDA_TILE_CELLS / BH0);
    const int BH = min(H, MSDA_TILE_CELLS / BW);
    const int nr = (H + BH - 1) / BH, nc = (W + BW - 1) / BW;
    if (t < nr * nc) {
      const int br = t / nc, bc = t - br * nc;
      k.l = l; k.H = H; k.W = W; k.start = starts[l];
      k.r0 = br * BH; k.r1 = min(H, k.r0 + BH);
      k.c0 = bc * BW; k.c1 = min(W, k.c0 + BW);
      k.valid = true;
      return k;
    }
    t -= nr * nc;
  }
  return k;
}

constexpr int MSDA_LDS_NT = 1024;      // 16 waves: the gout gathers are latency-bound, two such workgroups per CU
template <int UN>                      // UN sample pairs in flight per wave
__global__ __launch_bounds__(MSDA_LDS_NT) void msda_bwd_value_lds_kernel(
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ gout, const unsigned* __restrict__ absmax, int S, int M,
    int L, int Lq, int P, float* __restrict__ gvalue) {
  constexpr int D = 32, NT = MSDA_LDS_NT;
  __shared__ unsigned long long tile[MSDA_TILE_CELLS * D];
  const MsdaTile k = msda_find_tile(shapes, starts, L, blockIdx.x);
  if (!k.valid) return;                                  // the grid is an upper bound on the tile count
  const int m = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, d = lane & 31;
  const int tw = k.c1 - k.c0, cells = (k.r1 - k.r0) * tw;
  for (int i = tid; i < cells * D; i += NT) tile[i] = 0ull;
  // fixed-point scale: 2^(40 - e) with 2^e >= max|grad_out| * max|attn| (bilinear weights are <= 1)
  const float bound = __uint_as_float(absmax[0]) * __uint_as_float(absmax[1]);
  int e = 0;
  if (bound > 0.f && bound < __builtin_huge_valf()) (void)frexpf(bound, &e);      // bound = f * 2^e, f in [0.5, 1)
  // a non-finite grad_out / attn poisons the whole gradient (the reference's float atomics would poison the cells it
  // reaches): every element written below becomes NaN instead of a silently wrong finite number
  const bool finite = bound < __builtin_huge_valf();      // false for inf and NaN
  const double to_fixed = ldexp(1.0, MSDA_FIXED_BITS - e);
  const double from_fixed = finite ? ldexp(1.0, e - MSDA_FIXED_BITS) : (double)__builtin_nanf("");
  __syncthreads();
  const long long total = (long long)Lq * P;
  const size_t pair_stride = (size_t)M * L * P;          // (q -> q+1) in units of samples
  const size_t pair0 = ((size_t)n * Lq * M + m) * L * P + (size_t)k.l * P;
  const float fH = (float)k.H, fW = (float)k.W;
  const float* go = gout + ((size_t)n * Lq * M + m) * D + d;       // + q * M * D
  const size_t go_stride = (size_t)M * D;
  const int total_i = (int)total;
  for (int i0 = 0; i0 < total_i; i0 += NT) {
    const int i = i0 + tid;
    bool pass = false;
    int q = 0;
    float h_im = 0.f, w_im = 0.f, aw = 0.f;
    if (i < total_i) {
      q = i / P;
      const int pt = i - q * P;
      const size_t sidx = pair0 + (size_t)q * pair_stride + pt;
      const float lx = loc[sidx * 2], ly = loc[sidx * 2 + 1];
      w_im = lx * fW - 0.5f;
      h_im = ly * fH - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW) {
        const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
        const bool rows = (h0 >= k.r0 && h0 < k.r1) || (h0 + 1 >= k.r0 && h0 + 1 < k.r1);     // h0 = -1 / h0+1 = H never match
        const bool cols = (w0 >= k.c0 && w0 < k.c1) || (w0 + 1 >= k.c0 && w0 + 1 < k.c1);
        pass = rows && cols;
        if (pass) aw = attn[sidx];
      }
    }
    unsigned long long mask = __ballot(pass);
    while (mask) {
      // up to UN pairs of passing samples: lanes 0-31 take the even ones, lanes 32-63 the odd ones; all gout rows are
      // requested before the first LDS atomic
      int sq[UN];
      float sh[UN], sw[UN], tgv[UN];
      bool act[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        int s0 = -1, s1 = -1;
        if (mask) { s0 = __ffsll((long long)mask) - 1; mask &= mask - 1; }
        if (mask) { s1 = __ffsll((long long)mask) - 1; mask &= mask - 1; }
        const int src = half ? s1 : s0;
        act[u] = src >= 0;
        const int from = src >= 0 ? src : 0;
        sq[u] = __shfl(q, from);
        sh[u] = __shfl(h_im, from);
        sw[u] = __shfl(w_im, from);
        tgv[u] = __shfl(aw, from);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) tgv[u] *= act[u] ? go[(size_t)sq[u] * go_stride] : 0.f;
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (act[u]) {
          const float hf = floorf(sh[u]), wf = floorf(sw[u]);
          const int h0 = (int)hf, w0 = (int)wf;
          const float lh = sh[u] - hf, lw = sw[u] - wf, hh = 1.f - lh, hw = 1.f - lw;
          const bool r_lo = h0 >= k.r0 && h0 < k.r1, r_hi = h0 + 1 >= k.r0 && h0 + 1 < k.r1;
          const bool c_lo = w0 >= k.c0 && w0 < k.c1, c_hi = w0 + 1 >= k.c0 && w0 + 1 < k.c1;
          unsigned long long* base = tile + ((h0 - k.r0) * tw + (w0 - k.c0)) * D + d;
          // round(c * 2^(40-e)) as a two's-complement integer: adding 1.5 * 2^52 leaves it in the low mantissa bits
          // (|c * to_fixed| <= 2^40); a native double -> int64 conversion does not exist on this ISA
          auto fx = [&](float c) {
            const double t = fma((double)c, to_fixed, 6755399441055744.0);
            return (unsigned long long)(__double_as_longlong(t) - 0x4338000000000000ll);
          };
          if (r_lo && c_lo) atomicAdd(base, fx(hh * hw * tgv[u]));
          if (r_lo && c_hi) atomicAdd(base + D, fx(hh * lw * tgv[u]));
          if (r_hi && c_lo) atomicAdd(base + tw * D, fx(lh * hw * tgv[u]));
          if (r_hi && c_hi) atomicAdd(base + tw * D + D, fx(lh * lw * tgv[u]));
        }
      }
    }
  }
  __syncthreads();
  const size_t rs = (size_t)M * D;
  float* gv = gvalue + (size_t)n * S * rs + (size_t)m * D + (size_t)k.start * rs;
  for (int i = tid; i < cells * D; i += NT) {
    const int cell = i >> 5, ch = i & 31;
    const int r = cell / tw, c = cell - r * tw;
    gv[(size_t)((k.r0 + r) * k.W + k.c0 + c) * rs + ch] = (float)((double)(long long)tile[i] * from_fixed);
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
// grad_value, CELL-SORTED path (round 5). The binned kernel above spends 15 of its 19 VALU slots per
// sample on making the accumulation order-independent (two 64-bit fixed-point conversions + two LDS atomics per lane).
// Here the counting sort's key is the CELL: key = (image, head, level, home cell, q mod sub) -- `sub` sub-keys per cell
// on the levels that receive many samples per cell, so that a key holds ~5 - 25 records -- and one WAVE owns one ROW of
// cells of one (image, head, level):
//   msda_cell_kernel<0>      per-key record counts (one global atomic per in-image sample)
//   msda_scan_*_kernel       exclusive scan of the keys (two launches)
//   msda_cell_kernel<1>      the 16-byte records to their keys' runs (position inside a run = arrival order)
//   msda_bwd_value_rows_kernel  a wave walks its row's keys in blocks of whole runs with <= 64 records: lane = record
//                            computes the four corner weights and the RANK of its record among its run's records by
//                            (query, point) -- the run is consumed in that order, so the fp32 sums do not depend on the
//                            arrival order --, stages weights / grad_out row offsets in LDS at the ranked position, then
//                            lanes = (column, channel) add the records one by one (two FMAs per record) into two
//                            registers. At a cell boundary the left column's sums and the previous cell's right-column
//                            sums give the cell's "upper" part (stored to grad_value) and the "lower" part of the cell
//                            below (stored to a side tensor, the next row belongs to another wave).
//   msda_rows_merge_kernel   grad_value += lower parts
// A run longer than 64 records (never at uniform locations) is added in 64-bit fixed point scaled by its own per-lane
// maximum, which is order-independent too. No LDS atomics, no fixed-point conversion on the common path, bit-reproducible.
// MEASURED AND NOT ADOPTED (opt-in, MSS_MSDA_BWD_ROWS=1; profiles/r05/msda_bwd_rows.md): at C4 N = 16 the per-sample global
// atomics of the two sort passes cost 0.53 + 0.98 ms (the binned path's chunk-level LDS histograms: 0.08 + 0.14 -- a chunk
// of 4096 samples meets 54 x 8 tile keys but 93 k cell keys, nothing to aggregate), and the row kernel takes 0.65 ms, what
// the binned accumulation takes: the two FMAs per record come with ~15 instructions of run bookkeeping, rank loop and
// staging per record, and three dependent global latencies per block of 64 records.
constexpr int MSDA_ROW_MAXL = 8;
// sh: log2(sub-keys per cell); kstart: first key inside one (image, head); cstart: first cell inside one image of the side tensor
struct MsdaRowLevel { int H, W, sh, kstart, cstart, item0; };
struct MsdaRowGeom { MsdaRowLevel lv[MSDA_ROW_MAXL]; int L, KS, rows, cells; };

static bool msda_row_geom(const int64_t* hs, int L, int Lq, int P, MsdaRowGeom& g) {
  if (L < 1 || L > MSDA_ROW_MAXL) return false;
  g.L = L;
  long long ks = 0, rows = 0, cells = 0;
  double weight[MSDA_ROW_MAXL];
  for (int l = 0; l < L; ++l) {
    const long long H = hs[2 * l], W = hs[2 * l + 1];
    if (H < 1 || W < 1 || H > 32767 || W > 32767) return false;
    const double density = (double)Lq * P / ((double)H * W);
    int sh = 0;
    while (sh < 4 && density / (1 << sh) > 24.0) ++sh;
    g.lv[l].H = (int)H; g.lv[l].W = (int)W; g.lv[l].sh = sh;
    if (ks + ((H * W) << sh) >= (1ll << 30)) return false;
    g.lv[l].kstart = (int)ks;
    g.lv[l].cstart = (int)cells;
    ks += (H * W) << sh;
    cells += H * W;
    weight[l] = density * (double)W;                        // records per row: the heaviest rows are handed out first
    rows += H;
  }
  bool done[MSDA_ROW_MAXL] = {false};
  int item0 = 0;
  for (int k = 0; k < L; ++k) {
    int best = -1;
    for (int l = 0; l < L; ++l) if (!done[l] && (best < 0 || weight[l] > weight[best])) best = l;
    done[best] = true;
    g.lv[best].item0 = item0;
    item0 += g.lv[best].H;
  }
  for (int l = L; l < MSDA_ROW_MAXL; ++l) { g.lv[l] = g.lv[L - 1]; g.lv[l].item0 = 0x7fffffff; }
  g.KS = (int)ks;
  g.rows = (int)rows;
  g.cells = (int)cells;
  return true;
}

// A record: header = kill << 31 | query << 4 | point, then A, B, lw as in the binned path (corners outside the image folded
// into the weights, so the consumer adds all four unconditionally). MODE 0: count. MODE 1: scatter.
template <int MODE>
__global__ __launch_bounds__(256) void msda_cell_kernel(MsdaRowGeom g, const float* __restrict__ loc, const float* __restrict__ attn,
                                                        int M, int Lq, int P, long long per_image, int* __restrict__ counts_or_cursor,
                                                        f32x4* __restrict__ records) {
  __shared__ MsdaRowLevel lv[MSDA_ROW_MAXL];
  const int tid = threadIdx.x, n = blockIdx.y;
  if (tid < g.L) lv[tid] = g.lv[tid];
  __syncthreads();
  const int L = g.L;
  int* gk = counts_or_cursor + (size_t)n * M * g.KS;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long s = ((long long)blockIdx.x * 4 + it) * 256 + tid;
    if (s >= per_image) break;
    const long long gs = (long long)n * per_image + s;
    const float lx = loc[gs * 2], ly = loc[gs * 2 + 1];
    const unsigned su = (unsigned)s;                         // per_image < 2^31 (host-checked)
    const unsigned sp = su / (unsigned)P, p = su - sp * (unsigned)P;
    const unsigned l = sp % (unsigned)L;
    const unsigned sm = sp / (unsigned)L;
    const unsigned m = sm % (unsigned)M, q = sm / (unsigned)M;
    const MsdaRowLevel v = lv[l];
    const float w_im = __fmaf_rn(lx, (float)v.W, -0.5f), h_im = __fmaf_rn(ly, (float)v.H, -0.5f);   // one rounding in both modes
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)v.H && w_im < (float)v.W)) continue;
    const float hf = floorf(h_im), wf = floorf(w_im);
    int h0 = (int)hf, w0 = (int)wf;
    const int hc = max(h0, 0), wc = max(w0, 0);
    const int key = (int)m * g.KS + v.kstart + ((hc * v.W + wc) << v.sh) + (int)(q & ((1u << v.sh) - 1u));
    if (MODE == 0) {
      (void)__hip_atomic_fetch_add(gk + key, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int pos = __hip_atomic_fetch_add(gk + key, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float aw = attn[gs];
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      float A, B, lwq;
      unsigned kill;
      if (h0 < 0) { A = aw * lh; B = 0.f; }                               // row -1 is outside: row 0 moves up
      else { A = aw * hh; B = (h0 + 1 <= v.H - 1) ? aw * lh : 0.f; }
      if (w0 < 0) { lwq = hw; kill = 1u; }                                // column 0 takes 1 - (1 - lw)
      else { lwq = lw; kill = (w0 + 1 <= v.W - 1) ? 0u : 1u; }
      f32x4 r;
      r.x = __uint_as_float((kill << 31) | (q << 4) | p);
      r.y = A; r.z = B; r.w = lwq;
      records[pos] = r;
    }
  }
}

// exclusive scan of counts[n] -> offsets[n + 1] and cursor[n], 4096 elements per workgroup: block sums, then each
// workgroup adds up the sums in front of it (a few hundred) and scans its own elements
__global__ __launch_bounds__(1024) void msda_scan_sums_kernel(const int* __restrict__ counts, int n, int* __restrict__ bsum) {
  __shared__ int red[16];
  const int tid = threadIdx.x, base = blockIdx.x * 4096 + tid * 4;
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) sum += base + i < n ? counts[base + i] : 0;
#pragma unroll
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int i = 0; i < 16; ++i) t += red[i];
    bsum[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void msda_scan_apply_kernel(const int* __restrict__ counts, int n, const int* __restrict__ bsum,
                                                               int* __restrict__ offsets, int* __restrict__ cursor) {
  __shared__ int part[1024];
  __shared__ int red[16];
  const int tid = threadIdx.x, base = blockIdx.x * 4096 + tid * 4;
  int pre = 0;
  for (int i = tid; i < (int)blockIdx.x; i += 1024) pre += bsum[i];
#pragma unroll
  for (int o = 32; o; o >>= 1) pre += __shfl_xor(pre, o);
  if ((tid & 63) == 0) red[tid >> 6] = pre;
  int c[4], sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i] = base + i < n ? counts[base + i] : 0; sum += c[i]; }
  part[tid] = sum;
  __syncthreads();
  int front = 0;
  for (int i = 0; i < 16; ++i) front += red[i];
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = front + part[tid] - sum;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (base + i < n) { offsets[base + i] = run; cursor[base + i] = run; }
    run += c[i];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 1023) offsets[n] = front + part[1023];
}

template <int UN>
__global__ __launch_bounds__(256) void msda_bwd_value_rows_kernel(MsdaRowGeom g, const int64_t* __restrict__ starts,
                                                                  const float* __restrict__ gout, const f32x4* __restrict__ records,
                                                                  const int* __restrict__ offsets, int* __restrict__ ticket, int S, int M,
                                                                  int Lq, int N, float* __restrict__ gvalue, float* __restrict__ gbot) {
  constexpr int D = 32;
  __shared__ MsdaRowLevel slv[MSDA_ROW_MAXL];
  __shared__ f32x4 sw_all[4][64];
  __shared__ unsigned srow_all[4][64];
  __shared__ unsigned skey_all[4][64];
  __shared__ int soff_all[4][64];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, d = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < g.L) slv[tid] = g.lv[tid];
  __syncthreads();
  f32x4* sw = sw_all[wave];
  unsigned* srow = srow_all[wave];
  unsigned* skey = skey_all[wave];
  int* soff = soff_all[wave];
  const float* swf = reinterpret_cast<const float*>(sw) + half * 2;        // this lane's column: (upper, lower) weight of record i at swf[4 i]
  const int NM = N * M, nitems = g.rows * NM;
  const size_t rs = (size_t)M * D;
  const unsigned rs4 = (unsigned)(rs * sizeof(float)), d4 = (unsigned)d * 4u;
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(gout), 0, (int)(unsigned)min((unsigned long long)N * Lq * rs * 4ull, 0xffffffffull), 0x00020000);
  const f32x4 zero_rec = {0.f, 0.f, 0.f, 0.f};
  const int swap_addr = (lane ^ 32) << 2;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(ticket, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= nitems) break;
    const int irow = item / NM, nm = item - irow * NM;       // row-major over the levels, heaviest level first
    int l = 0;
    for (int k = 1; k < g.L; ++k) if (irow >= slv[k].item0 && irow < slv[k].item0 + slv[k].H) l = k;
    const MsdaRowLevel v = slv[l];
    const int r = irow - v.item0, n = nm / M, m = nm - n * M;
    const int nk = v.W << v.sh;
    const unsigned submask = (1u << v.sh) - 1u;
    const int* offp = offsets + (size_t)nm * g.KS + v.kstart + (size_t)((r * v.W) << v.sh);
    const unsigned go_nm = (unsigned)(((size_t)n * Lq * M + m) * D * sizeof(float));
    float* gv = gvalue + ((size_t)n * S + (size_t)starts[l] + (size_t)r * v.W) * rs + (size_t)m * D + d;
    float* gb = gbot + ((size_t)n * g.cells + (size_t)v.cstart + (size_t)(r + 1) * v.W) * rs + (size_t)m * D + d;
    const bool below = r + 1 < v.H;
    float accT = 0.f, accB = 0.f, prevT = 0.f, prevB = 0.f;
    int c = 0, kpos = 0;
    // the cell is complete: lanes 0-31 hold its left-column sums (its own upper part and the lower part of the cell below),
    // lanes 32-63 the right-column sums, which belong to the NEXT cell of this row and the one below that
    auto flush = [&]() {
      const float pT = __int_as_float(__builtin_amdgcn_ds_bpermute(swap_addr, __float_as_int(prevT)));
      const float pB = __int_as_float(__builtin_amdgcn_ds_bpermute(swap_addr, __float_as_int(prevB)));
      if (half == 0) {
        gv[(size_t)c * rs] = accT + pT;
        if (below) gb[(size_t)c * rs] = accB + pB;
      }
      prevT = accT; prevB = accB;
      accT = 0.f; accB = 0.f;
      ++c;
    };
    // record -> staging slot `pos`: the four corner weights and the byte offset of its grad_out row
    auto stage = [&](const f32x4& rec, int pos) {
      const unsigned hd = __float_as_uint(rec.x);
      const float lwv = rec.w, lwk = (hd >> 31) ? 0.f : lwv, hwv = 1.f - lwv;
      f32x4 w;
      w.x = rec.y * hwv; w.y = rec.z * hwv;                  // left column: upper, lower
      w.z = rec.y * lwk; w.w = rec.z * lwk;                  // right column
      sw[pos] = w;
      srow[pos] = ((hd >> 4) & 0x7ffffffu) * rs4;
    };
    while (kpos < nk) {
      const int o = offp[min(kpos + lane, nk)];
      const int o0 = __builtin_amdgcn_readfirstlane(o);
      const int rem = min(63, nk - kpos);
      const unsigned long long fits = __ballot(lane <= rem && o - o0 <= 64);        // a prefix of the lanes (lane 0 always)
      int t = __popcll(fits) - 1;
      if (t == 0) {
        // ---- one key with more than 64 records: order-independent 64-bit fixed-point sums, scaled per lane ----
        const int beg = o0, end = __builtin_amdgcn_readlane(o, 1);
        float mx = 0.f;
        bool bad = false;
        long long sT = 0, sB = 0;
        double to_fixed = 0.0, from_fixed = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
          for (int b = beg; b < end; b += 64) {
            stage(b + lane < end ? records[b + lane] : zero_rec, lane);
            const int cntb = min(64, end - b);
            for (int i = 0; i < cntb; ++i) {
              const float gq = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, srow[i] + d4, go_nm, 0));
              const float cT = swf[4 * i] * gq, cB = swf[4 * i + 1] * gq;
              if (pass == 0) {
                mx = fmaxf(mx, fmaxf(fabsf(cT), fabsf(cB)));
                // (an exponent test, not c - c == 0: the compiler contracts that with the product above into an fma = the rounding error)
                bad |= (__float_as_uint(cT) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(cB) & 0x7f800000u) == 0x7f800000u;
              } else {
                sT += __double_as_longlong(fma((double)cT, to_fixed, 6755399441055744.0)) - 0x4338000000000000ll;
                sB += __double_as_longlong(fma((double)cB, to_fixed, 6755399441055744.0)) - 0x4338000000000000ll;
              }
            }
          }
          if (pass == 0) {
            int e = 0;
            if (mx > 0.f && !bad) (void)frexpf(mx, &e);
            to_fixed = bad ? 0.0 : ldexp(1.0, 30 - e);
            from_fixed = ldexp(1.0, e - 30);
          }
        }
        accT += bad ? __builtin_nanf("") : (float)((double)sT * from_fixed);
        accB += bad ? __builtin_nanf("") : (float)((double)sB * from_fixed);
        if (((unsigned)(kpos + 1) & submask) == 0u) flush();
        kpos += 1;
        continue;
      }
      const int cnt = __builtin_amdgcn_readlane(o, t) - o0;
      if (cnt > 0) {
        soff[lane] = o - o0;
        const f32x4 rec = lane < cnt ? records[o0 + lane] : zero_rec;
        // this record's key: the last one that starts at or before it
        int lo = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1) {
          const int mid = lo + step;
          if (mid < t && soff[mid] <= lane) lo = mid;
        }
        const int rs_ = soff[lo], re_ = soff[lo + 1];
        const unsigned mykey = lane < cnt ? (__float_as_uint(rec.x) & 0x7fffffffu) : 0xffffffffu;
        skey[lane] = mykey;
        int len = lane < cnt ? re_ - rs_ : 0;
#pragma unroll
        for (int of = 32; of; of >>= 1) len = max(len, __shfl_xor(len, of));
        const int maxr = __builtin_amdgcn_readfirstlane(len);
        int rank = 0;
        for (int j = 0; j < maxr; ++j) {
          const int pj = rs_ + j;
          rank += (pj < re_ && skey[pj & 63] < mykey) ? 1 : 0;
        }
        stage(rec, lane < cnt ? rs_ + rank : lane);
      }
      // the records in (key, query, point) order: two FMAs per record and lane
      int kk = 0;
      int e_u = __builtin_amdgcn_readlane(o, 1) - o0;
      auto end_key = [&]() {
        if (((unsigned)(kpos + kk + 1) & submask) == 0u) flush();
        ++kk;
        e_u = kk < t ? __builtin_amdgcn_readlane(o, kk + 1) - o0 : 0x7fffffff;
      };
      for (int i0 = 0; i0 < cnt; i0 += UN) {
        float gq[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
          if (i0 + u < cnt) gq[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, srow[i0 + u] + d4, go_nm, 0));
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = i0 + u;
          if (i < cnt) {
            while (i == e_u) end_key();
            accT = __fmaf_rn(swf[4 * i], gq[u], accT);
            accB = __fmaf_rn(swf[4 * i + 1], gq[u], accB);
          }
        }
      }
      while (kk < t) end_key();
      kpos += t;
    }
  }
}

// grad_value[rows 1 .. H-1 of every level] += the lower parts the rows above them left in `gbot`
__global__ __launch_bounds__(256) void msda_rows_merge_kernel(MsdaRowGeom g, const int64_t* __restrict__ starts, int S, int M, int N,
                                                            const float* __restrict__ gbot, float* __restrict__ gvalue) {
  const int l = blockIdx.z, n = blockIdx.y;
  const MsdaRowLevel v = g.lv[l];
  const size_t rs4 = (size_t)M * 8;                          // float4s per cell
  const size_t total = (size_t)(v.H - 1) * v.W * rs4;
  const f32x4* b4 = reinterpret_cast<const f32x4*>(gbot) + ((size_t)n * g.cells + (size_t)v.cstart + (size_t)v.W) * rs4;
  f32x4* g4 = reinterpret_cast<f32x4*>(gvalue) + ((size_t)n * S + (size_t)starts[l] + (size_t)v.W) * rs4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    f32x4 a = g4[i];
    const f32x4 b = b4[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    g4[i] = a;
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
int msda_forward_fast(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc,
                      const float* attn, const float* ref, int N, int S, int M, int L, int Lq, int P, float* out,
                      hipStream_t stream, long long ldo = 0, long long ldl = 0, float* loc_out = nullptr, float* attn_out = nullptr) {
  constexpr int HPW = 64 / LPH;
  const long long npairs = (long long)N * Lq * M;
  const long long nblocks = (npairs + 4 * HPW - 1) / (4 * HPW);
  if (ldo <= 0) ldo = (long long)M * L * P * 2;
  if (ldl <= 0) ldl = (long long)M * L * P;
  const size_t smem = (size_t)4 * HPW * L * P * 3 * sizeof(float);
  // buffer-resource addressing (32-bit offsets, hardware zero fill) when the value tensor is below 4 GB; MSS_MSDA_BUF=0: A/B
  const bool buf = (unsigned long long)N * S * M * (4 * LPH) * 4ull < 0xffffffffull && MSS_ENV_INT("MSS_MSDA_BUF", 1) != 0;
#define MSDA_LAUNCH(FUSED_, BUF_)                                                                                              \
  hipLaunchKernelGGL((msda_fwd_fast_kernel<LPH, FUSED_, BUF_>), dim3((unsigned)nblocks), dim3(256), smem, stream, value, shapes, \
                     starts, loc, attn, ref, npairs, S, M, L, Lq, P, out, ldo, ldl)
  // r04: per-sample records prepared once per (query, head) group (msda_fwd_rec_kernel); needs the buffer addressing and, for
  // the 0xffffff00 out-of-range marker, a tensor below that size. MSS_MSDA_REC=0: the round-2/3 kernel (A/B, tests)
  const size_t smem_rec = (size_t)4 * HPW * (L * P + 1) * 9 * sizeof(float);
  if (buf && (unsigned long long)N * S * M * (4 * LPH) * 4ull < 0xffffff00ull && smem_rec <= 65536 &&
      MSS_ENV_INT("MSS_MSDA_REC", 1) != 0) {
    if (ref)
      hipLaunchKernelGGL((msda_fwd_rec_kernel<LPH, true>), dim3((unsigned)nblocks), dim3(256), smem_rec, stream, value, shapes, starts, loc,
                         attn, ref, npairs, S, M, L, Lq, P, out, ldo, ldl, L * P + 1, loc_out, attn_out);
    else
      hipLaunchKernelGGL((msda_fwd_rec_kernel<LPH, false>), dim3((unsigned)nblocks), dim3(256), smem_rec, stream, value, shapes, starts, loc,
                         attn, ref, npairs, S, M, L, Lq, P, out, ldo, ldl, L * P + 1);
    return mss_launch_status();
  }
  if (loc_out) return MSS_ERR_UNSUPPORTED;              // only the record kernel hands its locations / weights back
  if (ref) { if (buf) MSDA_LAUNCH(true, true); else MSDA_LAUNCH(true, false); }
  else { if (buf) MSDA_LAUNCH(false, true); else MSDA_LAUNCH(false, false); }
#undef MSDA_LAUNCH
  return mss_launch_status();
}

int msda_forward_window(const float* value, const int64_t* starts, const float* loc, const float* attn, const float* ref,
                        const int64_t* host_shapes, int N, int S, int M, int L, int Lq, int P, float* out, hipStream_t stream) {
  MsdaLevels lv;
  lv.L = L;
  long long sum = 0;
  int tiles = 0;
  for (int l = 0; l < L; ++l) {
    const long long h = host_shapes[2 * l], w = host_shapes[2 * l + 1];
    if (h <= 0 || w <= 0 || h > 32767 || w > 32767) return MSS_ERR_UNSUPPORTED;
    lv.H[l] = (int)h, lv.W[l] = (int)w;
    lv.qstart[l] = (int)sum;
    lv.tiles_x[l] = (int)((w + 7) / 8);
    lv.tile_start[l] = tiles;
    tiles += (int)(((h + 7) / 8) * ((w + 7) / 8));
    sum += h * w;
  }
  lv.tile_start[L] = tiles;
  for (int l = L; l < MSDA_WIN_MAXL; ++l) lv.H[l] = lv.W[l] = lv.qstart[l] = lv.tiles_x[l] = 0, lv.tile_start[l + 1] = tiles;
  const int grid_mode = sum == (long long)Lq;          // the queries are the pixels of the levels (encoder self-attention)
  const int ntiles = grid_mode ? tiles : (Lq + 63) / 64;
  const long long nblocks = (long long)N * ntiles * M;
  if (nblocks > 0x7fffffffll) return MSS_ERR_UNSUPPORTED;
  const size_t smem = ((size_t)MSDA_WIN_ROWS * 32 + (size_t)3 * 64 * L * P) * sizeof(float);
  if (ref)
    hipLaunchKernelGGL(msda_fwd_window_kernel<true>, dim3((unsigned)nblocks), dim3(256), smem, stream, value, starts, loc, attn, ref,
                       lv, grid_mode, ntiles, S, M, Lq, P, out);
  else
    hipLaunchKernelGGL(msda_fwd_window_kernel<false>, dim3((unsigned)nblocks), dim3(256), smem, stream, value, starts, loc, attn, ref,
                       lv, grid_mode, ntiles, S, M, Lq, P, out);
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
  const int lds_path = MSS_ENV_INT("MSS_MSDA_BWD_LDS", 1);      // 0: atomic kernel only, 2: owner-computes path at any size (tests)
  // owner-computes path: fp32, D = 32, at most 2^22 samples per (image, head, level) (fixed-point headroom)
  const bool owner = lds_path && sizeof(T) == 4 && D == 32 && npairs > 0 && M <= 65535 && N <= 65535 &&
                     (long long)Lq * P <= (1ll << 22) &&
                     (lds_path == 2 || npairs * L * P >= (6ll << 20));   // below ~6 M samples the atomic kernel wins
  if (N > 0 && !owner) {
    hipError_t e = hipMemsetAsync(gvalue, 0, (size_t)N * S * M * D * sizeof(T), stream);
    if (e != hipSuccess) return (int)e;
  }
  if (npairs == 0) return MSS_OK;
  if (!gout || !gloc || !gattn) return MSS_ERR_BAD_ARG;
  if (owner) {
    // grad_value: tiles accumulated in LDS. The first 8 bytes of grad_loc hold max|grad_out|, max|attn| until the
    // gather pass below overwrites them with the real gradient.
    unsigned* absmax = reinterpret_cast<unsigned*>(gloc);
    hipError_t e = hipMemsetAsync(absmax, 0, 2 * sizeof(unsigned), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(msda_absmax_kernel, dim3(1024), dim3(256), 0, stream, reinterpret_cast<const float*>(gout),
                       npairs * D, reinterpret_cast<const float*>(attn), npairs * L * P, absmax);
    // an upper bound on the tile count of any set of level shapes with S positions in all (checked exhaustively on
    // random shape sets up to 600 x 600); surplus workgroups exit at once
    const unsigned tiles_bound = (unsigned)(S / 32 + 4 * L + 4);
    // 4 sample pairs in flight per wave (2 and 8 measured within 2 % / 12 % slower)
    hipLaunchKernelGGL(msda_bwd_value_lds_kernel<4>, dim3(tiles_bound, (unsigned)M, (unsigned)N), dim3(MSDA_LDS_NT), 0, stream, shapes,
                       starts, reinterpret_cast<const float*>(loc), reinterpret_cast<const float*>(attn),
                       reinterpret_cast<const float*>(gout), absmax, S, M, L, Lq, P, reinterpret_cast<float*>(gvalue));
    // grad_loc / grad_attn: the gather pass (no atomics)
    const size_t smem = (size_t)4 * 8 * L * P * 4 * sizeof(float);
    const bool aligned = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(gout)) & 15) == 0;
    if (aligned && smem <= 65536) {
      const long long nblocks = (npairs + 31) / 32;
      const bool buf = (unsigned long long)N * S * M * D * 4ull < 0xffffffffull && MSS_ENV_INT("MSS_MSDA_BUF", 1) != 0;
      if (buf)
        hipLaunchKernelGGL(msda_bwd_gather_fast_kernel<true>, dim3((unsigned)nblocks), dim3(256), smem, stream,
                           reinterpret_cast<const float*>(value), shapes, starts, reinterpret_cast<const float*>(loc),
                           reinterpret_cast<const float*>(attn), reinterpret_cast<const float*>(gout), npairs, S, M, L, Lq, P,
                           reinterpret_cast<float*>(gloc), reinterpret_cast<float*>(gattn));
      else
        hipLaunchKernelGGL(msda_bwd_gather_fast_kernel<false>, dim3((unsigned)nblocks), dim3(256), smem, stream,
                           reinterpret_cast<const float*>(value), shapes, starts, reinterpret_cast<const float*>(loc),
                           reinterpret_cast<const float*>(attn), reinterpret_cast<const float*>(gout), npairs, S, M, L, Lq, P,
                           reinterpret_cast<float*>(gloc), reinterpret_cast<float*>(gattn));
    } else {
      const long long nblocks = (npairs + 7) / 8;
      hipLaunchKernelGGL((msda_bwd_kernel<T, 32, false>), dim3((unsigned)nblocks), dim3(256), 0, stream, value, shapes,
                         starts, loc, attn, gout, npairs, S, M, D, L, Lq, P, gvalue, gloc, gattn);
    }
    return mss_launch_status();
  }
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
  const bool buf = (unsigned long long)N * S * M * 32 * 4ull < 0xffffffffull && MSS_ENV_INT("MSS_MSDA_BUF", 1) != 0;
  if (buf)
    hipLaunchKernelGGL(msda_bwd_gather_fast_kernel<true>, dim3((unsigned)nblocks), dim3(256), smem_gather, stream, value, shapes, starts,
                       loc, attn, gout, npairs, S, M, L, Lq, P, gloc, gattn, goff, ldo, glog, ldl);
  else
    hipLaunchKernelGGL(msda_bwd_gather_fast_kernel<false>, dim3((unsigned)nblocks), dim3(256), smem_gather, stream, value, shapes, starts,
                       loc, attn, gout, npairs, S, M, L, Lq, P, gloc, gattn, goff, ldo, glog, ldl);
  return mss_launch_status();
}

// workspace of the cell-sorted backward: [header 64 B: ticket][counts][offsets + 1][cursor][block sums] | records | lower parts
struct MsdaRowWs { size_t counts, offsets, cursor, bsum, records, gbot, total; long long nkeys; int nblk; };
static bool msda_row_layout(const MsdaRowGeom& g, int N, int M, int L, int Lq, int P, MsdaRowWs& w) {
  const long long per_image = (long long)Lq * M * L * P;
  if (per_image >= (1ll << 31) || (long long)N * per_image >= (1ll << 31)) return false;
  if (P > 16 || Lq >= (1 << 27)) return false;                       // record header: query << 4 | point
  w.nkeys = (long long)N * M * g.KS;
  if (w.nkeys >= (1ll << 30) || (long long)g.rows * N * M >= (1ll << 30)) return false;
  w.nblk = (int)((w.nkeys + 4095) / 4096);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  w.counts = 64;
  w.offsets = up(w.counts + (size_t)w.nkeys * 4);
  w.cursor = up(w.offsets + (size_t)(w.nkeys + 1) * 4);
  w.bsum = up(w.cursor + (size_t)w.nkeys * 4);
  w.records = up(w.bsum + (size_t)w.nblk * 4);
  w.gbot = up(w.records + (size_t)N * per_image * 16);
  w.total = up(w.gbot + (size_t)N * g.cells * M * 32 * 4);
  return true;
}

// MSS_ERR_UNSUPPORTED: shapes this path does not take (the caller goes on to the binned path)
static int msda_backward_rows(const float* value, const int64_t* shapes, const int64_t* starts, const int64_t* host_shapes,
                              const float* loc, const float* attn, const float* gout, int N, int S, int M, int L, int Lq, int P,
                              float* gvalue, float* gloc, float* gattn, void* ws, size_t ws_bytes, hipStream_t stream, float* goff,
                              long long ldo, float* glog, long long ldl) {
  MsdaRowGeom g;
  MsdaRowWs w;
  if (!msda_row_geom(host_shapes, L, Lq, P, g) || !msda_row_layout(g, N, M, L, Lq, P, w)) return MSS_ERR_UNSUPPORTED;
  if (ws_bytes < w.total || (reinterpret_cast<uintptr_t>(gvalue) & 15)) return MSS_ERR_UNSUPPORTED;
  long long cells = 0;
  for (int l = 0; l < L; ++l) cells += host_shapes[2 * l] * host_shapes[2 * l + 1];
  if (cells < S) {          // rows of the value tensor outside the L levels come back zero (ms_deform_attn_cuda.cu:126)
    hipError_t ez = hipMemsetAsync(gvalue, 0, (size_t)N * S * M * 32 * sizeof(float), stream);
    if (ez != hipSuccess) return (int)ez;
  }
  char* base = static_cast<char*>(ws);
  int* ticket = reinterpret_cast<int*>(base);
  int* counts = reinterpret_cast<int*>(base + w.counts);
  int* offsets = reinterpret_cast<int*>(base + w.offsets);
  int* cursor = reinterpret_cast<int*>(base + w.cursor);
  int* bsum = reinterpret_cast<int*>(base + w.bsum);
  f32x4* records = reinterpret_cast<f32x4*>(base + w.records);
  float* gbot = reinterpret_cast<float*>(base + w.gbot);
  hipError_t e = hipMemsetAsync(base, 0, w.counts + (size_t)w.nkeys * 4, stream);
  if (e != hipSuccess) return (int)e;
  const long long per_image = (long long)Lq * M * L * P;
  const unsigned chunks = (unsigned)((per_image + 1023) / 1024);
  hipLaunchKernelGGL(msda_cell_kernel<0>, dim3(chunks, (unsigned)N), dim3(256), 0, stream, g, loc, attn, M, Lq, P, per_image, counts, records);
  hipLaunchKernelGGL(msda_scan_sums_kernel, dim3((unsigned)w.nblk), dim3(1024), 0, stream, counts, (int)w.nkeys, bsum);
  hipLaunchKernelGGL(msda_scan_apply_kernel, dim3((unsigned)w.nblk), dim3(1024), 0, stream, counts, (int)w.nkeys, bsum, offsets, cursor);
  hipLaunchKernelGGL(msda_cell_kernel<1>, dim3(chunks, (unsigned)N), dim3(256), 0, stream, g, loc, attn, M, Lq, P, per_image, cursor, records);
  const long long nitems = (long long)g.rows * N * M;
  const long long want = (nitems + 3) / 4;
  const unsigned nwg = (unsigned)(want < 256 * 8 ? want : 256 * 8);        // up to eight 4-wave workgroups per CU, a row per wave
  hipLaunchKernelGGL(msda_bwd_value_rows_kernel<8>, dim3(nwg), dim3(256), 0, stream, g, starts, gout, records, offsets, ticket, S, M, Lq, N,
                     gvalue, gbot);
  hipLaunchKernelGGL(msda_rows_merge_kernel, dim3(128, (unsigned)N, (unsigned)L), dim3(256), 0, stream, g, starts, S, M, N, gbot, gvalue);
  return msda_launch_gather(value, shapes, starts, loc, attn, gout, N, S, M, L, Lq, P, gloc, gattn, goff, ldo, glog, ldl, stream);
}

// workspace of the binned backward: [header 64 B: ticket, max|grad_out|, max|attn|][counts][offsets + 1][cursor] | records | halo
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
  // round 5: the cell-sorted path, opt-in (MSS_MSDA_BWD_ROWS=1; tests and A/B): parity-green and bit-reproducible, but slower than
  // the binned path at every size measured -- 2.71 ms against 1.35 at C4 N = 16 (profiles/r05/msda_bwd_rows.md)
  if (MSS_ENV_INT("MSS_MSDA_BWD_ROWS", 0) != 0) {
    rc = msda_backward_rows(value, shapes, starts, host_shapes, loc, attn, gout, N, S, M, L, Lq, P, gvalue, gloc, gattn, ws, ws_bytes, stream,
                            goff, ldo, glog, ldl);
    if (rc != MSS_ERR_UNSUPPORTED) return rc;
  }
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
  const size_t smem_per_lp = (size_t)4 * 3 * sizeof(float) * L * P;
  if (aligned && D == 32 && smem_per_lp * 8 <= 65536)
    return msda_forward_fast<8>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, N, S, M, L,
                                Lq, P, out, s);
  if (aligned && D == 16 && smem_per_lp * 16 <= 65536)
    return msda_forward_fast<4>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, N, S, M, L,
                                Lq, P, out, s);
  if (aligned && D == 64 && smem_per_lp * 4 <= 65536)
    return msda_forward_fast<16>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, N, S, M, L,
                                 Lq, P, out, s);
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
  const size_t smem_per_lp = (size_t)4 * 3 * sizeof(float) * L * P;
  if (!aligned) return MSS_ERR_UNSUPPORTED;
  if (D == 32 && smem_per_lp * 8 <= 65536)
    return msda_forward_fast<8>(value, spatial_shapes, level_start_index, offsets, logits, reference_points, N, S, M, L, Lq, P, out, s,
                                ld_offsets, ld_logits, sampling_loc_out, attn_weight_out);
  if (D == 16 && smem_per_lp * 16 <= 65536)
    return msda_forward_fast<4>(value, spatial_shapes, level_start_index, offsets, logits, reference_points, N, S, M, L, Lq, P, out, s,
                                ld_offsets, ld_logits, sampling_loc_out, attn_weight_out);
  if (D == 64 && smem_per_lp * 4 <= 65536)
    return msda_forward_fast<16>(value, spatial_shapes, level_start_index, offsets, logits, reference_points, N, S, M, L, Lq, P, out, s,
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

// forward through LDS windows (msda_fwd_window_kernel): `host_shapes` is a HOST copy of spatial_shapes [L][2]; reference_points
// NULL = sampling_loc / attn_weight given (the op), else raw offsets / logits (the fused form). fp32, D = 32, L <= 8,
// L*P <= 20, 16-byte aligned value / out; MSS_ERR_UNSUPPORTED otherwise.
int mss_msda_forward_window_f32(const float* value, const int64_t* host_shapes, const int64_t* level_start_index,
                                const float* loc_or_offsets, const float* attn_or_logits, const float* reference_points, int N,
                                int S, int M, int D, int L, int Lq, int P, float* out, void* stream) {
  if (!host_shapes) return MSS_ERR_BAD_ARG;
  int rc = msda_check(value, host_shapes, level_start_index, loc_or_offsets, attn_or_logits, N, S, M, D, L, Lq, P);
  if (rc) return rc;
  if ((long long)N * Lq * M == 0) return MSS_OK;
  if (!out) return MSS_ERR_BAD_ARG;
  const bool aligned = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (!aligned || D != 32 || L > MSDA_WIN_MAXL || L * P > MSDA_MAX_LP) return MSS_ERR_UNSUPPORTED;
  return msda_forward_window(value, level_start_index, loc_or_offsets, attn_or_logits, reference_points, host_shapes, N, S, M, L,
                             Lq, P, out, static_cast<hipStream_t>(stream));
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
  MsdaRowGeom gr;
  MsdaRowWs wr;
  if (MSS_ENV_INT("MSS_MSDA_BWD_ROWS", 0) != 0 && msda_row_geom(host_shapes, L, Lq, P, gr) && msda_row_layout(gr, N, M, L, Lq, P, wr) &&
      wr.total > w.total)
    return (long long)wr.total;
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
