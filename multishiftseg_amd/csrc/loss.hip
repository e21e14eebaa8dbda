// Fused RelContrastiveLoss for MI355X: forward value and both gradients (dlogit, dscore) in a
// handful of streaming passes instead of the ~25 ATen kernels + topk + host randperm of the
// reference (lib/loss.py:34-156).
//
//   pass1   one read of logit/score/target: per-pixel logsumexp + CE, pixel kind, partial sums, the augmented
//           half's CE values for the selection, AND the gradient of every pixel whose weight is already known:
//           the original half always, the augmented half when no selection is asked (loss.py:46-69,90-96)
//   select  exact k-th smallest CE by 4 x 8-bit radix histograms on the fp32 bit pattern
//           (replaces torch.topk over ~4 M values, loss.py:98-102)
//   pass2   augmented half only (selection mode): softmax-minus-onehot gradient of the selected pixels (the logits of
//           the ~20 % rejected pixels are not re-read), zeros elsewhere, target mutation (loss.py:103-111)
//   compact ordered stream compaction of the three score sets              (loss.py:122-124)
//   pairs   hinge terms over permuted pairs + scatter of dscore             (loss.py:129-137)
//   cin     in-distribution consistency term                                (loss.py:139-145)
// HBM traffic per pixel with selection: 76 B logits once (+76 B again for the selected augmented pixels: 0.4 x 76 on
// average), dlogit 76 B written once, target/score/kind/lse/ce side data ~25 B: ~210 B against the 168 B of SURVEY 8(d),
// which assumes the gradient weights are known up front (they are not: k = int(0.8 * #in-distribution augmented pixels)
// and the threshold CE exist only after every augmented logit has been seen).
// The fast kernels take 4 consecutive pixels per lane (16-B loads on every class plane, all 19 in flight, the 19x4
// logits stay in registers between the max, the sum and the gradient); H*W % 4 != 0 or C != 19 use the scalar ones.
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

enum { CNT_SUM_CE_ORIG = 0, CNT_N_IN_ORIG, CNT_N_IN_AUG, CNT_N_OOD, CNT_SUM_CIN, CNT_N_SAME, CNT_SUM_CE_AUG_ALL,
       CNT_SUM_SEL, CNT_N_SEL, CNT_SUM_CORIG, CNT_SUM_CAUG, CNT_N_PAIRS, CNT_BAD_TARGET };

typedef long long i64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <int NV>
__device__ __forceinline__ void block_add(const float (&v)[NV], double* dst, const int* slots) {
  __shared__ float red[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float s = mss_wave_sum(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (s != 0.f) atomicAdd(&dst[slots[threadIdx.x]], (double)s);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void rcl_pass1_kernel(MssRclArgs a, float* __restrict__ lse_out,
                                                        float* __restrict__ ce_aug, uint8_t* __restrict__ kind_out,
                                                        double* __restrict__ counters, float* __restrict__ dlogit) {
  const long long HW = (long long)a.H * a.W;
  const long long total = (long long)a.B * HW;
  const int h = a.B / 2;
  const long long half = (long long)h * HW;
  const float g_orig = a.w_ce_orig / (float)half, g_aug = a.w_ce_aug / (float)half;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const long long p = i - (long long)b * HW;
    const int64_t t = a.target[i];
    bool in = t < 99;                            // in_id  (loss.py:31,47)
    const bool ood = t > 99 && t != 255;         // void_id (loss.py:32,46)
    if (in && (t < 0 || t >= a.C)) { in = false; acc[7] += 1.f; }   // nll_loss raises on such a label: flagged, never indexed
    const float* lp = a.logit + (long long)b * a.C * HW + p;
    float m = -__builtin_huge_valf();
    for (int c = 0; c < a.C; ++c) m = fmaxf(m, lp[(long long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s += expf(lp[(long long)c * HW] - m);
    const float lse = m + logf(s);
    lse_out[i] = lse;
    float ce = 0.f;
    if (in) ce = lse - lp[(long long)t * HW];
    kind_out[i] = in ? 1 : (ood ? 2 : 0);
    if (b < h) {
      acc[0] += ce;
      acc[1] += in ? 1.f : 0.f;
      // pair (i, i + half): in-distribution consistency term (loss.py:141-145)
      const int64_t t2 = a.target[i + half];
      if (in && t2 < 99) {
        acc[4] += fmaxf(a.score[i + half] - a.score[i] - a.m2, 0.f);
        acc[5] += 1.f;
      }
    } else {
      ce_aug[i - half] = in ? ce : __builtin_huge_valf();
      acc[2] += in ? 1.f : 0.f;
      acc[6] += ce;
    }
    acc[3] += ood ? 1.f : 0.f;
    if (dlogit && (b < h || !a.select)) {        // gradient weight known without the selection: written here, once
      float* dp = dlogit + (long long)b * a.C * HW + p;
      const float g = in ? (b < h ? g_orig : g_aug) : 0.f;
      for (int c = 0; c < a.C; ++c)
        dp[(long long)c * HW] = g != 0.f ? g * (expf(lp[(long long)c * HW] - lse) - (c == (int)t ? 1.f : 0.f)) : 0.f;
    }
  }
  __shared__ int slots[8];
  if (threadIdx.x == 0) {
    slots[0] = CNT_SUM_CE_ORIG; slots[1] = CNT_N_IN_ORIG; slots[2] = CNT_N_IN_AUG; slots[3] = CNT_N_OOD;
    slots[4] = CNT_SUM_CIN; slots[5] = CNT_N_SAME; slots[6] = CNT_SUM_CE_AUG_ALL; slots[7] = CNT_BAD_TARGET;
  }
  __syncthreads();
  block_add<8>(acc, counters, slots);
}

// 4 consecutive pixels of one image plane per lane (H*W % 4 == 0), C classes compile-time.
template <int C>
__global__ __launch_bounds__(256) void rcl_pass1_v4_kernel(MssRclArgs a, float* __restrict__ lse_out,
                                                           float* __restrict__ ce_aug, uint8_t* __restrict__ kind_out,
                                                           double* __restrict__ counters, float* __restrict__ dlogit) {
  const long long HW = (long long)a.H * a.W;
  const long long total4 = (long long)a.B * HW / 4;
  const int h = a.B / 2;
  const long long half = (long long)h * HW;
  const float g_orig = a.w_ce_orig / (float)half, g_aug = a.w_ce_aug / (float)half;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long g4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; g4 < total4;
       g4 += (long long)gridDim.x * blockDim.x) {
    const long long i = g4 * 4;
    const int b = (int)(i / HW);
    const long long p = i - (long long)b * HW;
    // uniform plane base (SGPR pair) + one 32-bit per-lane byte offset shared by all C loads / stores (the launcher
    // guarantees B*C*H*W*4 < 2^32): 19 independent 64-bit per-lane addresses would cost 76 more registers
    const unsigned voff = (unsigned)(((long long)b * C * HW + p) * 4);
    const char* lbase = reinterpret_cast<const char*>(a.logit);
    f32x4 v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const f32x4*>(lbase + (size_t)c * HW * 4 + voff);
    const i64x2 t01 = *reinterpret_cast<const i64x2*>(a.target + i), t23 = *reinterpret_cast<const i64x2*>(a.target + i + 2);
    const long long t[4] = {t01.x, t01.y, t23.x, t23.y};
    int tt[4];
    uint32_t kinds = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bool in = t[e] < 99;                                           // in_id  (loss.py:31,47)
      const bool ood = t[e] > 99 && t[e] != 255;                     // void_id (loss.py:32,46)
      if (in && (t[e] < 0 || t[e] >= C)) { in = false; acc[7] += 1.f; }   // nll_loss raises: flagged, never indexed
      tt[e] = in ? (int)t[e] : -1;
      kinds |= (uint32_t)(in ? 1 : (ood ? 2 : 0)) << (8 * e);
      acc[3] += ood ? 1.f : 0.f;
    }
    f32x4 m = v[0], xt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      m.x = fmaxf(m.x, v[c].x); m.y = fmaxf(m.y, v[c].y); m.z = fmaxf(m.z, v[c].z); m.w = fmaxf(m.w, v[c].w);
#pragma unroll
      for (int e = 0; e < 4; ++e) xt[e] = tt[e] == c ? v[c][e] : xt[e];     // raw logit of the target class
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      v[c].x = expf(v[c].x - m.x); v[c].y = expf(v[c].y - m.y); v[c].z = expf(v[c].z - m.z); v[c].w = expf(v[c].w - m.w);
      sum += v[c];
    }
    f32x4 lse, ce, gsc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lse[e] = m[e] + logf(sum[e]);
      ce[e] = tt[e] >= 0 ? lse[e] - xt[e] : 0.f;
      gsc[e] = 0.f;
    }
    *reinterpret_cast<uint32_t*>(kind_out + i) = kinds;
    if (b < h) {
      const i64x2 u01 = *reinterpret_cast<const i64x2*>(a.target + i + half), u23 = *reinterpret_cast<const i64x2*>(a.target + i + half + 2);
      const long long u[4] = {u01.x, u01.y, u23.x, u23.y};
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.score + i), s1 = *reinterpret_cast<const f32x4*>(a.score + i + half);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] += ce[e];
        acc[1] += tt[e] >= 0 ? 1.f : 0.f;
        if (tt[e] >= 0 && u[e] < 99) {          // pair (i, i + half): in-distribution consistency term (loss.py:141-145)
          acc[4] += fmaxf(s1[e] - s0[e] - a.m2, 0.f);
          acc[5] += 1.f;
        }
        gsc[e] = tt[e] >= 0 ? g_orig : 0.f;
      }
    } else {
      f32x4 cev;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cev[e] = tt[e] >= 0 ? ce[e] : __builtin_huge_valf();
        acc[2] += tt[e] >= 0 ? 1.f : 0.f;
        acc[6] += ce[e];
        gsc[e] = tt[e] >= 0 ? g_aug : 0.f;
      }
      *reinterpret_cast<f32x4*>(ce_aug + (i - half)) = cev;
      *reinterpret_cast<f32x4*>(lse_out + i) = lse;      // pass 2 only ever reads the augmented half's
    }
    if (dlogit && (b < h || !a.select)) {
      char* dbase = reinterpret_cast<char*>(dlogit);
      f32x4 inv;
#pragma unroll
      for (int e = 0; e < 4; ++e) inv[e] = gsc[e] / sum[e];          // g * softmax_c = g * exp(x_c - m) / sum
#pragma unroll
      for (int c = 0; c < C; ++c) {
        f32x4 o = v[c] * inv;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (tt[e] == c) o[e] -= gsc[e];
        *reinterpret_cast<f32x4*>(dbase + (size_t)c * HW * 4 + voff) = o;
      }
    }
  }
  __shared__ int slots[8];
  if (threadIdx.x == 0) {
    slots[0] = CNT_SUM_CE_ORIG; slots[1] = CNT_N_IN_ORIG; slots[2] = CNT_N_IN_AUG; slots[3] = CNT_N_OOD;
    slots[4] = CNT_SUM_CIN; slots[5] = CNT_N_SAME; slots[6] = CNT_SUM_CE_AUG_ALL; slots[7] = CNT_BAD_TARGET;
  }
  __syncthreads();
  block_add<8>(acc, counters, slots);
}

// ---- radix select --------------------------------------------------------------------------
// sel[0] prefix key, sel[1] n_less (elements with key < final threshold), sel[2] k, sel[3] need_equal,
// sel[4] tie ticket counter, sel[5] k_remaining (internal)
__global__ void rcl_select_init_kernel(const double* __restrict__ counters, float ratio, uint32_t* sel,
                                       uint32_t* hist) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float n_in = (float)counters[CNT_N_IN_AUG];
    const uint32_t k = (uint32_t)(int)(ratio * n_in);  // int(selection_ratio * total_num), float32 as in torch
    sel[0] = 0; sel[1] = 0; sel[2] = k; sel[3] = 0; sel[4] = 0; sel[5] = k; sel[6] = 0;
  }
  for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
}

__global__ __launch_bounds__(256) void rcl_hist_kernel(const float* __restrict__ v, long long n, const uint32_t* sel,
                                                       int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = sel[0];
  const uint32_t mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
  if (sel[2] != 0) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
      const uint32_t key = f2key(v[i]);
      if ((key & mask) == (prefix & mask)) atomicAdd(&lh[(key >> shift) & 255], 1u);
    }
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// Histogram of one radix digit for the local (single-process) selection. WAVE_AGG: the first pass looks at the sign +
// exponent byte, which takes ~6 distinct values for CE values -- every lane of a wave hits the same few LDS words, so a
// wave adds ONE count per distinct digit instead of 64 serialised atomics (41 us -> a few us on 2.1 M values).
// (A single-launch variant whose last workgroup also picked the digit was measured at 65 us per pass: the release
// fence + returning ticket atomic of 1024 workgroups cost more than the second launch.)
template <bool WAVE_AGG>
__global__ __launch_bounds__(256) void rcl_hist2_kernel(const float* __restrict__ v, long long n, const uint32_t* sel,
                                                        int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = sel[0];
  const uint32_t mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
  if (sel[2] != 0) {
    const long long step = (long long)gridDim.x * blockDim.x;
    const long long n_round = (n + step - 1) / step * step;     // whole waves stay converged through the ballot loop
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += step) {
      uint32_t key = 0;
      bool take = false;
      if (i < n) { key = f2key(v[i]); take = (key & mask) == (prefix & mask); }
      const uint32_t digit = (key >> shift) & 255u;
      if (WAVE_AGG) {
        unsigned long long todo = __ballot(take);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t d0 = (uint32_t)__shfl((int)digit, leader);
          const unsigned long long same = __ballot(take && digit == d0) & todo;
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[d0], (uint32_t)__popcll(same));
          todo &= ~same;
        }
      } else if (take) {
        atomicAdd(&lh[digit], 1u);
      }
    }
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// digit d with excl(d) < krem <= incl(d), by a 256-lane prefix sum (the serial scan of rcl_pick_kernel took 8 us)
__global__ __launch_bounds__(256) void rcl_pick_par_kernel(uint32_t* sel, uint32_t* hist, int shift) {
  __shared__ uint32_t scan[256];
  const uint32_t cnt = hist[threadIdx.x];
  scan[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t t = threadIdx.x >= (unsigned)o ? scan[threadIdx.x - o] : 0u;
    __syncthreads();
    scan[threadIdx.x] += t;
    __syncthreads();
  }
  if (sel[2] != 0) {
    const uint32_t krem = sel[5], incl = scan[threadIdx.x], excl = incl - cnt;
    const bool mine = excl < krem && krem <= incl;
    const bool fallback = threadIdx.x == 255 && krem > incl;     // cannot happen for a consistent k; mirrors the serial pick
    if (mine || fallback) {
      sel[0] |= (uint32_t)threadIdx.x << shift;
      sel[1] += excl;
      sel[5] = krem - excl;
      if (shift == 0) sel[3] = krem - excl;     // how many elements equal to the threshold are taken
    }
  }
  hist[threadIdx.x] = 0;
}

__global__ void rcl_pick_kernel(uint32_t* sel, uint32_t* hist, int shift) {
  if (threadIdx.x == 0 && sel[2] != 0) {
    uint32_t krem = sel[5];  // 1-based rank of the wanted element inside the current bucket
    uint32_t less = sel[1];
    uint32_t d = 0;
    for (; d < 256; ++d) {
      const uint32_t c = hist[d];
      if (krem <= c) break;
      krem -= c;
      less += c;
    }
    if (d > 255) d = 255;
    sel[0] |= d << shift;
    sel[1] = less;
    sel[5] = krem;
    if (shift == 0) sel[3] = krem;  // how many elements equal to the threshold are taken
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
}

// ---- the selection in 5 launches instead of 9 (round 5): the pick of digit d+1 rides in front of the histogram of digit d ----
// Every workgroup of the histogram pass of byte `shift` first repeats the pick of the byte above it from the finished
// histogram of that byte (a 256-lane scan: ~1 us, against a launch of its own), so the chain is hist(24) hist(16) hist(8)
// hist(0) pick(0). The selection state travels through two alternating 8-word buffers (workgroup 0 writes the state AFTER
// its pick for the next launch; the others may still be reading the state before it), each pass has its own histogram.
// FIRST: no pick, the state is the initial one (k from the pass-1 counters, as rcl_select_init_kernel).
struct RclSelState { uint32_t prefix, less, k, krem; };
__device__ __forceinline__ RclSelState rcl_pick_digit(const uint32_t* __restrict__ sel_in, const uint32_t* __restrict__ hist_prev,
                                                      int shift_prev, uint32_t* scan, uint32_t* st) {
  const uint32_t cnt = hist_prev[threadIdx.x];
  scan[threadIdx.x] = cnt;
  if (threadIdx.x == 0) { st[0] = sel_in[0]; st[1] = sel_in[1]; st[2] = sel_in[2]; st[3] = sel_in[5]; }
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t t = threadIdx.x >= (unsigned)o ? scan[threadIdx.x - o] : 0u;
    __syncthreads();
    scan[threadIdx.x] += t;
    __syncthreads();
  }
  const uint32_t k = st[2], krem = st[3], prefix = st[0], less = st[1];
  __syncthreads();
  if (k != 0) {
    const uint32_t incl = scan[threadIdx.x], excl = incl - cnt;
    const bool mine = excl < krem && krem <= incl;
    const bool fallback = threadIdx.x == 255 && krem > incl;     // cannot happen for a consistent k; mirrors rcl_pick_par_kernel
    if (mine || fallback) { st[0] = prefix | ((uint32_t)threadIdx.x << shift_prev); st[1] = less + excl; st[3] = krem - excl; }
  }
  __syncthreads();
  RclSelState r = {st[0], st[1], st[2], st[3]};
  return r;
}

template <bool WAVE_AGG, bool FIRST>
__global__ __launch_bounds__(256) void rcl_hist_pick_kernel(const float* __restrict__ v, long long n, const double* __restrict__ counters,
                                                            float ratio, const uint32_t* __restrict__ sel_in, uint32_t* __restrict__ sel_out,
                                                            const uint32_t* __restrict__ hist_prev, uint32_t* __restrict__ hist_cur, int shift) {
  __shared__ uint32_t lh[256], scan[256], st[4];
  lh[threadIdx.x] = 0;
  RclSelState s;
  if (FIRST) {
    const float n_in = (float)counters[CNT_N_IN_AUG];
    const uint32_t k = (uint32_t)(int)(ratio * n_in);            // int(selection_ratio * total_num), float32 as in torch
    s.prefix = 0; s.less = 0; s.k = k; s.krem = k;
    __syncthreads();
  } else {
    s = rcl_pick_digit(sel_in, hist_prev, shift + 8, scan, st);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sel_out[0] = s.prefix; sel_out[1] = s.less; sel_out[2] = s.k; sel_out[3] = 0; sel_out[4] = 0; sel_out[5] = s.krem; sel_out[6] = 0;
  }
  const uint32_t prefix = s.prefix;
  const uint32_t mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
  if (s.k != 0) {
    const long long step = (long long)gridDim.x * blockDim.x;
    const long long n_round = (n + step - 1) / step * step;     // whole waves stay converged through the ballot loop
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += step) {
      uint32_t key = 0;
      bool take = false;
      if (i < n) { key = f2key(v[i]); take = (key & mask) == (prefix & mask); }
      const uint32_t digit = (key >> shift) & 255u;
      if (WAVE_AGG) {
        unsigned long long todo = __ballot(take);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t d0 = (uint32_t)__shfl((int)digit, leader);
          const unsigned long long same = __ballot(take && digit == d0) & todo;
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[d0], (uint32_t)__popcll(same));
          todo &= ~same;
        }
      } else if (take) {
        atomicAdd(&lh[digit], 1u);
      }
    }
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist_cur[threadIdx.x], lh[threadIdx.x]);
}

// the last pick (byte 0) -> the canonical selection words (sel[3] = how many elements equal to the threshold are taken)
__global__ __launch_bounds__(256) void rcl_pick_final_kernel(const uint32_t* __restrict__ sel_in, const uint32_t* __restrict__ hist_prev,
                                                             uint32_t* __restrict__ sel) {
  __shared__ uint32_t scan[256], st[4];
  const RclSelState s = rcl_pick_digit(sel_in, hist_prev, 0, scan, st);
  if (threadIdx.x == 0) {
    sel[0] = s.prefix; sel[1] = s.less; sel[2] = s.k; sel[3] = s.k != 0 ? s.krem : 0u; sel[4] = 0; sel[5] = s.krem; sel[6] = 0;
  }
}

// ---- pass 2 ---------------------------------------------------------------------------------
// Selection mode only, augmented half only: which pixels made the easiest-k cut (key < threshold, plus `need_eq` of the
// ones equal to it, first come first served), their gradient, the target mutation of all the others.
__global__ __launch_bounds__(256) void rcl_pass2_kernel(MssRclArgs a, const float* __restrict__ lse,
                                                        const float* __restrict__ ce_aug,
                                                        const uint8_t* __restrict__ kind, uint32_t* sel,
                                                        double* __restrict__ counters, float grad_scale,
                                                        float* __restrict__ dlogit) {
  const long long HW = (long long)a.H * a.W;
  const long long total = (long long)a.B * HW;
  const long long half = (long long)(a.B / 2) * HW;
  const uint32_t thr = sel[0], k = sel[2], need_eq = sel[3];
  const float g_aug = k ? grad_scale * a.w_ce_aug / (float)k : 0.f;
  float acc[2] = {0.f, 0.f};
  for (long long i = half + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const long long p = i - (long long)b * HW;
    bool chosen = false;
    if (kind[i] == 1 && k) {
      const float ce = ce_aug[i - half];
      const uint32_t key = f2key(ce);
      if (key < thr) chosen = true;
      else if (key == thr) chosen = atomicAdd(&sel[4], 1u) < need_eq;
      if (chosen) { acc[0] += ce; acc[1] += 1.f; }
    }
    int t = -1;
    if (chosen) t = (int)a.target[i];
    else a.target[i] = 255;                      // loss.py:110-111,115 (every non-selected pixel)
    if (dlogit) {
      float* dp = dlogit + (long long)b * a.C * HW + p;
      if (chosen) {
        const float* lp = a.logit + (long long)b * a.C * HW + p;
        const float l = lse[i];
        for (int c = 0; c < a.C; ++c) dp[(long long)c * HW] = g_aug * (expf(lp[(long long)c * HW] - l) - (c == t ? 1.f : 0.f));
      } else {
        for (int c = 0; c < a.C; ++c) dp[(long long)c * HW] = 0.f;
      }
    }
  }
  __shared__ int slots[2];
  if (threadIdx.x == 0) { slots[0] = CNT_SUM_SEL; slots[1] = CNT_N_SEL; }
  __syncthreads();
  block_add<2>(acc, counters, slots);
}

template <int C>
__global__ __launch_bounds__(256) void rcl_pass2_v4_kernel(MssRclArgs a, const float* __restrict__ lse,
                                                           const float* __restrict__ ce_aug,
                                                           const uint8_t* __restrict__ kind, uint32_t* sel,
                                                           double* __restrict__ counters, float grad_scale,
                                                           float* __restrict__ dlogit) {
  const long long HW = (long long)a.H * a.W;
  const long long half = (long long)(a.B / 2) * HW;
  const long long n4 = ((long long)a.B * HW - half) / 4;
  const uint32_t thr = sel[0], k = sel[2], need_eq = sel[3];
  const float g_aug = k ? grad_scale * a.w_ce_aug / (float)k : 0.f;
  float acc[2] = {0.f, 0.f};
  for (long long g4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; g4 < n4;
       g4 += (long long)gridDim.x * blockDim.x) {
    const long long i = half + g4 * 4;
    const int b = (int)(i / HW);
    const long long p = i - (long long)b * HW;
    const f32x4 ce = *reinterpret_cast<const f32x4*>(ce_aug + (i - half));
    const uint32_t kinds = *reinterpret_cast<const uint32_t*>(kind + i);
    bool chosen[4];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      chosen[e] = false;
      if (((kinds >> (8 * e)) & 255u) == 1u && k) {
        const uint32_t key = f2key(ce[e]);
        if (key < thr) chosen[e] = true;
        else if (key == thr) chosen[e] = atomicAdd(&sel[4], 1u) < need_eq;
        if (chosen[e]) { acc[0] += ce[e]; acc[1] += 1.f; }
      }
      any |= chosen[e];
    }
    int t[4] = {-1, -1, -1, -1};
    if (any) {
      const i64x2 t01 = *reinterpret_cast<const i64x2*>(a.target + i), t23 = *reinterpret_cast<const i64x2*>(a.target + i + 2);
      const long long tv[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = chosen[e] ? (int)tv[e] : -1;
      if (!(chosen[0] && chosen[1] && chosen[2] && chosen[3])) {
        i64x2 n01 = t01, n23 = t23;
        if (!chosen[0]) n01.x = 255;
        if (!chosen[1]) n01.y = 255;
        if (!chosen[2]) n23.x = 255;
        if (!chosen[3]) n23.y = 255;
        *reinterpret_cast<i64x2*>(a.target + i) = n01;
        *reinterpret_cast<i64x2*>(a.target + i + 2) = n23;
      }
    } else {
      const i64x2 v255 = {255, 255};             // loss.py:110-111,115 (every non-selected pixel)
      *reinterpret_cast<i64x2*>(a.target + i) = v255;
      *reinterpret_cast<i64x2*>(a.target + i + 2) = v255;
    }
    if (dlogit) {
      const unsigned voff = (unsigned)(((long long)b * C * HW + p) * 4);
      char* dbase = reinterpret_cast<char*>(dlogit);
      if (any) {
        const char* lbase = reinterpret_cast<const char*>(a.logit);
        f32x4 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const f32x4*>(lbase + (size_t)c * HW * 4 + voff);
        const f32x4 l = *reinterpret_cast<const f32x4*>(lse + i);
        f32x4 gv;
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = chosen[e] ? g_aug : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = gv[e] * (expf(v[c][e] - l[e]) - (t[e] == c ? 1.f : 0.f));
          *reinterpret_cast<f32x4*>(dbase + (size_t)c * HW * 4 + voff) = o;
        }
      } else {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(dbase + (size_t)c * HW * 4 + voff) = z;
      }
    }
  }
  __shared__ int slots[2];
  if (threadIdx.x == 0) { slots[0] = CNT_SUM_SEL; slots[1] = CNT_N_SEL; }
  __syncthreads();
  block_add<2>(acc, counters, slots);
}

// ---- ordered compaction ------------------------------------------------------------------------
constexpr int CB = 1024;  // pixels per compaction block (4 per thread)

__device__ __forceinline__ int kind_class(uint8_t kind, bool first_half) {
  if (kind == 2) return 2;
  if (kind == 1) return first_half ? 0 : 1;
  return -1;
}

__global__ __launch_bounds__(256) void rcl_count_kernel(const uint8_t* __restrict__ kind, long long total,
                                                        long long half, uint32_t* __restrict__ block_counts,
                                                        int nblocks) {
  const long long base = (long long)blockIdx.x * CB;
  float c[3] = {0.f, 0.f, 0.f};
  for (int e = 0; e < 4; ++e) {
    const long long i = base + threadIdx.x * 4 + e;
    if (i < total) {
      const int k = kind_class(kind[i], i < half);
      if (k >= 0) c[k] += 1.f;
    }
  }
  __shared__ float red[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    float s = mss_wave_sum(c[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    block_counts[threadIdx.x * nblocks + blockIdx.x] =
        (uint32_t)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// exclusive scan of each of the 3 rows of block_counts (single workgroup), totals to n_out
__global__ __launch_bounds__(1024) void rcl_scan_kernel(uint32_t* block_counts, int nblocks, uint32_t* n_out) {
  // one workgroup per list (r05: the three lists used to be scanned one after the other by ONE workgroup, 20 barriers each: 12 us);
  // partial sums per thread, inclusive scan inside each wave with shuffles, then over the 16 wave totals
  __shared__ uint32_t wave_tot[16];
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* row = block_counts + (size_t)k * nblocks;
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
  uint32_t s = 0;
  for (int i = b0; i < b1; ++i) s += row[i];
  uint32_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before += wave_tot[w];
  uint32_t run = before + incl - s;
  for (int i = b0; i < b1; ++i) { const uint32_t c = row[i]; row[i] = run; run += c; }
  if (threadIdx.x == 1023) n_out[k] = before + incl;
}

__global__ __launch_bounds__(256) void rcl_scatter_kernel(const uint8_t* __restrict__ kind, long long total,
                                                          long long half, const uint32_t* __restrict__ block_offs,
                                                          int nblocks, int32_t* __restrict__ idx0,
                                                          int32_t* __restrict__ idx1, int32_t* __restrict__ idx2) {
  const long long base = (long long)blockIdx.x * CB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int kc[4];
  int cnt[3] = {0, 0, 0};
  for (int e = 0; e < 4; ++e) {
    const long long i = base + threadIdx.x * 4 + e;
    kc[e] = i < total ? kind_class(kind[i], i < half) : -1;
    if (kc[e] >= 0) cnt[kc[e]]++;
  }
  // exclusive prefix of cnt[k] over the 256 threads (thread order == pixel order)
  __shared__ int wsum[4][3];
  int pre[3];
  for (int k = 0; k < 3; ++k) {
    int v = cnt[k];
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    pre[k] = inc - v;
    if (lane == 63) wsum[wave][k] = inc;
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w][k];
    pre[k] += off + (int)block_offs[(size_t)k * nblocks + blockIdx.x];
  }
  int32_t* outs[3] = {idx0, idx1, idx2};
  for (int e = 0; e < 4; ++e) {
    if (kc[e] >= 0) {
      outs[kc[e]][pre[kc[e]]] = (int32_t)(base + threadIdx.x * 4 + e);
      pre[kc[e]]++;
    }
  }
}

// ---- pair terms -------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// keyed bijection of [0, n): balanced Feistel network on 2*hb bits + cycle walking
__device__ __forceinline__ uint32_t feistel_perm(uint32_t i, uint32_t n, uint32_t seed) {
  int bits = 32 - __clz(n > 1 ? n - 1 : 1);
  if (bits < 2) bits = 2;
  const int hb = (bits + 1) >> 1;
  const uint32_t hm = (1u << hb) - 1;
  uint32_t x = i;
  do {
    uint32_t l = x >> hb, r = x & hm;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const uint32_t f = mix32(r ^ (seed + 0x9e3779b9u * (rd + 1))) & hm;
      const uint32_t nl = r;
      r = l ^ f;
      l = nl;
    }
    x = (l << hb) | r;
  } while (x >= n);
  return x;
}

template <bool FEISTEL>
__global__ __launch_bounds__(256) void rcl_pairs_kernel(const float* __restrict__ score,
                                                        const int32_t* __restrict__ idx_a,
                                                        const int64_t* __restrict__ perm_a,
                                                        const int32_t* __restrict__ idx_o,
                                                        const int64_t* __restrict__ perm_o, long long n_host,
                                                        const uint32_t* __restrict__ n_out, int set_a,
                                                        long long max_samples, uint32_t seed_a, uint32_t seed_o,
                                                        float margin, double* __restrict__ counters, int slot,
                                                        float grad_w, float* __restrict__ dscore) {
  long long n = n_host;
  uint32_t na = 0, no = 0;
  if (FEISTEL) {
    na = n_out[set_a]; no = n_out[2];
    n = max_samples;
    if (n > (long long)n_out[0]) n = n_out[0];
    if (n > (long long)n_out[1]) n = n_out[1];
    if (n > (long long)n_out[2]) n = n_out[2];
  }
  const float coef = n > 0 ? grad_w / (float)n : 0.f;
  float acc[1] = {0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    long long ja, jo;
    if (FEISTEL) { ja = feistel_perm((uint32_t)i, na, seed_a); jo = feistel_perm((uint32_t)i, no, seed_o); }
    else { ja = perm_a[i]; jo = perm_o[i]; }
    const int32_t pa = idx_a[ja], po = idx_o[jo];
    const float v = score[pa] + margin - score[po];
    if (v > 0.f) {
      acc[0] += v;
      if (dscore) { atomicAdd(&dscore[pa], coef); atomicAdd(&dscore[po], -coef); }
    }
  }
  __shared__ int slots[1];
  if (threadIdx.x == 0) slots[0] = slot;
  __syncthreads();
  block_add<1>(acc, counters, slots);
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[CNT_N_PAIRS] = (double)n;
}

// r04: both hinge terms of the device-pairing mode in ONE launch and WITHOUT atomics. The two launches of rcl_pairs_kernel<true>
// were 2 x 34 us of the 0.42 ms loss at 2 x 19 x 1024 x 2048, bound by ~4e5 scattered float atomics into dscore. Pair i of BOTH
// terms meets the same OOD element (the OOD permutation has one seed), the three permutations are bijections, and the three sets
// are disjoint -- so thread i is the ONLY writer of dscore[orig_i], dscore[aug_i] and dscore[ood_i]: plain read-modify-writes, in
// the order the two launches applied them (orig term, then aug term), hence the same bits, and no order dependence at all.
__global__ __launch_bounds__(256) void rcl_pairs2_kernel(const float* __restrict__ score, const int32_t* __restrict__ idx_orig,
                                                         const int32_t* __restrict__ idx_aug, const int32_t* __restrict__ idx_ood,
                                                         const uint32_t* __restrict__ n_out, long long max_samples, uint32_t seed0,
                                                         uint32_t seed1, uint32_t seed_o, float margin0, float margin1,
                                                         double* __restrict__ counters, float grad_w, float* __restrict__ dscore) {
  long long n = max_samples;
  if (n > (long long)n_out[0]) n = n_out[0];
  if (n > (long long)n_out[1]) n = n_out[1];
  if (n > (long long)n_out[2]) n = n_out[2];
  if ((long long)blockIdx.x * blockDim.x >= n && blockIdx.x != 0) return;          // workgroup-uniform
  const uint32_t n0 = n_out[0], n1 = n_out[1], no = n_out[2];
  const float coef = n > 0 ? grad_w / (float)n : 0.f;
  float acc[2] = {0.f, 0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const uint32_t j0 = feistel_perm((uint32_t)i, n0, seed0), j1 = feistel_perm((uint32_t)i, n1, seed1);
    const uint32_t jo = feistel_perm((uint32_t)i, no, seed_o);
    const int32_t p0 = idx_orig[j0], p1 = idx_aug[j1], po = idx_ood[jo];
    const float so = score[po];
    const float v0 = score[p0] + margin0 - so, v1 = score[p1] + margin1 - so;
    if (v0 > 0.f) acc[0] += v0;
    if (v1 > 0.f) acc[1] += v1;
    if (dscore) {
      if (v0 > 0.f) dscore[p0] += coef;
      if (v1 > 0.f) dscore[p1] += coef;
      if (v0 > 0.f || v1 > 0.f) {
        float t = dscore[po];
        if (v0 > 0.f) t += -coef;
        if (v1 > 0.f) t += -coef;
        dscore[po] = t;
      }
    }
  }
  __shared__ int slots[2];
  if (threadIdx.x == 0) { slots[0] = CNT_SUM_CORIG; slots[1] = CNT_SUM_CAUG; }
  __syncthreads();
  block_add<2>(acc, counters, slots);
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[CNT_N_PAIRS] = (double)n;
}

// ---- data-parallel ("global") pairing ------------------------------------------------------------
// The three score sets are the rank-major concatenation of every rank's compaction. Pair i couples
// element perm_a(i) of the global set A with element perm_o(i) of the global OOD set (keyed Feistel
// bijections shared by all ranks). Every rank walks all n pairs and keeps those whose A element it
// owns; the OOD score comes from the all-gathered [W][cap] vector and its gradient goes into a
// [W][cap] buffer that the host all-reduces and scatters back into the owner's dscore.
__global__ __launch_bounds__(256) void rcl_pairs_global_kernel(
    const float* __restrict__ score, const int32_t* __restrict__ idx_a, uint32_t a_off, uint32_t a_cnt_local,
    uint32_t a_cnt_global, const float* __restrict__ ood_all, const uint32_t* __restrict__ ood_off, int W, uint32_t cap,
    uint32_t n_pairs, uint32_t seed_a, uint32_t seed_o, float margin, double* __restrict__ counters, int slot,
    float coef, float* __restrict__ dscore, float* __restrict__ g_ood) {
  const uint32_t n_ood = ood_off[W];
  float acc[1] = {0.f};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += gridDim.x * blockDim.x) {
    const uint32_t ja = feistel_perm(i, a_cnt_global, seed_a);
    if (ja < a_off || ja >= a_off + a_cnt_local) continue;
    const uint32_t jo = feistel_perm(i, n_ood, seed_o);
    int r = 0;
    while (r + 1 < W && jo >= ood_off[r + 1]) ++r;
    const size_t slot_o = (size_t)r * cap + (jo - ood_off[r]);
    const int32_t pa = idx_a[ja - a_off];
    const float v = score[pa] + margin - ood_all[slot_o];
    if (v > 0.f) {
      acc[0] += v;
      if (dscore) { atomicAdd(&dscore[pa], coef); atomicAdd(&g_ood[slot_o], -coef); }
    }
  }
  __shared__ int slots[1];
  if (threadIdx.x == 0) slots[0] = slot;
  __syncthreads();
  block_add<1>(acc, counters, slots);
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[CNT_N_PAIRS] = (double)n_pairs;
}

__global__ void rcl_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, uint32_t n,
                                  float* __restrict__ dst) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void rcl_scatter_add_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx, uint32_t n,
                                       float* __restrict__ dst) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[idx[i]] += g[i];
}

// dscore of the consistency term; ASSIGNS every element (so it doubles as the zero fill)
__global__ void rcl_cin_bwd_kernel(MssRclArgs a, const uint8_t* __restrict__ kind,
                                   const double* __restrict__ counters, float grad_w, float* __restrict__ dscore) {
  const long long HW = (long long)a.H * a.W;
  const long long half = (long long)(a.B / 2) * HW;
  const double ns = counters[CNT_N_SAME];
  const float coef = ns > 0 ? grad_w / (float)ns : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < half;
       i += (long long)gridDim.x * blockDim.x) {
    float g = 0.f;
    if (kind[i] == 1 && kind[i + half] == 1 && a.score[i + half] - a.score[i] - a.m2 > 0.f) g = coef;
    dscore[i] = -g;
    dscore[i + half] = g;
  }
  // odd batch sizes leave a tail that belongs to no pair
  const long long total = (long long)a.B * HW;
  for (long long i = 2 * half + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x)
    dscore[i] = 0.f;
}

// loss = w0*ce_orig + w1*ce_aug + wc*(c_orig + c_aug + c_in)   (loss.py:73-88,147)
__global__ void rcl_finalize_kernel(MssRclArgs a, const double* __restrict__ counters, const uint32_t* sel,
                                    float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  const double half = (double)(a.B / 2) * a.H * a.W;
  const float ce_orig = (float)(counters[CNT_SUM_CE_ORIG] / half);
  float ce_aug;
  if (a.select) {
    const uint32_t k = sel[2];
    ce_aug = k ? (float)(counters[CNT_SUM_SEL] / (double)k) : 0.f;
  } else {
    ce_aug = (float)(counters[CNT_SUM_CE_AUG_ALL] / half);
  }
  const double np = counters[CNT_N_PAIRS];
  const float c_orig = (float)(counters[CNT_SUM_CORIG] / np);   // 0/0 -> NaN, as mean() of an empty tensor
  const float c_aug = (float)(counters[CNT_SUM_CAUG] / np);
  const float c_in = (float)(counters[CNT_SUM_CIN] / counters[CNT_N_SAME]);
  out[0] = (a.w_ce_orig * ce_orig + a.w_ce_aug * ce_aug) + a.w_contras * ((c_orig + c_aug) + c_in);
  out[1] = ce_orig; out[2] = ce_aug; out[3] = c_orig; out[4] = c_aug; out[5] = c_in;
  // labels in [C, 99) or below 0: F.nll_loss raises on them (loss.py:59); here the loss turns NaN and the count is reported
  out[6] = (float)counters[CNT_BAD_TARGET];
  if (counters[CNT_BAD_TARGET] > 0) out[0] = __builtin_nanf("");
}

inline int grid_for(long long work_items, int cap = 256 * 16) {
  long long b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int rcl_check(const MssRclArgs* a) {
  if (!a || !a->logit || !a->score || !a->target) return MSS_ERR_BAD_ARG;
  if (a->B < 2 || (a->B & 1) || a->C < 1 || a->H < 1 || a->W < 1) return MSS_ERR_BAD_ARG;  // [orig...; aug...] pairs
  if ((long long)a->B * a->H * a->W >= (1ll << 31)) return MSS_ERR_UNSUPPORTED;
  return MSS_OK;
}

}  // namespace

#define S_(x) static_cast<hipStream_t>(x)

extern "C" {

int mss_rcl_num_compact_blocks(int B, int H, int W) { return (int)(((long long)B * H * W + CB - 1) / CB); }

static bool rcl_vec4(const MssRclArgs* a) {
  return a->C == 19 && ((long long)a->H * a->W) % 4 == 0 && (long long)a->B * a->C * a->H * a->W * 4 < (1ll << 32) &&
         ((reinterpret_cast<uintptr_t>(a->logit) | reinterpret_cast<uintptr_t>(a->score) | reinterpret_cast<uintptr_t>(a->target)) & 15) == 0;
}

// dlogit (optional, [B][C][H][W]): pass 1 writes the gradient of every pixel whose weight does not depend on the
// selection -- the original half always, the augmented half when a->select == 0 (then pass 2 is not needed at all).
static int rcl_pass1_impl(const MssRclArgs* a, float* lse, float* ce_aug, uint8_t* kind, double* counters, float* dlogit,
                          void* stream, bool clear_counters);
int mss_rcl_pass1_f32(const MssRclArgs* a, float* lse, float* ce_aug, uint8_t* kind, double* counters, float* dlogit,
                      void* stream) {
  return rcl_pass1_impl(a, lse, ce_aug, kind, counters, dlogit, stream, true);
}
static int rcl_pass1_impl(const MssRclArgs* a, float* lse, float* ce_aug, uint8_t* kind, double* counters, float* dlogit,
                          void* stream, bool clear_counters) {
  int rc = rcl_check(a);
  if (rc) return rc;
  if (!lse || !ce_aug || !kind || !counters) return MSS_ERR_BAD_ARG;
  if (clear_counters) {
    hipError_t e = hipMemsetAsync(counters, 0, 16 * sizeof(double), S_(stream));
    if (e != hipSuccess) return (int)e;
  }
  const long long total = (long long)a->B * a->H * a->W;
  const bool v4 = rcl_vec4(a) && ((reinterpret_cast<uintptr_t>(lse) | reinterpret_cast<uintptr_t>(ce_aug) |
                                   reinterpret_cast<uintptr_t>(kind) | reinterpret_cast<uintptr_t>(dlogit)) & 15) == 0;
  if (v4)
    // 768 = 3 resident workgroups per CU (147 registers): every workgroup ends with 8 double atomics on the same 8 counters,
    // and same-address atomics retire at ~80 per microsecond -- 8192 workgroups spent more time there than streaming
    hipLaunchKernelGGL(rcl_pass1_v4_kernel<19>, dim3(grid_for(total / 4, 768)), dim3(256), 0, S_(stream), *a, lse,
                       ce_aug, kind, counters, dlogit);
  else
    hipLaunchKernelGGL(rcl_pass1_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), *a, lse, ce_aug, kind,
                       counters, dlogit);
  return mss_launch_status();
}

int mss_rcl_select_f32(const float* ce_aug, long long n, const double* counters, float selection_ratio,
                       uint32_t* hist_ws, uint32_t* sel, void* stream) {
  if (!ce_aug || !counters || !hist_ws || !sel || n <= 0) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_select_init_kernel, dim3(1), dim3(256), 0, S_(stream), counters, selection_ratio, sel,
                     hist_ws);
  const dim3 grid(grid_for(n, 1024));
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (shift == 24) hipLaunchKernelGGL(rcl_hist2_kernel<true>, grid, dim3(256), 0, S_(stream), ce_aug, n, sel, shift, hist_ws);
    else hipLaunchKernelGGL(rcl_hist2_kernel<false>, grid, dim3(256), 0, S_(stream), ce_aug, n, sel, shift, hist_ws);
    hipLaunchKernelGGL(rcl_pick_par_kernel, dim3(1), dim3(256), 0, S_(stream), sel, hist_ws, shift);
  }
  return mss_launch_status();
}

// The same selection in 5 launches (the picks ride in front of the next byte's histogram, see rcl_hist_pick_kernel). scratch:
// MSS_RCL_SELECT_SCRATCH_WORDS 32-bit words (four histograms + two state buffers), cleared here unless the caller says it already
// is (scratch_zeroed != 0: mss_rcl_loss_device_f32 clears it together with the counters).
int mss_rcl_select_merged_f32(const float* ce_aug, long long n, const double* counters, float selection_ratio, uint32_t* scratch,
                              int scratch_zeroed, uint32_t* sel, void* stream) {
  if (!ce_aug || !counters || !scratch || !sel || n <= 0) return MSS_ERR_BAD_ARG;
  if (!scratch_zeroed) {
    hipError_t e = hipMemsetAsync(scratch, 0, MSS_RCL_SELECT_SCRATCH_WORDS * sizeof(uint32_t), S_(stream));
    if (e != hipSuccess) return (int)e;
  }
  uint32_t* hist = scratch;                     // [4][256]: bytes 3, 2, 1, 0
  uint32_t* sb = scratch + 4 * 256;             // [2][8]
  const dim3 grid(grid_for(n, 1024));
  hipLaunchKernelGGL((rcl_hist_pick_kernel<true, true>), grid, dim3(256), 0, S_(stream), ce_aug, n, counters, selection_ratio, sb, sb,
                     hist, hist, 24);
  hipLaunchKernelGGL((rcl_hist_pick_kernel<false, false>), grid, dim3(256), 0, S_(stream), ce_aug, n, counters, selection_ratio, sb, sb + 8,
                     hist, hist + 256, 16);
  hipLaunchKernelGGL((rcl_hist_pick_kernel<false, false>), grid, dim3(256), 0, S_(stream), ce_aug, n, counters, selection_ratio, sb + 8, sb,
                     hist + 256, hist + 512, 8);
  hipLaunchKernelGGL((rcl_hist_pick_kernel<false, false>), grid, dim3(256), 0, S_(stream), ce_aug, n, counters, selection_ratio, sb, sb + 8,
                     hist + 512, hist + 768, 0);
  hipLaunchKernelGGL(rcl_pick_final_kernel, dim3(1), dim3(256), 0, S_(stream), sb + 8, hist + 768, sel);
  return mss_launch_status();
}

// The same selection, one radix pass at a time, so a data-parallel caller can all-reduce the 256-bin
// histogram between hist and pick (4 all-reduces of 1 KB give the exact global k-th smallest).
int mss_rcl_select_init_f32(const double* counters, float selection_ratio, uint32_t* hist_ws, uint32_t* sel,
                            void* stream) {
  if (!counters || !hist_ws || !sel) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_select_init_kernel, dim3(1), dim3(256), 0, S_(stream), counters, selection_ratio, sel,
                     hist_ws);
  return mss_launch_status();
}
int mss_rcl_select_hist_f32(const float* ce_aug, long long n, const uint32_t* sel, int shift, uint32_t* hist_ws,
                            void* stream) {
  if (!ce_aug || !sel || !hist_ws || n <= 0 || shift < 0 || shift > 24 || shift % 8) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_hist_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, S_(stream), ce_aug, n, sel, shift,
                     hist_ws);
  return mss_launch_status();
}
int mss_rcl_select_pick_f32(uint32_t* sel, uint32_t* hist_ws, int shift, void* stream) {
  if (!sel || !hist_ws || shift < 0 || shift > 24 || shift % 8) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_pick_kernel, dim3(1), dim3(256), 0, S_(stream), sel, hist_ws, shift);
  return mss_launch_status();
}

int mss_rcl_pairs_global_f32(const float* score, const int32_t* idx_a, uint32_t a_off, uint32_t a_cnt_local,
                             uint32_t a_cnt_global, const float* ood_all, const uint32_t* ood_off, int W,
                             uint32_t cap, uint32_t n_pairs, uint32_t seed_a, uint32_t seed_o, float margin,
                             double* counters, int slot, float coef, float* dscore, float* g_ood, void* stream) {
  if (!score || !idx_a || !ood_all || !ood_off || !counters || W < 1) return MSS_ERR_BAD_ARG;
  if (slot != 0 && slot != 1) return MSS_ERR_BAD_ARG;
  if (dscore && !g_ood) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_pairs_global_kernel, dim3(grid_for(n_pairs > 0 ? n_pairs : 1, 2048)), dim3(256), 0,
                     S_(stream), score, idx_a, a_off, a_cnt_local, a_cnt_global, ood_all, ood_off, W, cap, n_pairs,
                     seed_a, seed_o, margin, counters, slot == 0 ? CNT_SUM_CORIG : CNT_SUM_CAUG, coef, dscore, g_ood);
  return mss_launch_status();
}

int mss_rcl_gather_f32(const float* src, const int32_t* idx, uint32_t n, float* dst, void* stream) {
  if (n == 0) return MSS_OK;
  if (!src || !idx || !dst) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_gather_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, S_(stream), src, idx, n, dst);
  return mss_launch_status();
}
int mss_rcl_scatter_add_f32(const float* g, const int32_t* idx, uint32_t n, float* dst, void* stream) {
  if (n == 0) return MSS_OK;
  if (!g || !idx || !dst) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_scatter_add_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, S_(stream), g, idx, n, dst);
  return mss_launch_status();
}

int mss_rcl_pass2_f32(const MssRclArgs* a, const float* lse, const float* ce_aug, const uint8_t* kind,
                      uint32_t* sel, double* counters, float grad_scale, float* dlogit, void* stream) {
  int rc = rcl_check(a);
  if (rc) return rc;
  if (!lse || !ce_aug || !kind || !sel || !counters) return MSS_ERR_BAD_ARG;
  if (grad_scale != 1.f) return MSS_ERR_BAD_ARG;   // pass 1 writes its half of dlogit unscaled: one scale for both halves
  if (!a->select) return MSS_OK;       // nothing left to do: pass 1 wrote the whole gradient
  const long long n_aug = (long long)(a->B - a->B / 2) * a->H * a->W;
  const bool v4 = rcl_vec4(a) && ((reinterpret_cast<uintptr_t>(lse) | reinterpret_cast<uintptr_t>(ce_aug) |
                                   reinterpret_cast<uintptr_t>(kind) | reinterpret_cast<uintptr_t>(dlogit)) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(rcl_pass2_v4_kernel<19>, dim3(grid_for(n_aug / 4, 1024)), dim3(256), 0, S_(stream), *a, lse,
                       ce_aug, kind, sel, counters, grad_scale, dlogit);
  else
    hipLaunchKernelGGL(rcl_pass2_kernel, dim3(grid_for(n_aug)), dim3(256), 0, S_(stream), *a, lse, ce_aug, kind, sel,
                       counters, grad_scale, dlogit);
  return mss_launch_status();
}

int mss_rcl_compact_f32(const uint8_t* kind, int B, int H, int W, int32_t* idx_orig, int32_t* idx_aug,
                        int32_t* idx_ood, uint32_t* block_counts, uint32_t* n_out, void* stream) {
  if (!kind || !idx_orig || !idx_aug || !idx_ood || !block_counts || !n_out) return MSS_ERR_BAD_ARG;
  const long long total = (long long)B * H * W;
  const long long half = (long long)(B / 2) * H * W;
  const int nb = mss_rcl_num_compact_blocks(B, H, W);
  hipLaunchKernelGGL(rcl_count_kernel, dim3(nb), dim3(256), 0, S_(stream), kind, total, half, block_counts, nb);
  hipLaunchKernelGGL(rcl_scan_kernel, dim3(3), dim3(1024), 0, S_(stream), block_counts, nb, n_out);
  hipLaunchKernelGGL(rcl_scatter_kernel, dim3(nb), dim3(256), 0, S_(stream), kind, total, half, block_counts, nb,
                     idx_orig, idx_aug, idx_ood);
  return mss_launch_status();
}

int mss_rcl_pairs_f32(const float* score, const int32_t* idx_a, const int64_t* perm_a, const int32_t* idx_o,
                      const int64_t* perm_o, long long n, float margin, double* counters, int slot, float grad_w,
                      float* dscore, void* stream) {
  if (!score || !idx_a || !idx_o || !counters || n < 0) return MSS_ERR_BAD_ARG;
  if (n > 0 && (!perm_a || !perm_o)) return MSS_ERR_BAD_ARG;
  if (slot != 0 && slot != 1) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_pairs_kernel<false>, dim3(grid_for(n > 0 ? n : 1, 2048)), dim3(256), 0, S_(stream), score,
                     idx_a, perm_a, idx_o, perm_o, n, nullptr, 0, 0ll, 0u, 0u, margin, counters,
                     slot == 0 ? CNT_SUM_CORIG : CNT_SUM_CAUG, grad_w, dscore);
  return mss_launch_status();
}

int mss_rcl_pairs_device_f32(const float* score, const int32_t* idx_a, const int32_t* idx_o, const uint32_t* n_out,
                             int set_a, long long max_samples, uint32_t seed_a, uint32_t seed_o, float margin,
                             double* counters, int slot, float grad_w, float* dscore, void* stream) {
  if (!score || !idx_a || !idx_o || !n_out || !counters) return MSS_ERR_BAD_ARG;
  if ((slot != 0 && slot != 1) || (set_a != 0 && set_a != 1)) return MSS_ERR_BAD_ARG;
  // 256 workgroups: the pair count is only known on the device, and each workgroup ends with one same-address atomic
  hipLaunchKernelGGL(rcl_pairs_kernel<true>, dim3(256), dim3(256), 0, S_(stream), score, idx_a, nullptr, idx_o,
                     nullptr, 0ll, n_out, set_a, max_samples, seed_a, seed_o, margin, counters,
                     slot == 0 ? CNT_SUM_CORIG : CNT_SUM_CAUG, grad_w, dscore);
  return mss_launch_status();
}

int mss_rcl_pairs_device2_f32(const float* score, const int32_t* idx_orig, const int32_t* idx_aug, const int32_t* idx_ood,
                              const uint32_t* n_out, long long max_samples, uint32_t seed_orig, uint32_t seed_aug, uint32_t seed_ood,
                              float margin_orig, float margin_aug, double* counters, float grad_w, float* dscore, void* stream) {
  if (!score || !idx_orig || !idx_aug || !idx_ood || !n_out || !counters || max_samples < 0) return MSS_ERR_BAD_ARG;
  // one thread per pair up to the host-side bound of the pair count (every set is a subset of the batch's pixels); the
  // workgroups beyond the device-side count return immediately
  long long blocks = (max_samples + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;       // each live workgroup ends with one same-address atomic on its sum
  hipLaunchKernelGGL(rcl_pairs2_kernel, dim3((unsigned)blocks), dim3(256), 0, S_(stream), score, idx_orig, idx_aug, idx_ood, n_out,
                     max_samples, seed_orig, seed_aug, seed_ood, margin_orig, margin_aug, counters, grad_w, dscore);
  return mss_launch_status();
}

// ---- the whole loss in ONE call (device pairing, one process) ------------------------------------------------------------------
// The sequence above is ~17 dependent launches of 4-120 us; issued one by one through the Python binding (~10 us of host time
// each) the standalone loss was bound by the HOST (0.42 ms at 2 x 19 x 1024 x 2048 against ~0.33 ms of kernel time). One entry
// point carves every intermediate out of one workspace and issues the launches back to back.
struct RclWs { size_t lse, ce_aug, kind, counters, sel, hist, idx, block_counts, n_out, total; };
static RclWs rcl_ws_layout(int B, int H, int W) {
  const size_t total = (size_t)B * H * W, half = (size_t)(B / 2) * H * W;
  const size_t nb = (size_t)mss_rcl_num_compact_blocks(B, H, W);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  RclWs w;
  w.lse = 0;
  w.ce_aug = up(w.lse + total * 4);
  w.kind = up(w.ce_aug + half * 4);
  w.counters = up(w.kind + total);
  w.sel = up(w.counters + 16 * 8);
  w.hist = up(w.sel + 8 * 4);
  w.idx = up(w.hist + MSS_RCL_SELECT_SCRATCH_WORDS * 4);     // the merged selection's scratch (four histograms + two state buffers)
  w.block_counts = up(w.idx + 3 * total * 4);
  w.n_out = up(w.block_counts + 3 * nb * 4);
  w.total = up(w.n_out + 4 * 4);
  return w;
}

long long mss_rcl_workspace_bytes(int B, int H, int W) {
  if (B < 2 || B % 2 || H <= 0 || W <= 0) return -1;
  return (long long)rcl_ws_layout(B, H, W).total;
}

int mss_rcl_loss_device_f32(const MssRclArgs* a, void* workspace, long long workspace_bytes, long long max_samples, uint32_t seed,
                            float* dlogit, float* dscore, float* out, void* stream) {
  int rc = rcl_check(a);
  if (rc) return rc;
  if (!workspace || !out || (reinterpret_cast<uintptr_t>(workspace) & 255)) return MSS_ERR_BAD_ARG;
  const RclWs w = rcl_ws_layout(a->B, a->H, a->W);
  if (workspace_bytes < (long long)w.total) return MSS_ERR_BAD_ARG;
  char* base = static_cast<char*>(workspace);
  float* lse = reinterpret_cast<float*>(base + w.lse);
  float* ce_aug = reinterpret_cast<float*>(base + w.ce_aug);
  uint8_t* kind = reinterpret_cast<uint8_t*>(base + w.kind);
  double* counters = reinterpret_cast<double*>(base + w.counters);
  uint32_t* sel = reinterpret_cast<uint32_t*>(base + w.sel);
  uint32_t* hist = reinterpret_cast<uint32_t*>(base + w.hist);
  const size_t total = (size_t)a->B * a->H * a->W;
  int32_t* idx0 = reinterpret_cast<int32_t*>(base + w.idx);
  int32_t* idx1 = idx0 + total;
  int32_t* idx2 = idx1 + total;
  uint32_t* block_counts = reinterpret_cast<uint32_t*>(base + w.block_counts);
  uint32_t* n_out = reinterpret_cast<uint32_t*>(base + w.n_out);
  // counters, selection words and the selection's scratch lie next to each other: ONE clear for all of them
  {
    hipError_t e = hipMemsetAsync(counters, 0, (w.hist - w.counters) + MSS_RCL_SELECT_SCRATCH_WORDS * sizeof(uint32_t), S_(stream));
    if (e != hipSuccess) return (int)e;
  }
  if ((rc = rcl_pass1_impl(a, lse, ce_aug, kind, counters, dlogit, stream, false))) return rc;
  const bool select = a->select != 0;
  if (select) {
    rc = mss_rcl_select_merged_f32(ce_aug, (long long)(a->B / 2) * a->H * a->W, counters, a->selection_ratio, hist, 1, sel, stream);
    if (rc) return rc;
    if ((rc = mss_rcl_pass2_f32(a, lse, ce_aug, kind, sel, counters, 1.0f, dlogit, stream))) return rc;
  } else {
    hipError_t e = hipMemsetAsync(sel, 0, 8 * sizeof(uint32_t), S_(stream));
    if (e != hipSuccess) return (int)e;
  }
  if ((rc = mss_rcl_compact_f32(kind, a->B, a->H, a->W, idx0, idx1, idx2, block_counts, n_out, stream))) return rc;
  if (dscore && (rc = mss_rcl_cin_bwd_f32(a, kind, counters, a->w_contras, dscore, stream))) return rc;
  const uint32_t s0 = seed * 0x9E3779B1u;
  if ((rc = mss_rcl_pairs_device2_f32(a->score, idx0, idx1, idx2, n_out, max_samples, s0 + 1, s0 + 2, s0 + 7, a->m0, a->m1, counters,
                                      a->w_contras, dscore, stream)))
    return rc;
  return mss_rcl_finalize_f32(a, counters, sel, out, stream);
}

int mss_rcl_cin_bwd_f32(const MssRclArgs* a, const uint8_t* kind, const double* counters, float grad_w,
                        float* dscore, void* stream) {
  int rc = rcl_check(a);
  if (rc) return rc;
  if (!kind || !counters || !dscore) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_cin_bwd_kernel, dim3(grid_for((long long)(a->B / 2) * a->H * a->W)), dim3(256), 0,
                     S_(stream), *a, kind, counters, grad_w, dscore);
  return mss_launch_status();
}

int mss_rcl_finalize_f32(const MssRclArgs* a, const double* counters, const uint32_t* sel, float* out,
                         void* stream) {
  int rc = rcl_check(a);
  if (rc) return rc;
  if (!counters || !sel || !out) return MSS_ERR_BAD_ARG;
  hipLaunchKernelGGL(rcl_finalize_kernel, dim3(1), dim3(64), 0, S_(stream), *a, counters, sel, out);
  return mss_launch_status();
}

}  // extern "C"
