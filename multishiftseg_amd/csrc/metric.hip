// Pixel-level OOD metrics on the device: AUROC, average precision (AUPRC) and FPR at 95 % recall over all
// labelled pixels of an evaluation sweep -- SURVEY 8(f)-1. Replaces eval_ood_measure / get_measures /
// fpr_and_fdr_at_recall (lib/utils/metric.py:87-127,130-153,170-180), which copy every full-resolution score
// map to the host, concatenate in NumPy and let sklearn argsort 1e7-1e8 pixels on one core.
//
// Exact, not binned: the positive (OOD, label == id_out) and negative (label == id_in) scores are compacted into
// two arrays of order-preserving 32-bit keys, each array is radix-sorted (rocPRIM device primitive, keys only),
// and every metric is a function of rank counts found by binary search:
//   AUROC = sum_pos (#neg < s + 1/2 #neg == s) / (P N)              -- the area under sklearn's ROC trapezoids,
//                                                                       accumulated in 64-bit integers (exact)
//   AP    = 1/P sum_pos  #pos>=s / (#pos>=s + #neg>=s)              -- sklearn's sum_k (R_k - R_k-1) P_k: recall only
//                                                                       moves at thresholds that are positive scores
//   FPR95 = fps / N at the threshold the reference's argmin |recall - 0.95| picks (ties -> higher recall, the
//           last threshold of that recall level, metric.py:117-127)
// -0.0 is folded onto +0.0 first (NumPy compares them equal, so they are one threshold).
#include <rocprim/device/device_radix_sort.hpp>

#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

constexpr int NT = 256;
typedef unsigned long long u64;

__device__ __forceinline__ uint32_t score_key(float v) {
  v += 0.0f;                                   // -0.0 -> +0.0
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One pass over a batch: the keys of the id_in pixels are packed at the front of keys[0..n) (ascending slots from
// 0), those of the id_out pixels at the back (descending slots from n-1); the two regions cannot meet. counts[0] /
// counts[1] (zero on entry) end as the two totals. A workgroup takes 4096 consecutive pixels at a time, ranks them
// with wave ballots and reserves its output range with ONE atomic per class (same-address atomics serialise: one per
// wave and step made this kernel 100x slower than the memory system). The order inside a region is irrelevant, the
// keys are sorted afterwards.
constexpr int ITEMS = 16;
__global__ __launch_bounds__(NT) void oodm_compact_kernel(const float* __restrict__ score, const long long* __restrict__ label,
                                                          long long n, long long id_in, long long id_out,
                                                          uint32_t* __restrict__ keys, u64* __restrict__ counts) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 below = (1ull << lane) - 1;
  __shared__ unsigned wneg[NT / 64], wpos[NT / 64];
  __shared__ u64 base[2];
  const long long chunk = (long long)NT * ITEMS;
  for (long long c0 = (long long)blockIdx.x * chunk; c0 < n; c0 += (long long)gridDim.x * chunk) {
    uint32_t key[ITEMS];
    u64 mneg[ITEMS], mpos[ITEMS];
    unsigned cneg = 0, cpos = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const long long i = c0 + (long long)j * NT + threadIdx.x;
      long long l = 0;
      float s = 0.f;
      if (i < n) { l = label[i]; s = score[i]; }
      key[j] = score_key(s);
      mneg[j] = __ballot(i < n && l == id_in);
      mpos[j] = __ballot(i < n && l == id_out);
      cneg += __popcll(mneg[j]);
      cpos += __popcll(mpos[j]);
    }
    if (lane == 0) { wneg[wave] = cneg; wpos[wave] = cpos; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned a = 0, b = 0;
      for (int w = 0; w < NT / 64; ++w) { a += wneg[w]; b += wpos[w]; }
      base[0] = a ? atomicAdd(&counts[0], (u64)a) : 0;
      base[1] = b ? atomicAdd(&counts[1], (u64)b) : 0;
    }
    __syncthreads();
    u64 oneg = base[0], opos = base[1];
    for (int w = 0; w < wave; ++w) { oneg += wneg[w]; opos += wpos[w]; }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if ((mneg[j] >> lane) & 1) keys[oneg + __popcll(mneg[j] & below)] = key[j];
      if ((mpos[j] >> lane) & 1) keys[n - 1 - (long long)(opos + __popcll(mpos[j] & below))] = key[j];
      oneg += __popcll(mneg[j]);
      opos += __popcll(mpos[j]);
    }
    __syncthreads();                                        // wneg / wpos / base are reused by the next chunk
  }
}

__device__ __forceinline__ long long lower_bound(const uint32_t* __restrict__ a, long long lo, long long hi, uint32_t v) {
  while (lo < hi) {                                        // first index with a[i] >= v
    const long long mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ long long upper_bound(const uint32_t* __restrict__ a, long long lo, long long hi, uint32_t v) {
  while (lo < hi) {                                        // first index with a[i] > v
    const long long mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Per block: u2[b] = sum over its positives of (2 #neg< + #neg==), ap[b] = sum of precision at the positive's
// threshold. pos / neg ascending. Partial sums are combined in block order by the finalize kernel (deterministic).
__global__ __launch_bounds__(NT) void oodm_rank_kernel(const uint32_t* __restrict__ pos, long long P,
                                                       const uint32_t* __restrict__ neg, long long N,
                                                       u64* __restrict__ u2_part, double* __restrict__ ap_part) {
  u64 u2 = 0;
  double ap = 0.0;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < P; i += (long long)gridDim.x * NT) {
    const uint32_t v = pos[i];
    const long long lbn = lower_bound(neg, 0, N, v);
    const long long ubn = upper_bound(neg, lbn, N, v);
    const long long lbp = lower_bound(pos, 0, i, v);       // pos[i] == v, so the first >= v is at or before i
    u2 += 2ull * (u64)lbn + (u64)(ubn - lbn);
    const double tps = (double)(P - lbp), fps = (double)(N - lbn);
    ap += tps / (tps + fps);
  }
  for (int o = 32; o > 0; o >>= 1) {
    u2 += __shfl_xor(u2, o);
    ap += __shfl_xor(ap, o);
  }
  __shared__ u64 su[NT / 64];
  __shared__ double sa[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { su[wave] = u2; sa[wave] = ap; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 a = 0;
    double b = 0.0;
    for (int w = 0; w < NT / 64; ++w) { a += su[w]; b += sa[w]; }
    u2_part[blockIdx.x] = a;
    ap_part[blockIdx.x] = b;
  }
}

// One thread: combine the partials and pick the FPR threshold. out = [auroc, aupr, fpr].
__global__ void oodm_finalize_kernel(const uint32_t* __restrict__ pos, long long P, const uint32_t* __restrict__ neg,
                                     long long N, const u64* __restrict__ u2_part, const double* __restrict__ ap_part,
                                     int nparts, double recall_level, double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u64 u2 = 0;
  double ap = 0.0;
  for (int b = 0; b < nparts; ++b) { u2 += u2_part[b]; ap += ap_part[b]; }
  out[0] = (double)u2 / (2.0 * (double)P * (double)N);
  out[1] = ap / (double)P;
  // recall takes the values c/P for c = #pos >= v over the distinct positive scores v. Candidates around
  // recall_level * P: both ends of the runs holding the floor(t)-th and (floor(t)+1)-th largest positives.
  const double pd = (double)P;
  long long i0 = (long long)floor(recall_level * pd);
  long long best_c = -1;
  double best_f = 0.0;
  for (int k = 0; k < 2; ++k) {
    long long c = i0 + k;
    if (c < 1) c = 1;
    if (c > P) c = P;
    const uint32_t v = pos[P - c];                         // the c-th largest positive
    const long long c_end = P - lower_bound(pos, 0, P, v);   // #pos >= v
    const long long c_start = P - upper_bound(pos, 0, P, v); // #pos >  v  (end of the previous run)
    const long long cand[2] = {c_start, c_end};
    for (int j = 0; j < 2; ++j) {
      if (cand[j] < 1) continue;
      const double f = fabs((double)cand[j] / pd - recall_level);
      if (best_c < 0 || f < best_f || (f == best_f && cand[j] > best_c)) { best_f = f; best_c = cand[j]; }
    }
  }
  long long fps;
  if (best_c == P) {
    fps = N - lower_bound(neg, 0, N, pos[0]);              // first threshold that reaches recall 1: #neg >= min positive
  } else {
    const uint32_t v_next = pos[P - best_c - 1];           // next lower positive score
    fps = N - upper_bound(neg, 0, N, v_next);              // last threshold of this recall level: #neg > v_next
  }
  out[2] = (double)fps / (double)N;
}

inline int grid_for(long long n) {
  long long b = (n + NT - 1) / NT;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

#define S_(x) static_cast<hipStream_t>(x)

extern "C" {

int mss_oodm_compact_f32(const float* score, const long long* label, long long n, long long id_in, long long id_out,
                         unsigned int* keys, unsigned long long* counts, void* stream) {
  if (!counts || n < 0 || id_in == id_out) return MSS_ERR_BAD_ARG;
  if (n == 0) return MSS_OK;
  if (!score || !label || !keys) return MSS_ERR_BAD_ARG;
  long long blocks = (n + (long long)NT * ITEMS - 1) / ((long long)NT * ITEMS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(oodm_compact_kernel, dim3((unsigned)blocks), dim3(NT), 0, S_(stream), score, label, n, id_in, id_out,
                     keys, counts);
  return mss_launch_status();
}

long long mss_oodm_sort_temp_bytes(long long n) {
  if (n < 0) return -1;
  size_t bytes = 0;
  const uint32_t* in = nullptr;
  uint32_t* out = nullptr;
  if (rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t)n) != hipSuccess) return -1;
  return (long long)(bytes < 16 ? 16 : bytes);
}

int mss_oodm_sort_u32(const unsigned int* keys_in, unsigned int* keys_out, long long n, void* temp, long long temp_bytes,
                      void* stream) {
  if (n < 0) return MSS_ERR_BAD_ARG;
  if (n == 0) return MSS_OK;
  if (!keys_in || !keys_out || !temp || keys_in == keys_out) return MSS_ERR_BAD_ARG;
  size_t bytes = (size_t)temp_bytes;
  hipError_t e = rocprim::radix_sort_keys(temp, bytes, keys_in, keys_out, (size_t)n, 0, 32, S_(stream));
  return e == hipSuccess ? mss_launch_status() : (int)e;
}

int mss_oodm_rank_blocks(long long P) { return grid_for(P); }

int mss_oodm_measures_f64(const unsigned int* pos_sorted, long long P, const unsigned int* neg_sorted, long long N,
                          double recall_level, unsigned long long* u2_part, double* ap_part, double* out, void* stream) {
  if (P < 1 || N < 1) return MSS_ERR_BAD_ARG;              // the caller returns None for an empty class (metric.py:176-180)
  if (!pos_sorted || !neg_sorted || !u2_part || !ap_part || !out) return MSS_ERR_BAD_ARG;
  const int nb = grid_for(P);
  hipLaunchKernelGGL(oodm_rank_kernel, dim3(nb), dim3(NT), 0, S_(stream), pos_sorted, P, neg_sorted, N, u2_part, ap_part);
  hipLaunchKernelGGL(oodm_finalize_kernel, dim3(1), dim3(64), 0, S_(stream), pos_sorted, P, neg_sorted, N, u2_part, ap_part,
                     nb, recall_level, out);
  return mss_launch_status();
}

}  // extern "C"
