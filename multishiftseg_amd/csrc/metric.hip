// Pixel-level OOD metrics on the device: AUROC, average precision (AUPRC) and FPR at 95 % recall over all
// labelled pixels of an evaluation sweep -- SURVEY 8(f)-1. Replaces eval_ood_measure / get_measures /
// fpr_and_fdr_at_recall (lib/utils/metric.py:87-127,130-153,170-180), which copy every full-resolution score
// map to the host, concatenate in NumPy and let sklearn argsort 1e7-1e8 pixels on one core.
//
// Exact, not binned: the positive (OOD, label == id_out) and negative (label == id_in) scores are compacted into
// two arrays of order-preserving 32-bit keys, each array is radix-sorted (own LSD sort below: 4 passes of 8 bits;
// the rocPRIM primitive it replaced stays reachable with MSS_OODM_SORT=rocprim for A/B timing only),
// and every metric is a function of rank counts found by binary search:
//   AUROC = sum_pos (#neg < s + 1/2 #neg == s) / (P N)              -- the area under sklearn's ROC trapezoids,
//                                                                       accumulated in 64-bit integers (exact)
//   AP    = 1/P sum_pos  #pos>=s / (#pos>=s + #neg>=s)              -- sklearn's sum_k (R_k - R_k-1) P_k: recall only
//                                                                       moves at thresholds that are positive scores
//   FPR95 = fps / N at the threshold the reference's argmin |recall - 0.95| picks (ties -> higher recall, the
//           last threshold of that recall level, metric.py:117-127)
// -0.0 is folded onto +0.0 first (NumPy compares them equal, so they are one threshold).
#include <rocprim/device/device_radix_sort.hpp>

#include <stdlib.h>
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
// PACKED (r04): both totals in ONE 64-bit counter (id_in count in the low, id_out count in the high 32 bits; n < 2^32) -- one
// same-address atomic per 4096 pixels instead of two: the 512 + 512 serialised atomics of a 1024 x 2048 map were most of the
// kernel's 26 us.
// LANES (r06): the packed counter is still ONE address the 512 workgroups of a 1024 x 2048 map add to one after the other (an
// atomic with a returned value: ~40 ns each, the kernel's 20 us). With LANES = 8 the 4096-pixel chunks are dealt round-robin onto 8
// counters, each owning its own segment [lane * cap, (lane + 1) * cap) of the key buffer (cap = its chunks x 4096 pixels, so a
// segment cannot overflow): inliers from the front of the segment, OOD from its back, eight independent chains of 64 atomics.
constexpr int ITEMS = 16;
template <bool PACKED, int LANES = 1>
__device__ __forceinline__ void oodm_compact_body(const float* __restrict__ score, const long long* __restrict__ label,
                                                  long long n, long long id_in, long long id_out,
                                                  uint32_t* __restrict__ keys_all, u64* __restrict__ counts_all, long long cap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 below = (1ull << lane) - 1;
  __shared__ unsigned wneg[NT / 64], wpos[NT / 64];
  __shared__ u64 base[2];
  const long long chunk = (long long)NT * ITEMS;
  for (long long c0 = (long long)blockIdx.x * chunk; c0 < n; c0 += (long long)gridDim.x * chunk) {
    const int ln = LANES > 1 ? (int)((c0 / chunk) % LANES) : 0;
    uint32_t* __restrict__ keys = keys_all + (LANES > 1 ? (size_t)ln * cap : 0);
    u64* __restrict__ counts = counts_all + ln;
    const long long seg_n = LANES > 1 ? cap : n;         // OOD keys fill the segment from its last slot downwards
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
      if (PACKED) {
        const u64 old = (a | b) ? atomicAdd(&counts[0], (u64)a | ((u64)b << 32)) : 0;
        base[0] = old & 0xffffffffull;
        base[1] = old >> 32;
      } else {
        base[0] = a ? atomicAdd(&counts[0], (u64)a) : 0;
        base[1] = b ? atomicAdd(&counts[1], (u64)b) : 0;
      }
    }
    __syncthreads();
    u64 oneg = base[0], opos = base[1];
    for (int w = 0; w < wave; ++w) { oneg += wneg[w]; opos += wpos[w]; }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if ((mneg[j] >> lane) & 1) keys[oneg + __popcll(mneg[j] & below)] = key[j];
      if ((mpos[j] >> lane) & 1) keys[seg_n - 1 - (long long)(opos + __popcll(mpos[j] & below))] = key[j];
      oneg += __popcll(mneg[j]);
      opos += __popcll(mpos[j]);
    }
    __syncthreads();                                        // wneg / wpos / base are reused by the next chunk
  }
}

template <bool PACKED, int LANES = 1>
__global__ __launch_bounds__(NT) void oodm_compact_kernel(const float* __restrict__ score, const long long* __restrict__ label,
                                                          long long n, long long id_in, long long id_out,
                                                          uint32_t* __restrict__ keys_all, u64* __restrict__ counts_all, long long cap = 0) {
  oodm_compact_body<PACKED, LANES>(score, label, n, id_in, id_out, keys_all, counts_all, cap);
}

// up to MSS_OODM_BATCH maps in ONE launch (blockIdx.y = map): an update is ~8 us of kernel behind ~20 us of host work per call, so a
// sweep that holds its maps hands them over sixteen at a time (mss_oodm_compact_lanes_batch_f32)
__global__ __launch_bounds__(NT) void oodm_compact_batch_kernel(MssOodmBatch b, long long id_in, long long id_out) {
  const int m = blockIdx.y;
  const long long n = b.n[m];
  if ((long long)blockIdx.x * NT * ITEMS >= n) return;
  oodm_compact_body<true, 8>(b.score[m], b.label[m], n, id_in, id_out, b.keys[m], reinterpret_cast<u64*>(b.lane_counts[m]),
                             (((n + (long long)NT * ITEMS - 1) / ((long long)NT * ITEMS) + 7) / 8) * (long long)NT * ITEMS);
}

// One map's eight lane segments -> its slices of the sweep's contiguous id_in / id_out key arrays (what torch.cat did with 16 views per
// map): lane L's id_in keys to neg_out + (sum of the lanes before it), likewise id_out. The lane counts are read on the device.
__global__ __launch_bounds__(256) void oodm_gather_lanes_kernel(const uint32_t* __restrict__ keys, long long cap, const u64* __restrict__ lane_counts,
                                                                uint32_t* __restrict__ neg_out, uint32_t* __restrict__ pos_out) {
  const int ln = blockIdx.y;
  u64 nb = 0, pb = 0;
  for (int l = 0; l < ln; ++l) { nb += lane_counts[l] & 0xffffffffull; pb += lane_counts[l] >> 32; }
  const u64 a = lane_counts[ln] & 0xffffffffull, b = lane_counts[ln] >> 32;
  const uint32_t* seg = keys + (size_t)ln * cap;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < a; i += (u64)gridDim.x * 256) neg_out[nb + i] = seg[i];
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < b; i += (u64)gridDim.x * 256) pos_out[pb + i] = seg[cap - b + i];
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

namespace {

// ------------------------------------------------------------------------------------------
// LSD radix sort of 32-bit keys, 8 bits per pass, stable. The array is cut into NC <= 1024 contiguous chunks (whole tiles
// of 4096 keys); per pass:
//   rs_hist    : workgroup c histograms the pass digit over chunk c (LDS integer atomics) -> hist[c][256]
//   rs_scan    : workgroup d turns digit d's column of hist into its exclusive prefix over the chunks + the digit total
//   rs_scatter : workgroup c walks chunk c tile by tile. Keys are held wave-striped (wave w, round i, lane l <-> key
//                w*1024 + i*64 + l of the tile); a key's rank among equal digits is found with 8 ballots (the lanes that
//                agree on every digit bit) + a per-wave running count in LDS; the tile is first laid out digit-sorted in
//                LDS and then written, so a wave stores runs of consecutive addresses per digit instead of 4-byte
//                scatters. The running offsets of the chunk live in LDS and advance by the tile's digit counts.
// Traffic: 12 B per key and pass (histogram read, scatter read + write), no atomics on global memory.
constexpr int RS_NT = 256, RS_ITEMS = 16, RS_TILE = RS_NT * RS_ITEMS, RS_MAXCHUNKS = 1024;

__global__ __launch_bounds__(RS_NT) void rs_hist_kernel(const uint32_t* __restrict__ keys, long long n, long long chunk, int shift,
                                                        uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const long long lo = (long long)blockIdx.x * chunk, hi = min(n, lo + chunk);
  // 16-byte loads from the first aligned key on (the key arrays are slices of the compaction buffer: 4-byte aligned only)
  const long long mis = (long long)((reinterpret_cast<uintptr_t>(keys) >> 2) & 3);
  long long lo4 = lo + ((4 - ((mis + lo) & 3)) & 3);
  if (lo4 > hi) lo4 = hi;
  for (long long i = lo4 + 4ll * threadIdx.x; i + 3 < hi; i += 4ll * RS_NT) {
    const uint4 k = *reinterpret_cast<const uint4*>(keys + i);
    atomicAdd(&h[(k.x >> shift) & 255], 1u);
    atomicAdd(&h[(k.y >> shift) & 255], 1u);
    atomicAdd(&h[(k.z >> shift) & 255], 1u);
    atomicAdd(&h[(k.w >> shift) & 255], 1u);
  }
  {
    const long long tail = lo4 + ((hi - lo4) & ~3ll);       // the last 0..3 keys of the chunk, and its first 0..3
    if (tail + threadIdx.x < hi) atomicAdd(&h[(keys[tail + threadIdx.x] >> shift) & 255], 1u);
    if (lo + threadIdx.x < lo4) atomicAdd(&h[(keys[lo + threadIdx.x] >> shift) & 255], 1u);
  }
  __syncthreads();
  hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// exclusive scan over the 256 threads of a workgroup (4 waves): shuffles inside a wave, one LDS hop across waves
__device__ __forceinline__ uint32_t rs_block_excl_scan(uint32_t v, uint32_t* wsum /* [4] shared */, uint32_t* total_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  const uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
  const uint32_t base = wave == 0 ? 0u : wave == 1 ? w0 : wave == 2 ? w0 + w1 : w0 + w1 + w2;
  if (total_out) *total_out = w0 + w1 + w2 + w3;
  __syncthreads();                                           // wsum may be reused by the caller's next scan
  return base + inc - v;
}

// hist[c][d] (counts) -> exclusive prefix over the chunks of digit d, in place, + totals[d]. Workgroup d owns digit d,
// thread t the chunks 4t .. 4t+3 (nchunks <= 1024).
__global__ __launch_bounds__(256) void rs_scan_kernel(uint32_t* __restrict__ hist, int nchunks, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[4];
  const int d = blockIdx.x, t = threadIdx.x;
  uint32_t v[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * t + j;
    v[j] = c < nchunks ? hist[(size_t)c * 256 + d] : 0u;
    sum += v[j];
  }
  uint32_t total;
  uint32_t run = rs_block_excl_scan(sum, wsum, &total);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * t + j;
    if (c < nchunks) hist[(size_t)c * 256 + d] = run;
    run += v[j];
  }
  if (t == 0) totals[d] = total;
}

__global__ __launch_bounds__(RS_NT) void rs_scatter_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long long n,
                                                           long long chunk, int shift, const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ totals) {
  __shared__ uint32_t run_off[256];            // global offset of the next key of each digit, for this chunk
  __shared__ uint32_t wcount[4][256];          // per wave: keys of each digit seen so far in the tile
  __shared__ uint32_t tile_excl[256];          // exclusive scan of the tile's digit totals
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t sorted[RS_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1;
  run_off[tid] = rs_block_excl_scan(totals[tid], wsum, nullptr) + offsets[(size_t)blockIdx.x * 256 + tid];
  const long long lo = (long long)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (long long t0 = lo; t0 < hi; t0 += RS_TILE) {
#pragma unroll
    for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
    __syncthreads();
    uint32_t key[RS_ITEMS];
    unsigned short rank[RS_ITEMS];             // rank among the wave's keys of the same digit
    const long long wbase = t0 + (long long)wave * (RS_ITEMS * 64);
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
      const long long g = wbase + i * 64 + lane;
      key[i] = g < hi ? in[g] : 0xffffffffu;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
      const long long g = wbase + i * 64 + lane;
      const bool valid = g < hi;
      const unsigned d = (key[i] >> shift) & 255;
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((d >> b) & 1);
        m &= ((d >> b) & 1) ? bal : ~bal;
      }
      // m: the valid lanes of this round with my digit (including me when valid)
      const int leader = __ffsll((long long)m) - 1;
      uint32_t prev = 0;
      if (valid && lane == leader) {
        prev = wcount[wave][d];
        wcount[wave][d] = prev + (uint32_t)__popcll(m);
      }
      prev = __shfl(prev, leader < 0 ? 0 : leader);
      rank[i] = (unsigned short)(prev + (uint32_t)__popcll(m & lt));
    }
    __syncthreads();
    // digit totals of the tile, their exclusive scan, and the per-wave bases
    const uint32_t c0 = wcount[0][tid], c1 = wcount[1][tid], c2 = wcount[2][tid], c3 = wcount[3][tid];
    const uint32_t total = c0 + c1 + c2 + c3;
    const uint32_t excl = rs_block_excl_scan(total, wsum, nullptr);     // (its barriers also order the reads above)
    tile_excl[tid] = excl;
    wcount[0][tid] = excl;                     // reuse: position of wave w's first key of digit tid in the sorted tile
    wcount[1][tid] = excl + c0;
    wcount[2][tid] = excl + c0 + c1;
    wcount[3][tid] = excl + c0 + c1 + c2;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
      const long long g = wbase + i * 64 + lane;
      if (g < hi) {
        const unsigned d = (key[i] >> shift) & 255;
        sorted[wcount[wave][d] + rank[i]] = key[i];
      }
    }
    __syncthreads();
    const int cnt = (int)min((long long)RS_TILE, hi - t0);
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
      const int j = i * RS_NT + tid;
      if (j < cnt) {
        const uint32_t k = sorted[j];
        const unsigned d = (k >> shift) & 255;
        out[run_off[d] + (uint32_t)j - tile_excl[d]] = k;
      }
    }
    __syncthreads();
    run_off[tid] += total;
  }
}

struct RsPlan { int nchunks; long long chunk; };
inline RsPlan rs_plan(long long n) {
  RsPlan p;
  long long tiles = (n + RS_TILE - 1) / RS_TILE;
  if (tiles < 1) tiles = 1;
  p.nchunks = (int)(tiles < RS_MAXCHUNKS ? tiles : RS_MAXCHUNKS);
  p.chunk = ((tiles + p.nchunks - 1) / p.nchunks) * RS_TILE;
  p.nchunks = (int)((n + p.chunk - 1) / p.chunk);
  if (p.nchunks < 1) p.nchunks = 1;
  return p;
}
inline bool rs_use_rocprim() {
  static std::atomic<long long> slot{-(1ll << 32)};        // cached like MSS_ENV_INT (mss_common.h); the value is a word here
  const int gen = mss_env_generation();
  const long long v = slot.load(std::memory_order_relaxed);
  if ((int)(v >> 32) == gen) return (v & 1) != 0;
  const char* e = getenv("MSS_OODM_SORT");
  const bool r = e && e[0] == 'r';
  slot.store(((long long)gen << 32) | (r ? 1 : 0), std::memory_order_relaxed);
  return r;
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
  hipLaunchKernelGGL(oodm_compact_kernel<false>, dim3((unsigned)blocks), dim3(NT), 0, S_(stream), score, label, n, id_in, id_out,
                     keys, counts);
  return mss_launch_status();
}

int mss_oodm_compact_packed_f32(const float* score, const long long* label, long long n, long long id_in, long long id_out,
                                unsigned int* keys, unsigned long long* packed_count, void* stream) {
  if (!packed_count || n < 0 || n >= (1ll << 32) || id_in == id_out) return MSS_ERR_BAD_ARG;
  if (n == 0) return MSS_OK;
  if (!score || !label || !keys) return MSS_ERR_BAD_ARG;
  long long blocks = (n + (long long)NT * ITEMS - 1) / ((long long)NT * ITEMS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(oodm_compact_kernel<true>, dim3((unsigned)blocks), dim3(NT), 0, S_(stream), score, label, n, id_in, id_out,
                     keys, packed_count);
  return mss_launch_status();
}

// keys: 8 * mss_oodm_compact_lanes_cap(n) slots; lane_counts: device u64[8], zero on entry, ends as #id_in | (#id_out << 32) per lane;
// lane L's id_in keys are keys[L * cap .. + #id_in), its id_out keys keys[(L + 1) * cap - #id_out .. (L + 1) * cap)
long long mss_oodm_compact_lanes_cap(long long n) {
  if (n <= 0) return 0;
  const long long chunks = (n + (long long)NT * ITEMS - 1) / ((long long)NT * ITEMS);
  return ((chunks + 7) / 8) * (long long)NT * ITEMS;
}

int mss_oodm_compact_lanes_f32(const float* score, const long long* label, long long n, long long id_in, long long id_out,
                               unsigned int* keys, unsigned long long* lane_counts, void* stream) {
  if (!lane_counts || n < 0 || n >= (1ll << 32) || id_in == id_out) return MSS_ERR_BAD_ARG;
  if (n == 0) return MSS_OK;
  if (!score || !label || !keys) return MSS_ERR_BAD_ARG;
  long long blocks = (n + (long long)NT * ITEMS - 1) / ((long long)NT * ITEMS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((oodm_compact_kernel<true, 8>), dim3((unsigned)blocks), dim3(NT), 0, S_(stream), score, label, n, id_in, id_out,
                     keys, lane_counts, mss_oodm_compact_lanes_cap(n));
  return mss_launch_status();
}

int mss_oodm_compact_lanes_batch_f32(const MssOodmBatch* batch, int count, long long id_in, long long id_out, void* stream) {
  if (!batch || count < 0 || count > MSS_OODM_BATCH || id_in == id_out) return MSS_ERR_BAD_ARG;
  long long most = 0;
  for (int m = 0; m < count; ++m) {
    const long long n = batch->n[m];
    if (n < 0 || n >= (1ll << 32)) return MSS_ERR_BAD_ARG;
    if (n > 0 && (!batch->score[m] || !batch->label[m] || !batch->keys[m] || !batch->lane_counts[m])) return MSS_ERR_BAD_ARG;
    if (n > most) most = n;
  }
  if (count == 0 || most == 0) return MSS_OK;
  long long blocks = (most + (long long)NT * ITEMS - 1) / ((long long)NT * ITEMS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(oodm_compact_batch_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(NT), 0, S_(stream), *batch, id_in, id_out);
  return mss_launch_status();
}

int mss_oodm_gather_lanes_u32(const unsigned int* keys, long long cap, const unsigned long long* lane_counts, unsigned int* neg_out,
                              unsigned int* pos_out, void* stream) {
  if (cap < 0 || (cap > 0 && (!keys || !lane_counts || !neg_out || !pos_out))) return MSS_ERR_BAD_ARG;
  if (cap == 0) return MSS_OK;
  long long blocks = (cap + 4095) / 4096;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(oodm_gather_lanes_kernel, dim3((unsigned)blocks, 8), dim3(256), 0, S_(stream), keys, cap, lane_counts, neg_out, pos_out);
  return mss_launch_status();
}

long long mss_oodm_sort_temp_bytes(long long n) {
  if (n < 0 || n >= (1ll << 32)) return -1;
  size_t bytes = 0;
  const uint32_t* in = nullptr;
  uint32_t* out = nullptr;
  if (rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t)n) != hipSuccess) return -1;   // A/B route only
  // own sort: a ping-pong key buffer (16-byte aligned) + the [chunks][256] offset table
  const long long own = ((n * 4 + 15) & ~15ll) + (long long)(RS_MAXCHUNKS + 1) * 256 * 4;
  const long long need = own > (long long)bytes ? own : (long long)bytes;
  return need < 16 ? 16 : need;
}

int mss_oodm_sort_u32(const unsigned int* keys_in, unsigned int* keys_out, long long n, void* temp, long long temp_bytes,
                      void* stream) {
  if (n < 0 || n >= (1ll << 32)) return MSS_ERR_BAD_ARG;
  if (n == 0) return MSS_OK;
  if (!keys_in || !keys_out || !temp || keys_in == keys_out) return MSS_ERR_BAD_ARG;
  if (rs_use_rocprim()) {
    size_t bytes = (size_t)temp_bytes;
    hipError_t e = rocprim::radix_sort_keys(temp, bytes, keys_in, keys_out, (size_t)n, 0, 32, S_(stream));
    return e == hipSuccess ? mss_launch_status() : (int)e;
  }
  const long long kb = (n * 4 + 15) & ~15ll;
  if (temp_bytes < kb + (long long)(RS_MAXCHUNKS + 1) * 256 * 4) return MSS_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(keys_in) | reinterpret_cast<uintptr_t>(keys_out) | reinterpret_cast<uintptr_t>(temp)) & 3)
    return MSS_ERR_BAD_ARG;
  uint32_t* ping = static_cast<uint32_t*>(temp);
  uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(temp) + kb);
  uint32_t* totals = hist + (size_t)RS_MAXCHUNKS * 256;
  const RsPlan pl = rs_plan(n);
  const uint32_t* src = keys_in;
  for (int pass = 0; pass < 4; ++pass) {
    uint32_t* dst = (pass & 1) ? keys_out : ping;           // in -> ping -> out -> ping -> out
    hipLaunchKernelGGL(rs_hist_kernel, dim3(pl.nchunks), dim3(RS_NT), 0, S_(stream), src, n, pl.chunk, 8 * pass, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(256), 0, S_(stream), hist, pl.nchunks, totals);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(pl.nchunks), dim3(RS_NT), 0, S_(stream), src, dst, n, pl.chunk, 8 * pass, hist,
                       totals);
    src = dst;
  }
  return mss_launch_status();
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
