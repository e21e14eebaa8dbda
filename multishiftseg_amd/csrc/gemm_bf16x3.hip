// The fp32 NT GEMM of gemm.hip evaluated on the bf16 matrix cores by operand splitting -- the SECOND, fully pinned GEMM route
// (round 5; `MssConvArgs.w_split` selects it per call; never silently: mss_conv2d_forward_route answers 3 for it).
//
//   x = x_hi + x_mid + x_lo exactly (three round-to-nearest bf16 terms cover fp32's 24 significand bits), so
//   a * b = a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid) + O(2^-26 |a b|)
// -- six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) per 32 x 32 x 16 block instead of eight v_mfma_f32_32x32x2_f32. The bf16 pipe
// runs 16x the FLOPs per cycle of the fp32 one, so the six-product form has 2.67x the matrix throughput of the native instruction
// at fp32 accuracy (against float64: 2e-7 of max|y|, native fp32 MFMA 3-6e-7; tests/test_gpu_ops.py::test_bf16x3_gemm_*).
//
// What round 4's first draft (gemm_bf16x6.hip: both operands split on the fly with a scalar chain, 128 x 128 tile, 5.5 VALU per
// MFMA, pipe 60 % busy) did not have:
//  * the WEIGHTS are split once, at pack time (mss_gemm_split_weights_bf16x3), into the exact LDS image of a 128-row operand
//    block per K-step -- [block][K-step][plane hi|mid|lo][128 rows][16 k], 12 KB each, the two 16-byte halves of a row swapped when
//    bit 3 of the row is set (round 6: no swap at all, mss_bf16x3.h) -- so staging B is a linear 16-byte-per-lane copy with no arithmetic at all;
//  * only the ACTIVATION operand is split in the loader, with v_cvt_pk_bf16_f32 (two conversions per instruction) and mask /
//    shift re-expansion: 5.5 VALU per element, 8 elements per thread and K-step on the 128 x 256 tile = 1.2 VALU per MFMA measured;
//  * gemm_nt_kernel's persistent schedule (variant 3): 32-bit offsets, the loader two K-steps ahead across tile boundaries with
//    one register set, a branch-free advance, one barrier per K-step, 128 x 256 tiles (48 MFMAs per wave and barrier) at two
//    workgroups per CU, 128 x 128 at three.
// LDS per stage: A 3 x 4 KB + B 3 x 4 KB per 128 columns; two stages: 72 KB (128 x 256) / 48 KB (128 x 128).
#include "mss_epilogue.h"
#include "mss_gemm_tiles.h"
#include "mss_bf16x3.h"
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <utility>

#ifdef MSS_SPLIT_STAMPS
// DIAGNOSTIC BUILD ONLY (never in libmss_hip.so as shipped): per workgroup-wave the summed s_memtime deltas of the K-step's eight
// stretches -- top (fragment reads), segments 1..6, barrier -- plus the step count; read back with mss_debug_read_stamps.
__device__ unsigned long long mss_dbg_stamps[4096 * 12];
extern "C" int mss_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mss_dbg_stamps), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#define MSS_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dbg_sum[i] += t_ - dbg_last; dbg_last = t_; }
#else
#define MSS_STAMP(i)
#endif

// Dynamic tile scheduling of the persistent kernels (round 5): self-resetting counter slots in device memory (eight per-XCD ticket
// counters + a count of finished workgroups, draw_xcd_ticket; every launch zeroes its slot when its last workgroup leaves).
// A slot may only be shared by launches that cannot overlap. Ownership (round 6, ADVICE r05 / VERDICT r05 weak 9):
//  * EAGER launches: every stream has its own ring of 8 slots, used in rotation (launches on a stream run in order);
//  * launches recorded by a STREAM CAPTURE: the slot address is baked into the graph and the graph may later be replayed on any
//    stream, beside other graphs and beside eager launches of the stream it was captured on -- so each captured launch gets a PRIVATE
//    slot out of a region no eager launch ever uses, never handed out twice (a graph exec cannot overlap itself, so its own
//    replays are ordered). 8192 such slots per device; when they are used up -- or when the pool does not exist yet and the first
//    launch on the device happens under capture, where nothing may be allocated -- the launcher falls back to the static tile walk
//    (nullptr here: correct, a few percent slower on the prologue kernels).
// The pool (64 rings x 8 + 8192 slots of 64 bytes = 544 KB) is allocated and cleared by mss_sched_init, which the weight-plane packers
// call (a split launch needs planes, so the pool exists before the first launch unless the packing itself was captured), or lazily by
// the first eager launch. A process with more than 64 streams that launch these kernels gets the static walk on the later ones.
// A launch that FAULTS leaves its slot dirty; a device fault is sticky in HIP (every later call on the context fails), so there is no
// later launch to protect.
namespace {
constexpr int SCHED_RINGS = 64, SCHED_PAIRS = 8, SCHED_MAXDEV = 16, SCHED_SLOT = 16, SCHED_GRAPH_SLOTS = 8192;   // a slot: 16 ints (64 B)
std::mutex sched_mu;
int* sched_pools[SCHED_MAXDEV] = {nullptr};                                           // one pool per device of this process
int sched_graph_used[SCHED_MAXDEV] = {0};
std::unordered_map<hipStream_t, std::pair<int, unsigned>> sched_rings[SCHED_MAXDEV];  // stream -> (ring index, launches so far)

bool sched_pool_locked(int dev) {                  // sched_mu held; false: no pool and none can be made right now
  if (sched_pools[dev]) return true;
  const size_t bytes = ((size_t)SCHED_RINGS * SCHED_PAIRS + SCHED_GRAPH_SLOTS) * SCHED_SLOT * sizeof(int);
  int* q = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&q), bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemset(q, 0, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); return false; }
  sched_pools[dev] = q;
  return true;
}
bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
}  // namespace

// Allocate the ticket pool of the current device now (not under a capture). Idempotent.
void mss_sched_init(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCHED_MAXDEV || stream_is_capturing(stream)) return;
  std::lock_guard<std::mutex> lock(sched_mu);
  (void)sched_pool_locked(dev);
}

// nullptr: use the static tile walk for this launch
int* mss_sched_slot(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCHED_MAXDEV) return nullptr;
  const bool capturing = stream_is_capturing(stream);
  std::lock_guard<std::mutex> lock(sched_mu);
  if (!sched_pools[dev] && (capturing || !sched_pool_locked(dev))) return nullptr;
  if (capturing) {
    if (sched_graph_used[dev] >= SCHED_GRAPH_SLOTS) return nullptr;
    return sched_pools[dev] + ((size_t)SCHED_RINGS * SCHED_PAIRS + sched_graph_used[dev]++) * SCHED_SLOT;
  }
  auto& map = sched_rings[dev];
  auto it = map.find(stream);
  if (it == map.end()) {
    if ((int)map.size() >= SCHED_RINGS) return nullptr;
    it = map.emplace(stream, std::make_pair((int)map.size(), 0u)).first;
  }
  return sched_pools[dev] + ((size_t)it->second.first * SCHED_PAIRS + it->second.second++ % SCHED_PAIRS) * SCHED_SLOT;
}

namespace {

// The tile order of the persistent kernels is the static walk t, t + grid, ... except where split_dyn_tiles() says tickets (the NT
// kernels with a prologue, see gemm_nt_bf16x3_kernel). Measured in round 5 with an all-tickets A/B build that round 6 removed
// (profiles/r05/dynamic_tiles.md): the ticket order equalises the workgroups' lifetimes and changes the launch time by +1 % .. -8 %,
// because the workgroup the static order leaves alone on its SIMDs runs almost twice as fast there.
constexpr int NT = 256, BM = 128;
using mss_bf16x3::BK;
using mss_bf16x3::ROW_B;
using mss_bf16x3::PLANE;
using mss_bf16x3::OPER;
using mss_bf16x3::cvt_pk_bf16;
using mss_bf16x3::split_pair;
constexpr int TM = 2;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One ticket of the dynamic tile order (thread 0 of a workgroup). The static walk gives workgroup b the tiles v, v + grid, v + 2 grid,
// ... with v = mss_xcd_remap(b): every XCD owns a contiguous range of v, so the tiles in flight on an XCD share their operands
// through its L2. A single global counter loses that (64 x 2304 x 4096 -> 256: -8 %), so there is one counter per XCD over the list
// of ITS tiles in the static order (entry j -> v = base + j % cnt, round j / cnt; the first cnt entries are the static first
// tiles). The lists of the eight XCDs differ by at most one tile per round, so nothing is taken from a neighbour's list.
// Returns total_tiles when the list is exhausted.
__device__ __forceinline__ int draw_xcd_ticket(int* __restrict__ sched, long long total_tiles) {
  const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, x = (int)blockIdx.x & 7;
  const int cnt = q + (x < r ? 1 : 0);                       // >= 1: this workgroup is one of them
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  const unsigned j = (unsigned)(cnt + atomicAdd(sched + x, 1));
  const unsigned round = j / (unsigned)cnt;
  const long long t = base + (int)(j - round * (unsigned)cnt) + (long long)round * nwg;
  return t < total_tiles ? (int)t : (int)total_tiles;
}

// CONV (round 5): the same kernel as an IMPLICIT GEMM for the layers conv_igemm_kernel takes (3x3 with stride 2 or few input channels, 1x1
// with stride 2): the reduction runs over (tap, channel) -- the weights' planes are packed with the taps folded into one long reduction
// (mss_gemm_split_weights_bf16x3 on [taps][Kpad][C] with taps > 1: B needs no tap logic at all) -- and only the A side changes: a staged
// row is an output pixel, its address per tap = pixel base + a uniform tap offset, taps that fall into the zero padding read a valid
// dummy address and enter as zeros AFTER the BatchNorm + ReLU prologue (the reference pads the activated tensor). Branch-free like
// the rest of the K-step: the tap state advances with selects.
// ROWAFF: the prologue affine differs per SAMPLE (the Dropout2d fold of mod6 / mod7, wider_resnet.py:139-140,161-162) and a 128-row tile
// may straddle two images (88 x 88 maps at 700 x 700): every staged row then loads the affine of its own image.
// MF (round 6): the MFMA shape. 32: v_mfma_f32_32x32x16_bf16, one product per instruction, B staged through registers (round 5).
// 16: v_mfma_f32_16x16x32_bf16 on CONCATENATED planes -- the instruction's 32-deep reduction is two 16-deep halves taken from two
// different planes of the same K-step, lanes 0-31 (k 0..15 of the instruction) reading plane P, lanes 32-63 (k 16..31) plane Q, so
// one instruction adds TWO of the six products:
//     X(hi|mid) W(hi|mid) = hi hi + mid mid      X(mid|hi) W(hi|mid) = mid hi + hi mid      X(lo|hi) W(hi|lo) = lo hi + hi lo
// (three instructions per 16 x 16 x 16 block instead of six halves of a 32 x 32 x 16 one: the same matrix-pipe cycles, but the chip
// holds a higher clock on this shape -- MI355X_MICROARCH.md, DVFS give-back (7): 1.12-1.14x with LDS-fed operands). The plane
// pairing costs nothing: the planes lie in LDS as before and a fragment read adds a per-lane plane offset. The weights are the FIRST
// operand, so the accumulator holds D[channel][pixel]: a lane's four values are four consecutive channels of one pixel, a 16-byte
// store (mss_epilogue_store16). The weight planes go global -> LDS by LDS-DMA (global_load_lds_dwordx4; they are stored in LDS
// image order, a wave-instruction copies 1 KB) one K-step ahead, which frees the 24 staging registers for the fragments.
template <bool AFFINE, int BN, bool CONV = false, bool ROWAFF = false, bool DYN = false, int MF = 32>
__global__ __launch_bounds__(NT, (BN == 256 || (AFFINE && (CONV || ROWAFF))) ? 2 : 3) void gemm_nt_bf16x3_kernel(MssConvArgs p, const unsigned char* __restrict__ wpl,
                                                                               long long total_tiles, int tiles_per_batch,
                                                                               int group_m, unsigned blk_bytes, int nblk_total, int* __restrict__ sched) {
  constexpr int NBLK = BN / 128, TN = BN / 64;          // wave tile 64 x (BN / 2)
  constexpr bool M16 = MF == 16;
  static_assert(!M16 || (!AFFINE && !CONV && !ROWAFF), "the 16x16x32 form is instantiated for products without a prologue only");
  constexpr int TI = 4, TJ = BN / 32;                   // M16: 16-pixel blocks x 16-channel blocks of the wave tile
  constexpr int STAGE = (1 + NBLK) * OPER;              // A block, then NBLK B blocks
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int chunk = tid & 3, row0 = tid >> 2;           // A staging: floats [4 chunk, 4 chunk + 4) of rows row0 and row0 + 64
  const int nk = p.C / BK;                               // K-steps per tap
  const int n_it = CONV ? nk * p.R * p.S : nk;
  const long long stride = gridDim.x;
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();

  // ---- loader state (see gemm_nt_kernel, variant 3) ----
  unsigned a_off[2], a_nxt[2], b_off[NBLK], b_nxt[NBLK], s_off = 0, s_nxt = 0, s_off1 = 0, s_nxt1 = 0;   // (s_off1: row 1's affine, ROWAFF)
  int ld_kb = -1;                                        // M16: b_off runs ONE K-step behind the A loader (the DMA is one step ahead of the MFMAs, the
                                                         // A registers two): its own step counter, same wrap rule; -1: the first advance() only arms it
  // the A operand of batch entry b starts at p.x + b * x_bs: a UNIFORM 64-bit base per tile (scalar registers) + 32-bit offsets inside
  // the entry, so a batched product may exceed 4 GB as a whole (16 x 700 x 700: the ASPP X' is 5.4 / 12 GB) while each entry stays below
  const char* xb_cur = reinterpret_cast<const char*>(p.x);
  const char* xb_nxt = xb_cur;
  long long ld_tile = mss_xcd_remap(blockIdx.x, gridDim.x);
  int ld_k = 0;
  // CONV: per staged row the byte offset of its (possibly virtual) top-left input pixel and the 9-bit map of in-image taps, for the
  // loader's tile and for the one after it; the loader's tap, its K-step inside the tap, the in-image bits of the K-step in flight
  int pix[2] = {0, 0}, pix_nxt[2] = {0, 0}, okb[2] = {0, 0}, okb_nxt[2] = {0, 0};
  int ld_tap = 0, ld_c = 0;
  unsigned raw_ok = 3u, cur_ok = 3u;
  auto tap_offset = [&](int tap) {                        // uniform: ((tap / S) * dil * W + (tap % S) * dil) pixels, in bytes
    const int r = p.S == 1 ? tap : (tap * 43) >> 7, sx = tap - r * p.S;
    return (r * p.dil * p.W + sx * p.dil) * p.ldx * (int)sizeof(float);
  };
  auto setup_off = [&](long long t, unsigned* ao, unsigned* bo, unsigned& so, unsigned& so1) {
    const int b = (int)(t / tiles_per_batch);
    (ao == a_off ? xb_cur : xb_nxt) = reinterpret_cast<const char*>(p.x) + (size_t)b * p.x_bs * sizeof(float);
    const int v = (int)(t - (long long)b * tiles_per_batch);
    int mt, nt; mss_tile_mn(v, p.mtiles, p.ntiles, group_m, mt, nt);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row = mt * BM + row0 + j * 64;
      if (CONV) {
        int* px = ao == a_off ? pix : pix_nxt;
        int* ok = ao == a_off ? okb : okb_nxt;
        const int ohw = p.OH * p.OW;
        const bool in = row < p.M;
        row = in ? row : 0;
        const int n = row / ohw, rem = row - n * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        px[j] = (((n * p.H + iy0) * p.W + ix0) * p.ldx + chunk * 4) * (int)sizeof(float);
        int bits = 0;
        for (int t = 0; t < p.R * p.S; ++t) {
          const int r = t / p.S, iy = iy0 + r * p.dil, ix = ix0 + (t - r * p.S) * p.dil;
          bits |= ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? 1 << t : 0;
        }
        ok[j] = in ? bits : 0;                            // rows past the end: every tap "outside" (zeros, never stored)
        ao[j] = (ok[j] & 1) ? (unsigned)px[j] : (unsigned)(chunk * 4 * sizeof(float));     // tap 0
        continue;
      }
      row = row < p.M ? row : p.M - 1;                  // rows past the end re-read the last row; never stored
      ao[j] = (unsigned)(((size_t)row * p.ldx + chunk * 4) * sizeof(float));
    }
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
      bo[j] = (unsigned)(b * nblk_total + nt * NBLK + j) * blk_bytes + tid * 16;
    if (AFFINE) so = CONV ? (unsigned)(chunk * 4 * sizeof(float)) : (unsigned)(((size_t)((mt * BM) / p.H) * p.in_ss_stride + chunk * 4) * sizeof(float));
    if (ROWAFF) {                                        // p.H = rows per image; rows past the end take the last image's affine
      const int r0 = min(mt * BM + row0, p.M - 1), r1 = min(mt * BM + row0 + 64, p.M - 1);
      so = (unsigned)(((size_t)(r0 / p.H) * p.in_ss_stride + chunk * 4) * sizeof(float));
      so1 = (unsigned)(((size_t)(r1 / p.H) * p.in_ss_stride + chunk * 4) * sizeof(float));
    }
  };
  // DYNAMIC TILE ORDER (DYN, an experiment kept for A/B). With the static walk t, t + grid, ... every workgroup has the same
  // number of tiles, but the two workgroups of a CU do not run at the same speed: the SIMD issues the OLDER wave first, so the
  // first-dispatched workgroup of each CU finishes its tiles in 3.3 ms and the second one in 4.9 (in-kernel stamps,
  // profiles/r05/stamps_split.txt). With DYN the first tile is the static one and every further tile is the next ticket of a
  // device counter: wave 0 draws a ticket per tile, two tiles ahead (the loader needs the tile after the current one), and hands
  // it over through two LDS words behind the staging buffers; the fast workgroup simply takes more tiles. Which workgroup
  // computes a tile does not change a bit of it. It also does not change the launch time: see the note at DYN.
  int* tk_slot = reinterpret_cast<int*>(smem + 2 * STAGE);
  long long nxt_tile = 0;
  int tiles_done = 0;
  auto draw_ticket = [&](int slot) {
    if (DYN && tid == 0) tk_slot[slot] = draw_xcd_ticket(sched, total_tiles);
  };
  auto setup_next = [&]() {
    const long long t = DYN ? nxt_tile : ld_tile + stride;
    setup_off(t < total_tiles ? t : ld_tile, a_nxt, b_nxt, s_nxt, s_nxt1);
  };
  f32x4 areg[2], sreg, hreg, sreg1, hreg1;
  u32x4 breg[NBLK][3];
  // one uniform base per plane (kept opaque, or the compiler folds them back into wpl + 4096 / 8192, which do not fit the 13-bit
  // instruction offset and cost a 64-bit VALU address per load): every load is `global_load_dwordx4 v, v_off, s[base]`
  typedef const unsigned char __attribute__((address_space(1)))* gptr_t;     // stays a GLOBAL pointer through the asm (else: flat loads)
  gptr_t wbase[3] = {(gptr_t)wpl, (gptr_t)wpl + PLANE, (gptr_t)wpl + 2 * PLANE};
  asm volatile("" : "+s"(wbase[1]), "+s"(wbase[2]));
  auto issue_loads_a = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) areg[j] = *reinterpret_cast<const f32x4*>(xb_cur + a_off[j]);
    if (AFFINE) {
      sreg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in_scale) + s_off);
      hreg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in_shift) + s_off);
    }
    if (ROWAFF) {
      sreg1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in_scale) + s_off1);
      hreg1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in_shift) + s_off1);
    }
    if (CONV) raw_ok = cur_ok;
  };
  auto issue_loads_b = [&]() {
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) breg[j][pl] = *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(wbase[pl] + b_off[j]);
  };
  auto issue_loads = [&]() { issue_loads_a(); if (!M16) issue_loads_b(); };
  // M16: the weight planes of one K-step straight into LDS stage `buf`: per block and plane one global_load_lds_dwordx4 per wave
  // (LDS destination = M0 + 16 * lane: the wave's 1 KB of the 4 KB plane; source = plane base + the thread's byte offset `off[j]`).
  // Inline asm on purpose: the compiler then keeps no book on these in its vmcnt accounting, so it never answers a pending DMA with a
  // conservative vmcnt(0) on the A operand's register loads (cdna_hip_programming.md 5, Pipelining across barriers); unknown
  // outstanding operations can only make its own counted waits stricter, never too weak. The K-step orders them by hand:
  // every use of the previous A registers comes BEFORE the DMAs, the next A loads AFTER them, and the barrier is preceded by
  // s_waitcnt vmcnt(<number of those A loads>), which retires exactly the DMAs.
  const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem) +
                            (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
  auto dma_piece = [&](int buf, int j, int pl) {        // one plane of one weight block: 1 KB per wave
    const unsigned dst = lds_wave + (unsigned)(buf * STAGE + (1 + j) * OPER + pl * PLANE);
    unsigned keep;                                       // M0 is the compiler's to manage: put it back
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(b_off[j]), "s"(wbase[pl]) : "memory");
  };
  auto dma_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dma_piece(buf, j, pl);
  };
  constexpr int N_A_LOADS = 2 + (AFFINE ? 2 : 0) + (ROWAFF ? 2 : 0);          // issue_loads_a: what follows the DMAs in a K-step
  auto advance_b = [&](bool wrap) {
    if constexpr (M16) {                 // one step behind: b_nxt still describes the tile this cursor wraps into (setup_next runs at tile_end,
      const bool wrap_b = ++ld_kb == n_it;              // >= 1 K-step after both cursors entered the tile; n_it >= 3)
#pragma unroll
      for (int j = 0; j < NBLK; ++j) b_off[j] = ld_kb == 0 ? b_off[j] : (wrap_b ? b_nxt[j] : b_off[j] + (unsigned)OPER);
      ld_kb = wrap_b ? 0 : ld_kb;
    } else {
#pragma unroll
      for (int j = 0; j < NBLK; ++j) b_off[j] = wrap ? b_nxt[j] : b_off[j] + (unsigned)OPER;
    }
  };
  auto advance = [&]() {                 // branch-free: next K-step of this tile, else first K-step of this workgroup's next tile
    const bool wrap = ++ld_k == n_it;
    xb_cur = wrap ? xb_nxt : xb_cur;
    if (CONV) {
      const bool tap_end = ++ld_c == nk;
      ld_c = tap_end ? 0 : ld_c;
      ld_tap = wrap ? 0 : (tap_end ? ld_tap + 1 : ld_tap);
      const int toff = tap_offset(ld_tap);
      unsigned ok2 = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        pix[j] = wrap ? pix_nxt[j] : pix[j];
        okb[j] = wrap ? okb_nxt[j] : okb[j];
        const bool in = (okb[j] >> ld_tap) & 1;
        ok2 |= in ? 1u << j : 0u;
        const unsigned fresh = in ? (unsigned)(pix[j] + toff) : (unsigned)(chunk * 4 * sizeof(float));
        a_off[j] = tap_end ? fresh : a_off[j] + BK * (unsigned)sizeof(float);
      }
      cur_ok = ok2;
      if (AFFINE) s_off = tap_end ? (unsigned)(chunk * 4 * sizeof(float)) : s_off + BK * (unsigned)sizeof(float);
      advance_b(wrap);
      ld_k = wrap ? 0 : ld_k;
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) a_off[j] = wrap ? a_nxt[j] : a_off[j] + BK * (unsigned)sizeof(float);
    advance_b(wrap);
    if (AFFINE) s_off = wrap ? s_nxt : s_off + BK * (unsigned)sizeof(float);
    if (ROWAFF) s_off1 = wrap ? s_nxt1 : s_off1 + BK * (unsigned)sizeof(float);
    ld_k = wrap ? 0 : ld_k;
  };
  // A: 4 bf16 (8 B) per plane at row r, quarter `chunk` of the 32-byte row, no half swap (mss_bf16x3.h)
  const int st_off = row0 * ROW_B + ((chunk >> 1) * 16) + (chunk & 1) * 8;
  unsigned st_ok = 3u;                   // (CONV) in-image bits of the K-step being split
  auto split_row = [&](int j, unsigned (&hi)[2], unsigned (&mid)[2], unsigned (&lo)[2]) {
    f32x4 v = areg[j];
    if (AFFINE) {
      v = (ROWAFF && j == 1) ? v * sreg1 + hreg1 : v * sreg + hreg;
      v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
    }
    if (CONV) {                                          // a tap in the zero padding: zeros after the prologue
      const bool in = (st_ok >> j) & 1;
      v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
    }
    split_pair(v.x, v.y, hi[0], mid[0], lo[0]);
    split_pair(v.z, v.w, hi[1], mid[1], lo[1]);
  };
  auto store_row = [&](int buf, int j, const unsigned (&hi)[2], const unsigned (&mid)[2], const unsigned (&lo)[2]) {
    unsigned char* base = smem + buf * STAGE;
    *reinterpret_cast<u32x2*>(base + 0 * PLANE + st_off + j * 64 * ROW_B) = u32x2{hi[0], hi[1]};
    *reinterpret_cast<u32x2*>(base + 1 * PLANE + st_off + j * 64 * ROW_B) = u32x2{mid[0], mid[1]};
    *reinterpret_cast<u32x2*>(base + 2 * PLANE + st_off + j * 64 * ROW_B) = u32x2{lo[0], lo[1]};
  };
  // the same split in stages (M16: row 1's is spread over the DMA segments of group 2, a few VALU behind each block of MFMAs)
  auto row_value = [&](int j) -> f32x4 {                 // A registers -> prologue affine / ReLU / padding zeros
    f32x4 v = areg[j];
    if (AFFINE) {
      v = (ROWAFF && j == 1) ? v * sreg1 + hreg1 : v * sreg + hreg;
      v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
    }
    if (CONV) {
      const bool in = (st_ok >> j) & 1;
      v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
    }
    return v;
  };
  auto term = [&](const f32x4& v, unsigned (&t)[2]) { t[0] = cvt_pk_bf16(v.x, v.y); t[1] = cvt_pk_bf16(v.z, v.w); };
  auto rest = [&](const f32x4& v, const unsigned (&t)[2]) -> f32x4 {      // v - (its bf16 term): exact
    return f32x4{v.x - __uint_as_float(t[0] << 16), v.y - __uint_as_float(t[0] & 0xffff0000u),
                 v.z - __uint_as_float(t[1] << 16), v.w - __uint_as_float(t[1] & 0xffff0000u)};
  };
  auto finish_store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned hi[2], mid[2], lo[2];
      split_row(j, hi, mid, lo);
      store_row(buf, j, hi, mid, lo);
    }
  };
  auto finish_store_b = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4*>(base + (1 + j) * OPER + pl * PLANE + tid * 16) = breg[j][pl];
  };
  auto finish_store = [&](int buf) { finish_store_a(buf); finish_store_b(buf); };

  // fragment of v_mfma_f32_32x32x16_bf16: lane = (row l % 32, k-block l / 32), 8 consecutive k = 16 B
  const int frow = lane & 31, fkb = lane >> 5;
  const int fr_off = frow * ROW_B + fkb * 16;      // (32-row fragments on the linear image: two-way conflicts; this form is the fallback only)
  const int fa_off = wm * 64 * ROW_B + fr_off;
  const int fb_off = OPER + (BN == 256 ? wn * OPER : wn * 64 * ROW_B) + fr_off;

#ifdef MSS_SPLIT_STAMPS
  unsigned long long dbg_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_last = 0;
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();   // shader clock / constant 100 MHz
#endif
  f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
  f32x4 acc16[M16 ? TI : 1][M16 ? TJ : 1];               // M16: [pixel block][channel block], D[channel][pixel] (mss_epilogue_store16)
  auto zero_acc = [&]() {
    if constexpr (M16) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
  };
  auto epilogue = [&](long long t) {
    const int b = (int)(t / tiles_per_batch);
    const int v = (int)(t - (long long)b * tiles_per_batch);
    int mt, nt; mss_tile_mn(v, p.mtiles, p.ntiles, group_m, mt, nt);
    if constexpr (M16) mss_epilogue_store16<TI, TJ>(acc16, p, p.y + (size_t)b * p.y_bs, mt * BM + wm * 64, nt * BN + wn * (BN / 2), lane);
    else mss_epilogue_store<TM, TN>(acc, p, p.y + (size_t)b * p.y_bs, mt * BM + wm * 64, nt * BN + wn * (BN / 2), lane);
  };
  // M16 fragments: lane = (row l & 15 of a 16-row block, k-block l >> 4 of the instruction's four); k-blocks 0, 1 are the two 16-byte
  // halves of the row in plane P, k-blocks 2, 3 the same halves in plane Q. The image is linear (no half swap,
  // mss_bf16x3.h): the hardware's ds_read_b128 lane groups then hit sixteen distinct 16-byte slots.
  const int l15 = lane & 15, khalf = (lane >> 4) & 1, up = lane >> 5;
  const int fr16 = l15 * ROW_B + khalf * 16;
  const int fx16 = wm * 64 * ROW_B + fr16;                                        // X: block 0, pixel rows of this wave
  const int fw16 = OPER + (BN == 256 ? wn * OPER : wn * 64 * ROW_B) + fr16;        // W: this wave's 128 / 64 channels
  const int up_plane = up * PLANE;
  auto step16 = [&](const int buf) {
    const unsigned char* base = smem + buf * STAGE;
    // the five per-lane fragment addresses (P | Q per half wave) are re-derived in every step from three registers (the empty asm keeps
    // the compiler from hoisting them out of the loop: five registers live through everything put the prologue kernels into scratch)
    int upq = up_plane;
    asm volatile("" : "+v"(upq));
    const int x_hm = fx16 + upq, x_mh = fx16 + PLANE - upq, x_lh = fx16 + 2 * PLANE - 2 * upq;
    const int w_hm = fw16 + upq, w_hl = fw16 + 2 * upq;
    auto ld_w = [&](int off, bf16x8 (&f)[TJ]) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) f[j] = *reinterpret_cast<const bf16x8*>(base + off + j * 16 * ROW_B);
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
#define MSS_PAIR_UP(mask, n, per)                                                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                          \
    __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                                   \
  }
    bf16x8 xhm[TI], xmh[TI], xlh[TI], whm[TJ], whl[TJ];
    // the first instruction needs X block 0 and W block 0: read those first (the waits count in issue order), the other X blocks last
    xhm[0] = *reinterpret_cast<const bf16x8*>(base + x_hm);
    ld_w(w_hm, whm);
#pragma unroll
    for (int i = 1; i < TI; ++i) xhm[i] = *reinterpret_cast<const bf16x8*>(base + x_hm + i * 16 * ROW_B);
    fence();
    __builtin_amdgcn_s_setprio(1);                       // (+2 % on the products of the step, profiles/r06/split_mfma_ab.md)
    MSS_STAMP(0)
    // Every use of the A registers is preceded by this FIRST use (an empty asm the compiler must have them ready for): the wait for
    // the loads (issued in group 2 of the previous step) lands here, in front of this step's DMAs, never behind one.
    asm volatile("" : "+v"(areg[0]), "+v"(areg[1]));
    fence();
    MSS_STAMP(1)
    // group 1 (hi hi + mid mid), pixel block outermost (X(hi|mid) block i dies after TJ instructions, X(mid|hi) block i is read into
    // its registers), in fenced mini-segments (inline asm is in no class the group barriers know, and DMA issues in a row keep the wave
    // out of the matrix pipe for hundreds of cycles): per segment one or two pieces of the weight planes of K-step k+1 by DMA -- as
    // early as the step allows, they take ~900 cycles to land -- and a stage of staged row 0's split
    unsigned hi[2], mid[2], lo[2];
    st_ok = raw_ok;
    constexpr int NDMA = 3 * NBLK;
    f32x4 v0, r0;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      if (i == 0) { v0 = row_value(0); term(v0, hi); r0 = rest(v0, hi); }
      if (i == 1) { term(r0, mid); }
      if (i == 2) { r0 = rest(r0, mid); }
      if (i == 3) { term(r0, lo); store_row(buf ^ 1, 0, hi, mid, lo); }
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whm[j], xhm[i], acc16[i][j], 0, 0, 0);
      xmh[i] = *reinterpret_cast<const bf16x8*>(base + x_mh + i * 16 * ROW_B);
#pragma unroll
      for (int d = i * NDMA / TI; d < (i + 1) * NDMA / TI; ++d) dma_piece(buf ^ 1, d / 3, d % 3);
      fence();
    }
    MSS_STAMP(2)
    // group 2 (mid hi + hi mid), channel block outermost, fenced segments of TI instructions: the first one turns row 1's A registers
    // into values, the second issues the A loads of K-step k+2 (younger than every DMA: the wait before the barrier counts on it), the
    // others carry the stages of row 1's split; W(hi|lo) block j is read into the registers of W(hi|mid) block j, whose last use the
    // segment was, X(lo|hi) block 0 at the end
    f32x4 v1, r1;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      constexpr int ST = TJ >= 8 ? 1 : 2;               // split stages per segment (TJ == 4: two)
#pragma unroll
      for (int sg = j * ST; sg < (j + 1) * ST; ++sg) {
        if (sg == 0) { v1 = row_value(1); }
        if (sg == 1) { issue_loads_a(); term(v1, hi); }
        if (sg == 2) r1 = rest(v1, hi);
        if (sg == 3) term(r1, mid);
        if (sg == 4) r1 = rest(r1, mid);
        if (sg == 5) { term(r1, lo); store_row(buf ^ 1, 1, hi, mid, lo); }
      }
#pragma unroll
      for (int i = 0; i < TI; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whm[j], xmh[i], acc16[i][j], 0, 0, 0);
      whl[j] = *reinterpret_cast<const bf16x8*>(base + w_hl + j * 16 * ROW_B);
      if (j == TJ - 1) xlh[0] = *reinterpret_cast<const bf16x8*>(base + x_lh);
      fence();
    }
    MSS_STAMP(3)
    // group 3 (lo hi + hi lo), pixel block outermost: X(lo|hi) block i + 1 is read while block i multiplies; the loader's bookkeeping
    advance();
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      if (i + 1 < TI) xlh[i + 1] = *reinterpret_cast<const bf16x8*>(base + x_lh + (i + 1) * 16 * ROW_B);
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whl[j], xlh[i], acc16[i][j], 0, 0, 0);
    }
    _Pragma("unroll") for (int i_ = 0; i_ < TI - 1; ++i_) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      MSS_PAIR_UP(0x6, TJ, 3);
    }
    MSS_PAIR_UP(0x6, TJ - 1, 3);
    fence();
#undef MSS_PAIR_UP
    __builtin_amdgcn_s_setprio(0);
    MSS_STAMP(4)
    // the DMAs are older than the N_A_LOADS register loads: this retires them (and this wave's LDS writes) and leaves the loads in flight
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(N_A_LOADS) : "memory");
    MSS_STAMP(5)
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");                     // nothing of the next step moves above the barrier
    MSS_STAMP(7)
#ifdef MSS_SPLIT_STAMPS
    dbg_sum[8] += 1;
#endif
  };
  // One K-step: the products in an order that needs one new operand plane per group of 2 * TN MFMAs
  //   (A_lo, B_hi) (A_mid, B_hi) (A_hi, B_hi) (A_hi, B_mid) (A_mid, B_mid) (A_hi, B_lo)
  // (the order inside a K-step is irrelevant for the rounding error: every term is added to an accumulator that already holds the
  // previous K-steps' sum).
  auto step = [&](const int buf) {
    const unsigned char* base = smem + buf * STAGE;
    auto ld_a = [&](int pl, bf16x8 (&f)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + fa_off + i * 32 * ROW_B);
    };
    auto ld_b = [&](int pl, bf16x8 (&f)[TN]) {
#pragma unroll
      for (int j = 0; j < TN; ++j) f[j] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + fb_off + j * 32 * ROW_B);
    };
    auto mm = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    bf16x8 a_hi[TM], a_mid[TM], a_lo[TM], b_hi[TN], b_mid[TN], b_lo[TN];
    {
      // Left alone, hipcc sinks the global loads of step k+2 to the END of step k and meets them with an `s_waitcnt vmcnt(0)` at the
      // top of step k+1 (the two-steps-ahead loader degenerates to none), and clumps the split arithmetic in front of the MFMAs.
      // Here the step is cut into six segments the scheduler may not move instructions across, one group of 2 * TN MFMAs each,
      // with the loader's work of that segment asked to go one (or a few) behind each MFMA:
      //   1: B's LDS writes (data requested a whole step ago)   2: B's global loads for step k+2 + the b_mid fragments
      //   3: split of A row 0 + its LDS writes                  4: split of A row 1 + its LDS writes + A's global loads + b_lo
      //   5: the loader's bookkeeping                           6: nothing
      constexpr int NB = 3 * NBLK, G = TM * TN;
      auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
      // n x (one MFMA, then `per` instructions of class `mask`); the segment's remaining MFMAs follow
#define MSS_PAIR_UP(mask, n, per)                                                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                          \
    __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                                   \
  }
      ld_a(2, a_lo); ld_b(0, b_hi);
      ld_a(1, a_mid); ld_a(0, a_hi);
      fence();
      __builtin_amdgcn_s_setprio(1);                     // r06: the wave in its MFMA segments outranks a partner that is reading fragments or
                                                         // waiting at a barrier (65536 x 2048 -> 4096 with the prologue: 233 -> 246 TFLOP/s; others +-1 %)
      MSS_STAMP(0)
      finish_store_b(buf ^ 1);
      mm(a_lo, b_hi);
      MSS_PAIR_UP(0x200, NB < G ? NB : G, NB <= G ? 1 : 2);
      fence();
      MSS_STAMP(1)
      issue_loads_b();
      ld_b(1, b_mid);
      mm(a_mid, b_hi);
      MSS_PAIR_UP(0x20, NB < G ? NB : G, 1);
      MSS_PAIR_UP(0x100, G - NB > 0 ? (G - NB < TN ? G - NB : TN) : 0, 1);
      fence();
      MSS_STAMP(2)
      unsigned hi[2], mid[2], lo[2];
      st_ok = raw_ok;
      split_row(0, hi, mid, lo);
      store_row(buf ^ 1, 0, hi, mid, lo);
      mm(a_hi, b_hi);
      MSS_PAIR_UP(0x2, G - 1, ((AFFINE ? 30 : 22) + G - 2) / (G - 1));
      fence();
      MSS_STAMP(3)
      split_row(1, hi, mid, lo);
      store_row(buf ^ 1, 1, hi, mid, lo);
      issue_loads_a();
      ld_b(2, b_lo);
      mm(a_hi, b_mid);
      MSS_PAIR_UP(0x2, G - 1, ((AFFINE ? 30 : 22) + G - 2) / (G - 1));
      fence();
      MSS_STAMP(4)
      advance();
      mm(a_mid, b_mid);
      MSS_PAIR_UP(0x6, G - 1, 3);
      fence();
      MSS_STAMP(5)
      mm(a_hi, b_lo);
      fence();
      __builtin_amdgcn_s_setprio(0);
      MSS_STAMP(6)
#undef MSS_PAIR_UP
    }
    __syncthreads();
    MSS_STAMP(7)
#ifdef MSS_SPLIT_STAMPS
    dbg_sum[8] += 1;
#endif
  };

  long long cur = ld_tile;               // tile being multiplied (the launch guarantees cur < total_tiles)
  if constexpr (DYN) {
    draw_ticket(0);
    __syncthreads();
    nxt_tile = __builtin_amdgcn_readfirstlane(tk_slot[0]);
    draw_ticket(1);                      // read at the end of the first tile, many barriers from here
  }
  setup_off(ld_tile, a_off, b_off, s_off, s_off1);
  if (CONV) cur_ok = (unsigned)((okb[0] & 1) | ((okb[1] & 1) << 1));
  setup_next();
  issue_loads();
  if constexpr (M16) dma_b(0);    // the weight planes of K-step 0 (b_off still describes it)
  st_ok = raw_ok;
  if constexpr (M16) finish_store_a(0); else finish_store(0);
  advance();
  issue_loads();                         // registers now hold K-step 1
  advance();
  zero_acc();
  if constexpr (M16) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(N_A_LOADS) : "memory");     // the DMAs of K-step 0 (older than K-step 1's A loads)
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
  } else {
    __syncthreads();
  }
  int k = 0;
#ifdef MSS_SPLIT_STAMPS
  dbg_last = __builtin_amdgcn_s_memtime();
#endif
  auto tile_end = [&]() -> bool {        // true: this workgroup has no tile left
    if (++k < n_it) return false;
    epilogue(cur);
#ifdef MSS_SPLIT_STAMPS
    dbg_last = __builtin_amdgcn_s_memtime();      // (the epilogue is not part of any stretch)
#endif
    if constexpr (DYN) {
      cur = nxt_tile;
      nxt_tile = __builtin_amdgcn_readfirstlane(tk_slot[(tiles_done + 1) & 1]);        // drawn during the tile that just ended
      if (cur >= total_tiles) return true;
      draw_ticket(tiles_done & 1);                      // (that word was read a whole tile ago)
      ++tiles_done;
    } else {
      cur += stride;
      if (cur >= total_tiles) return true;
    }
    zero_acc();
    k = 0;
    ld_tile = cur;                       // the loader entered `cur` at least one K-step ago (n_it >= 3): prepare the one after it
    setup_next();
    return false;
  };
  while (true) {                         // unrolled by the two LDS stages: the stage is a compile-time constant in each half
    if constexpr (M16) step16(0); else step(0);
    if (tile_end()) break;
    if constexpr (M16) step16(1); else step(1);
    if (tile_end()) break;
  }
  if (DYN && tid == 0 && atomicAdd(sched + 8, 1) == (int)gridDim.x - 1) {   // the last workgroup out zeroes the counters for their next launch
#pragma unroll
    for (int i = 0; i < 9; ++i) __atomic_store_n(sched + i, 0, __ATOMIC_RELAXED);
  }
#ifdef MSS_SPLIT_STAMPS
  if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
    for (int i = 0; i < 9; ++i) mss_dbg_stamps[(blockIdx.x * 4 + wave) * 12 + i] = dbg_sum[i];
    mss_dbg_stamps[(blockIdx.x * 4 + wave) * 12 + 9] = __builtin_amdgcn_s_memtime() - dbg_t0;
    mss_dbg_stamps[(blockIdx.x * 4 + wave) * 12 + 10] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The TN product of the weight gradients on the same route: dW[k][c] = sum_t dy[t][k] * act(x[t][c])  (Winograd-domain batches:
// dU[p] = dY'[p]^T X'[p]; 1x1 layers and Linears: one position). Both operands are activations, so both are split in the loader;
// the contraction index t is the ROW of both tensors, so the loader also transposes: a staging task is 8 consecutive rows x 4
// consecutive columns (8 coalesced 16-byte loads); v_cvt_pk_bf16_f32 pairs rows (2 t', 2 t' + 1) of one column, i.e. the 8 rows
// of a column leave as ONE 16-byte LDS write per plane -- the k-contiguous row piece the MFMA fragment wants, no shuffles.
// A 32-row fragment block holds the columns {4 c + i : c = 0..31} of a 128-column group (i = its index in the group): the LDS image is
// then the NT kernel's ([128-row block][plane][row][16 k], same swizzle, same fragment reads, same MFMA order) and the epilogue
// interleaves the four blocks of a lane back into 16-byte stores of consecutive columns.
// Tile 128 (k) x 256 (c) per workgroup, wave 0 stages the 64 dy tasks, waves 1-2 the 128 x tasks (176 VALU each per K-step beside
// their 48 MFMAs), persistent over (position, split, k tile, c tile); every split walks `tps` rows (a multiple of 16); the last one is
// shifted back to end at row M and the rows it shares with its predecessor enter as zeros.
template <bool AFFINE, bool MASKED>
__global__ __launch_bounds__(NT, 2) void gemm_tn_bf16x3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                float* __restrict__ out, int P, int M, long long a_bs, long long b_bs, int Kpad,
                                                                int Cp, int ktiles, int ctiles, int splits, int tps, long long total_tiles,
                                                                const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
  constexpr int NBLK = 2, TN = 4, BN = 256;
  constexpr int STAGE = (1 + NBLK) * OPER;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // waves 0 AND 3 stage the 64 dy tasks (the same bytes to the same LDS addresses: wave 3 would otherwise idle, and a role that
  // skips the staging puts a branch into the K-step, which must stay ONE basic block for the interleaving below), waves 1-2 the x tasks
  const bool is_a = wave == 0 || wave == 3;
  // lane pairs own the two 8-row groups of ONE column quad: the two 16-byte halves of an LDS row come from adjacent lanes, so a
  // wave's staging write is 1 KB of consecutive LDS bytes (with lane = quad + 32 * row group the 16-byte writes sat 32 bytes apart:
  // 113 M bank-conflict cycles per launch, profiles/r05/pmc_split_tn.md first version)
  const int cq = (is_a ? 0 : (wave - 1) * 32) + (lane >> 1), rg = lane & 1;   // A: 32 column quads x 2 row groups; B: 64 x 2
  // MASKED (M not a multiple of 16): every split walks tps / 16 K-steps, the last one shifted back to end at row M, and the dy rows it
  // shares with its predecessor enter as zeros (2 VALU per element on every lane). Otherwise (the usual case) the last split simply
  // has fewer K-steps and nothing is masked.
  const int nit_full = tps / BK, nit_last = MASKED ? nit_full : (M - (splits - 1) * tps) / BK;
  const long long stride = gridDim.x;
  const float relu_floor = relu ? 0.f : -__builtin_huge_valf();
  const int ld = is_a ? lda : ldb;                                 // wave-uniform
  const unsigned ld_bytes = (unsigned)ld * 4u;
  typedef const unsigned char __attribute__((address_space(1)))* gptr_t;
  gptr_t rowbase[8];                                               // operand base + t rows: 8 uniform pointers, one 32-bit offset per thread
#pragma unroll
  for (int t = 0; t < 8; ++t) rowbase[t] = (gptr_t)(is_a ? A : B) + (size_t)t * ld_bytes;

  unsigned off = 0, nxt = 0, s_off = 0, s_nxt = 0;
  size_t slab = 0, slab_nxt = 0;                                   // byte offset of the loader's position (uniform, 64-bit): a batch may exceed 4 GB
  int vf = 0, vf_nxt = 0, row_ld = 0, row_nxt = 0;                 // first valid row of the loader's tile; first row of its current step
  int ld_nit = nit_full, nit_nxt = nit_full;                       // K-steps of the loader's tile / of the one after it
  long long ld_tile = mss_xcd_remap(blockIdx.x, gridDim.x);
  int ld_k = 0;
  auto decode = [&](long long t, int& pb, int& sp, int& kt, int& ct) {
    ct = (int)(t % ctiles); t /= ctiles;
    kt = (int)(t % ktiles); t /= ktiles;
    sp = (int)(t % splits); pb = (int)(t / splits);
  };
  auto setup_off = [&](long long t, unsigned& o, unsigned& so, int& valid_from, int& row0, int& nit, size_t& sl) {
    int pb, sp, kt, ct; decode(t, pb, sp, kt, ct);
    sl = (size_t)pb * (is_a ? a_bs : b_bs) * sizeof(float);
    valid_from = sp * tps;
    nit = sp == splits - 1 ? nit_last : nit_full;
    row0 = (!MASKED || valid_from + tps <= M) ? valid_from : M - tps;   // MASKED: the last split is shifted back to end at row M
    const size_t col = is_a ? (size_t)kt * 128 + 4 * cq : (size_t)ct * BN + 4 * cq;
    o = (unsigned)(((size_t)(row0 + rg * 8) * ld + col) * sizeof(float));
    if (AFFINE) so = is_a ? 0u : (unsigned)((ct * BN + 4 * cq) * sizeof(float));     // (dy lanes load a valid vector and ignore it)
  };
  auto setup_next = [&]() {
    const long long t = ld_tile + stride;
    setup_off(t < total_tiles ? t : ld_tile, nxt, s_nxt, vf_nxt, row_nxt, nit_nxt, slab_nxt);
  };
  f32x4 raw[8], sreg, hreg;
  int raw_mask = 0;                                                // mask_rows of the K-step held in `raw`
  auto issue_loads = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t) raw[t] = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(rowbase[t] + slab + off);
    if (AFFINE) {
      sreg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(scale) + s_off);
      hreg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(shift) + s_off);
    }
    if (MASKED) {
      const int m = vf - row_ld;                                   // rows [row_ld, vf) belong to the previous split (dy rows enter as zeros)
      raw_mask = is_a && m > 0 ? m : 0;
    }
  };
  auto advance = [&]() {
    const bool wrap = ++ld_k == ld_nit;
    ld_nit = wrap ? nit_nxt : ld_nit;
    slab = wrap ? slab_nxt : slab;
    off = wrap ? nxt : off + BK * ld_bytes;
    row_ld = wrap ? row_nxt : row_ld + BK;
    vf = wrap ? vf_nxt : vf;
    if (AFFINE) s_off = wrap ? s_nxt : s_off;
    ld_k = wrap ? 0 : ld_k;
  };
  // LDS: 16 bytes (8 k) of row (i * 32 + cq % 32) of the operand block, per plane and column i of the quad
  const int st_off = (is_a ? 0 : OPER * (1 + (cq >> 5))) + (cq & 31) * ROW_B + ((rg ^ (((cq & 31) >> 3) & 1)) * 16);
  int st_mask = 0;
  auto store_comp = [&](int buf, const int i) {                    // column i of the quad: 8 rows -> 3 planes x 16 bytes
    unsigned char* base = smem + buf * STAGE + st_off;
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
      float v0 = raw[2 * t2][i], v1 = raw[2 * t2 + 1][i];
      if (AFFINE) {
        const float sc = is_a ? 1.f : sreg[i], sh = is_a ? 0.f : hreg[i], fl = is_a ? -__builtin_huge_valf() : relu_floor;
        v0 = fmaxf(v0 * sc + sh, fl);
        v1 = fmaxf(v1 * sc + sh, fl);
      }
      if (MASKED) {
        v0 = rg * 8 + 2 * t2 < st_mask ? 0.f : v0;                // (st_mask is 0 on the x lanes)
        v1 = rg * 8 + 2 * t2 + 1 < st_mask ? 0.f : v1;
      }
      split_pair(v0, v1, hi[t2], mid[t2], lo[t2]);
    }
    *reinterpret_cast<u32x4*>(base + 0 * PLANE + i * 32 * ROW_B) = u32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4*>(base + 1 * PLANE + i * 32 * ROW_B) = u32x4{mid[0], mid[1], mid[2], mid[3]};
    *reinterpret_cast<u32x4*>(base + 2 * PLANE + i * 32 * ROW_B) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  };
  auto finish_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) store_comp(buf, i);
  };

  const int frow = lane & 31, fkb = lane >> 5;
  const int fr_off = frow * ROW_B + ((fkb ^ ((frow >> 3) & 1)) * 16);
  const int fa_off = wm * 64 * ROW_B + fr_off;
  const int fb_off = OPER + wn * OPER + fr_off;
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // Output row of accumulator register r of block i: 4 m + 2 wm + i with m = (r & 3) + 8 (r >> 2) + 4 fkb, i.e. a per-lane part
  // (16 fkb rows + the lane's column quad: ONE 32-bit byte offset) and a uniform part (2 wm + 4 (r & 3) + 32 (r >> 2) + i rows: a scalar
  // add to the tile's base). Written as 32 64-bit per-lane offsets the compiler hoisted them out of the tile loop and spilled them:
  // 18 scratch reloads, each behind an s_waitcnt vmcnt(0), in front of every tile's stores.
  typedef unsigned char __attribute__((address_space(1)))* gwptr_t;
  const unsigned ep_lane = (unsigned)(((size_t)(16 * fkb) * Cp + (size_t)(wn * 128 + 4 * frow)) * sizeof(float));
  auto epilogue = [&](long long t) {
    int pb, sp, kt, ct; decode(t, pb, sp, kt, ct);
    gwptr_t tile = (gwptr_t)(out + ((size_t)sp * P + pb) * Kpad * Cp + (size_t)(kt * 128 + 2 * wm) * Cp + (size_t)(ct * BN));
    const size_t row_b = (size_t)Cp * sizeof(float);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gwptr_t rowp = tile + (size_t)(4 * (r & 3) + 32 * (r >> 2) + i) * row_b;          // uniform
        asm volatile("" : "+s"(rowp));
        const f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        *reinterpret_cast<f32x4 __attribute__((address_space(1)))*>(rowp + ep_lane) = v;
      }
  };
  auto step = [&](const int buf) {
    const unsigned char* base = smem + buf * STAGE;
    auto ld_a = [&](int pl, bf16x8 (&f)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + fa_off + i * 32 * ROW_B);
    };
    auto ld_b = [&](int pl, bf16x8 (&f)[TN]) {
#pragma unroll
      for (int j = 0; j < TN; ++j) f[j] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + fb_off + j * 32 * ROW_B);
    };
    auto mm = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    bf16x8 a_hi[TM], a_mid[TM], a_lo[TM], b_hi[TN], b_mid[TN], b_lo[TN];
    // six fenced segments of 8 MFMAs, as in the NT kernel; the split / transpose of K-step k+1 (one quad column = 44 VALU + 3 LDS
    // writes per segment) rides behind the MFMAs of segments 2-5, the loads of step k+2 behind the last one
#define MSS_PAIR_UP(mask, n, per)                                                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                          \
    __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                                   \
  }
    constexpr int PER = (AFFINE ? 8 : 6) + (MASKED ? 2 : 0);
    ld_a(2, a_lo); ld_b(0, b_hi);
    ld_a(1, a_mid); ld_a(0, a_hi);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);                       // (as the NT kernels: +0-2 %)
    st_mask = raw_mask;
    mm(a_lo, b_hi);
    __builtin_amdgcn_sched_barrier(0);
    store_comp(buf ^ 1, 0);
    ld_b(1, b_mid);
    mm(a_mid, b_hi);
    MSS_PAIR_UP(0x2, 7, PER);
    __builtin_amdgcn_sched_barrier(0);
    store_comp(buf ^ 1, 1);
    mm(a_hi, b_hi);
    MSS_PAIR_UP(0x2, 7, PER);
    __builtin_amdgcn_sched_barrier(0);
    store_comp(buf ^ 1, 2);
    ld_b(2, b_lo);
    mm(a_hi, b_mid);
    MSS_PAIR_UP(0x2, 7, PER);
    __builtin_amdgcn_sched_barrier(0);
    store_comp(buf ^ 1, 3);
    mm(a_mid, b_mid);
    MSS_PAIR_UP(0x2, 7, PER);
    __builtin_amdgcn_sched_barrier(0);
    issue_loads();                       // K-step k+2 into the registers just drained
    advance();
    mm(a_hi, b_lo);
    MSS_PAIR_UP(0x20, 8, 1);
#undef MSS_PAIR_UP
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  };

  long long cur = ld_tile;
  setup_off(ld_tile, off, s_off, vf, row_ld, ld_nit, slab);
  setup_next();
  issue_loads();
  st_mask = raw_mask;
  finish_store(0);
  advance();
  issue_loads();
  advance();
  zero_acc();
  __syncthreads();
  int k = 0, cur_nit = ld_nit;
  auto tile_end = [&]() -> bool {
    if (++k < cur_nit) return false;
    epilogue(cur);
    cur += stride;
    if (cur >= total_tiles) return true;
    zero_acc();
    k = 0;
    cur_nit = (int)((cur / ((long long)ktiles * ctiles)) % splits) == splits - 1 ? nit_last : nit_full;
    ld_tile = cur;
    setup_next();
    return false;
  };
  while (true) {
    step(0);
    if (tile_end()) break;
    step(1);
    if (tile_end()) break;
  }
}

// fp32 weights [batch][Kpad][C] -> three bf16 planes in the LDS image order the kernel copies:
// byte ((b * Kpad/128 + n / 128) * C/16 + s) * 12288 + plane * 4096 + (n % 128) * 32 + h * 16 holds k = 16 s + 8 h .. + 7
// taps > 1 (implicit-GEMM layers: w is [taps][Kpad][C]): ONE plane set whose reduction index is tap * C + c (K-step s = tap * C/16 + ...)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int Kpad,
                                                            int C, long long w_bs, long long total, int taps) {
  const int nk = C / 16 * taps;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i & 1);
    long long r = i >> 1;
    const int s = (int)(r % nk); r /= nk;
    const int n = (int)(r % Kpad);
    const long long b = r / Kpad;
    const int tap = s / (C / 16), sc = s - tap * (C / 16);
    const float* src = w + b * w_bs + (size_t)tap * Kpad * C + (size_t)n * C + sc * 16 + h * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    unsigned hi[4], mid[4], lo[4];
    split_pair(v0.x, v0.y, hi[0], mid[0], lo[0]);
    split_pair(v0.z, v0.w, hi[1], mid[1], lo[1]);
    split_pair(v1.x, v1.y, hi[2], mid[2], lo[2]);
    split_pair(v1.z, v1.w, hi[3], mid[3], lo[3]);
    unsigned char* dst = out + ((size_t)(b * (Kpad / 128) + n / 128) * nk + s) * OPER + (n & 127) * ROW_B + h * 16;
    *reinterpret_cast<u32x4*>(dst) = u32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4*>(dst + PLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
    *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  }
}

// The ticket tile order (DYN) per instantiation: the kernels with a prologue on a 128 x 256 tile. Their static form spills 4 - 12
// registers into scratch inside the K-loop (whose loads and stores share the vector-memory counter with the tile prefetch), the
// ticket form none: 187 -> 236 TFLOP/s on 65536 x 2048 -> 4096 with BatchNorm + ReLU on the input (profiles/r05/dynamic_tiles.md).
// Everywhere else the static walk is as fast or faster (it keeps the XCD grouping of the tiles).
// The 128-wide kernel with a prologue: 175 registers in the static form (two workgroups per CU), 165 in the ticket form -- three
// workgroups per CU like the plain 128-wide kernel (the per-sample-affine variant spills there and stays as it was).
template <bool AFFINE, int BN, bool CONV, bool ROWAFF>
constexpr bool split_dyn_tiles() { return AFFINE && !CONV && !ROWAFF; }

// occupancy / CU count / the raised dynamic-LDS limit of one kernel instantiation, PER DEVICE (function attributes are per device; a
// process may drive several): filled on the first launch on that device under a mutex
struct SplitDevInfo { int occ = 0, cus = 256; };
template <typename KernelT>
int split_dev_info(KernelT kern, size_t smem, int fallback_occ, SplitDevInfo (&info)[SCHED_MAXDEV], std::mutex& mu, SplitDevInfo& out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCHED_MAXDEV) return MSS_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(mu);
  if (info[dev].occ == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) info[dev].cus = prop.multiProcessorCount;
    if (smem > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, NT, smem) != hipSuccess || n < 1) n = fallback_occ;
    info[dev].occ = n;
  }
  out = info[dev];
  return MSS_OK;
}

template <bool AFFINE, int BN, bool CONV, bool ROWAFF, bool DYN, int MF>
int launch_split_as(const MssConvArgs& p, hipStream_t stream, int* sched) {
  const int batch = p.batch > 1 ? p.batch : 1;
  const int tiles_per_batch = p.mtiles * p.ntiles;
  const long long total = (long long)tiles_per_batch * batch;
  if (total <= 0) return MSS_OK;
  const size_t smem = (size_t)2 * (1 + BN / 128) * OPER + 16;          // + the two ticket words of the dynamic tile order
  static SplitDevInfo info[SCHED_MAXDEV];
  static std::mutex mu;
  SplitDevInfo di;
  const int rc = split_dev_info(gemm_nt_bf16x3_kernel<AFFINE, BN, CONV, ROWAFF, DYN, MF>, smem, BN == 256 ? 2 : 3, info, mu, di);
  if (rc != MSS_OK) return rc;
  int per_cu_max = di.occ;
  // residency (per_cu_max or one less) whose last round of tiles is fuller, as launch_gemm (gemm.hip)
  int grid = 0;
  double best = -1.0;
  // (round 6: only when that saves more than 15 % of the rounds' slots -- one workgroup per CU leaves a single wave on each SIMD, which
  // keeps the matrix pipe 52-58 % busy against 72-79 % with two; 36 x 4096 x 512 -> 512, 2304 tiles: 192 -> 204 TFLOP/s at two per CU)
  for (int per_cu = per_cu_max; per_cu >= (per_cu_max > 1 ? per_cu_max - 1 : 1); --per_cu) {
    const long long slots = (long long)per_cu * di.cus;
    const long long g = total < slots ? total : slots;
    const long long rounds = (total + g - 1) / g;
    const double eff = (double)total / (double)(rounds * g);
    if (eff > best + (per_cu == per_cu_max ? 0.0 : 0.15)) { best = eff; grid = (int)g; }
  }
  const int group_m = MSS_ENV_INT("MSS_GEMM_GROUP_M", GEMM_GROUP_M_DEFAULT);
  const unsigned blk_bytes = (unsigned)(p.C / BK) * (CONV ? p.R * p.S : 1) * OPER;     // CONV: the taps are part of one long reduction
  hipLaunchKernelGGL((gemm_nt_bf16x3_kernel<AFFINE, BN, CONV, ROWAFF, DYN, MF>), dim3(grid), dim3(NT), smem, stream, p,
                     static_cast<const unsigned char*>(p.w_split), total, tiles_per_batch, group_m, blk_bytes, p.Kpad / 128, sched);
  return mss_launch_status();
}

// what mss_epilogue_store16's 16-byte accesses need
bool split_mf16_ok(const MssConvArgs& p) {
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.K % 4 || p.ldy % 4 || !al(p.y) || (p.batch > 1 && p.y_bs % 4)) return false;
  if (p.res && (p.ldres % 4 || !al(p.res))) return false;
  if (p.out_scale && (!al(p.out_scale) || !al(p.out_shift))) return false;
  return !p.stats || al(p.stats);
}

thread_local int split_last_mfma = 0;      // mss_gemm_split_last_mfma()

template <bool AFFINE, int BN, bool CONV, bool ROWAFF, int MF>
int launch_split_mf(const MssConvArgs& p, hipStream_t stream) {
  split_last_mfma = MF;
  if constexpr (split_dyn_tiles<AFFINE, BN, CONV, ROWAFF>()) {
    int* sched = MSS_ENV_INT("MSS_GEMM_SPLIT_STATIC", 0) ? nullptr : mss_sched_slot(stream);    // (tests: force the fallback)
    if (sched) return launch_split_as<AFFINE, BN, CONV, ROWAFF, true, MF>(p, stream, sched);
  }
  return launch_split_as<AFFINE, BN, CONV, ROWAFF, false, MF>(p, stream, nullptr);   // no slot to be had: the static walk
}

template <bool AFFINE, int BN, bool CONV = false, bool ROWAFF = false>
int launch_split(const MssConvArgs& p, hipStream_t stream) {
  // MSS_GEMM_SPLIT_MFMA=32: the round-5 form (one product per v_mfma_f32_32x32x16_bf16) for A/B and for outputs mss_epilogue_store16 cannot take
  // The 16x16x32 form takes the products WITHOUT a prologue -- the Winograd-domain batches and the plain 1x1 layers / Linears: measured
  // 1.02-1.08x the 32x32x16 form on the 128 x 256 tile, 1.0-1.07x on the 128-wide one (profiles/r06/split_mfma_ab.md). An MFMA of that
  // shape holds the SIMD's vector issue for 8 of its 16 cycles, and the BatchNorm + ReLU prologue's ten more registers put the wide
  // kernel into scratch inside the K-loop (0.96-0.99x): the prologue, implicit-GEMM and per-sample-affine kernels stay on the round-5
  // form. MSS_GEMM_SPLIT_MFMA=32: that form everywhere (A/B, tests).
  const int mf = MSS_ENV_INT("MSS_GEMM_SPLIT_MFMA", 16);
  if constexpr (!CONV && !AFFINE)
    if (mf != 32 && split_mf16_ok(p)) return launch_split_mf<AFFINE, BN, CONV, ROWAFF, 16>(p, stream);
  return launch_split_mf<AFFINE, BN, CONV, ROWAFF, 32>(p, stream);
}

}  // namespace

extern "C" int mss_gemm_split_last_mfma(void) { return split_last_mfma; }

// Shapes the split route takes: what gemm_nt_kernel's 128- / 256-wide variant-3 kernels take (p.M, p.mtiles set by the caller) with
// the weights' planes inside 32-bit byte offsets.
bool mss_gemm_nt_bf16x3_eligible(const MssConvArgs& p) {
  if (!p.w_split || p.K <= 64 || p.C / BK < 3 || p.Kpad % 128) return false;
  const long long nb = p.batch > 1 ? p.batch : 1;
  if (p.batch > 1 && p.w_bs != (long long)p.Kpad * p.C) return false;
  if ((unsigned long long)p.M * p.ldx * 4ull >= 0xffffffffull) return false;             // per batch entry (the entries' bases are 64-bit)
  if ((unsigned long long)nb * p.Kpad * p.C * 6ull >= 0xffffffffull) return false;
  return (reinterpret_cast<uintptr_t>(p.w_split) & 15) == 0;
}

// Called by mss_gemm_nt_dispatch (gemm.hip) with p.H (rows per affine group), p.mtiles set; picks the tile width as the native
// route does.
int mss_gemm_nt_bf16x3_launch(MssConvArgs p, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int bn = 0;                                  // (the A/B switch that forced one width went in round 6)
  const long long nb = p.batch > 1 ? p.batch : 1;
  const long long tiles256 = (long long)p.mtiles * (p.K / 256) * nb;
  bool wide = p.K % 256 == 0 && (bn == 256 || (bn == 0 && tiles256 >= 1024 && p.C >= 256));
  if (wide && bn == 0) {
    auto eff = [](long long total, long long slots) {
      const long long rounds = (total + slots - 1) / slots;
      return (double)total / (double)(rounds * slots);
    };
    const double ew = eff(tiles256, 512), en = eff(2 * tiles256, 768);
    if (ew < 0.8 && en > ew + 0.15) wide = false;
  }
  if (wide) {
    p.ntiles = p.K / 256;
    return p.in_scale ? launch_split<true, 256>(p, s) : launch_split<false, 256>(p, s);
  }
  p.ntiles = mss_cdiv(p.K, 128);
  return p.in_scale ? launch_split<true, 128>(p, s) : launch_split<false, 128>(p, s);
}


// ---- host side of the TN weight-gradient route ------------------------------------------------------------------------------------
void mss_wgrad_reduce_launch(const float* ws, float* dwp, long long slab4, int splits, hipStream_t stream);   // conv_igemm.hip

namespace {
struct TnSplitPlan { int ktiles, ctiles, splits, tps; long long total; };
// 512 workgroup slots (two per CU); the row range is cut so that the tiles fill whole rounds, every split a multiple of 16 rows,
// at least 48 (the loader runs two K-steps ahead) and at most M
TnSplitPlan tn_split_plan(const MssConvArgs& p) {
  TnSplitPlan pl;
  pl.ktiles = p.K / 128; pl.ctiles = p.C / 256;
  const long long base = (long long)(p.batch > 1 ? p.batch : 1) * pl.ktiles * pl.ctiles;
  const int slots = 512;
  int max_splits = p.M / 256;
  const int cap = p.batch > 1 ? 64 : 256;
  if (max_splits > cap) max_splits = cap;
  if (max_splits < 1) max_splits = 1;
  int splits = 1;
  double best = 0.0;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const long long total = base * sp;
    const double eff = (double)total / (double)(((total + slots - 1) / slots) * slots);
    if (eff > best + 1e-9) { best = eff; splits = sp; }
    if (eff >= 0.95 && total >= slots) break;
  }
  pl.tps = mss_cdiv(mss_cdiv(p.M, splits), 16) * 16;
  if (pl.tps > p.M) pl.tps = (p.M / 16) * 16;
  pl.splits = mss_cdiv(p.M, pl.tps);
  // unmasked kernel (M % 16 == 0): the last split is shorter; it needs its three K-steps too (the loader runs two ahead)
  if (p.M % 16 == 0 && pl.splits > 1 && p.M - (pl.splits - 1) * pl.tps < 48) {
    pl.splits -= 1;
    pl.tps = mss_cdiv(mss_cdiv(p.M, pl.splits), 16) * 16;
    pl.splits = mss_cdiv(p.M, pl.tps);
  }
  pl.total = base * pl.splits;
  return pl;
}
}  // namespace

// p as mss_conv2d_wgrad_f32 sees it (p.M set); lddy = row stride of dy
bool mss_wgrad_tn_bf16x3_eligible(const MssConvArgs& p, int lddy) {
  if (p.route != 1 || p.R * p.S != 1 || p.K % 128 || p.C % 256 || p.K < 128 || p.ldx != p.C || lddy < p.K || lddy % 4 || p.M < 64) return false;
  if (p.batch > 1 && (lddy != p.K || p.x_bs % 4 || p.y_bs % 4 || p.N != 1 || p.H != 1)) return false;
  if (p.batch <= 1 && (p.stride != 1 || p.pad != 0 || p.OH != p.H || p.OW != p.W)) return false;
  if ((p.in_scale || p.in_shift || p.in_relu) && (p.batch > 1 || p.in_ss_stride != 0 || !p.in_scale || !p.in_shift)) return false;
  if (p.in_scale && ((reinterpret_cast<uintptr_t>(p.in_scale) | reinterpret_cast<uintptr_t>(p.in_shift)) & 15)) return false;
  if ((unsigned long long)p.M * p.C * 4ull >= 0xffffffffull || (unsigned long long)p.M * lddy * 4ull >= 0xffffffffull) return false;   // per position
  return tn_split_plan(p).total >= 256;          // small products (fewer tiles than half the workgroup slots) stay on the native one-wave kernels
}

long long mss_wgrad_tn_bf16x3_ws_bytes(const MssConvArgs& p, int Cp) {
  const TnSplitPlan pl = tn_split_plan(p);
  return pl.splits > 1 ? (long long)pl.splits * (p.batch > 1 ? p.batch : 1) * p.Kpad * Cp * 4 : 0;
}

int mss_wgrad_tn_bf16x3_launch(const MssConvArgs& p, const float* dy, int lddy, float* dwp, int Cp, float* ws, long long ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const TnSplitPlan pl = tn_split_plan(p);
  const int P = p.batch > 1 ? p.batch : 1;
  if (p.Kpad != p.K || Cp != p.C) return MSS_ERR_BAD_ARG;          // whole tiles: no padding rows / columns to clear
  const long long slab = (long long)P * p.Kpad * Cp;
  if (pl.splits > 1 && (!ws || ws_bytes < (long long)pl.splits * slab * 4)) return MSS_ERR_BAD_ARG;
  float* out = pl.splits > 1 ? ws : dwp;
  const size_t smem = (size_t)2 * 3 * OPER + 16;
  static bool attr[SCHED_MAXDEV] = {false};                           // function attributes are per device
  static std::mutex attr_mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCHED_MAXDEV) return MSS_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> attr_lock(attr_mu);
  if (!attr[dev]) {
    const void* ks[4] = {reinterpret_cast<const void*>(gemm_tn_bf16x3_kernel<false, false>), reinterpret_cast<const void*>(gemm_tn_bf16x3_kernel<false, true>),
                         reinterpret_cast<const void*>(gemm_tn_bf16x3_kernel<true, false>), reinterpret_cast<const void*>(gemm_tn_bf16x3_kernel<true, true>)};
    for (const void* kf : ks) {
      hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    attr[dev] = true;
  }
  const long long slots = 512;
  const int grid = (int)(pl.total < slots ? pl.total : slots);
  const long long a_bs = p.batch > 1 ? p.y_bs : 0, b_bs = p.batch > 1 ? p.x_bs : 0;
  const bool masked = p.M % 16 != 0 || (pl.splits > 1 && p.M - (pl.splits - 1) * pl.tps < 48);
#define TN_LAUNCH(AFF, MSK, SC, SH, RL)                                                                                                      \
  hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<AFF, MSK>), dim3(grid), dim3(NT), smem, s, dy, lddy, p.x, p.C, out, P, p.M, a_bs, b_bs, p.Kpad, Cp, \
                     pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total, SC, SH, RL)
  if (p.in_scale) {
    if (masked) TN_LAUNCH(true, true, p.in_scale, p.in_shift, p.in_relu); else TN_LAUNCH(true, false, p.in_scale, p.in_shift, p.in_relu);
  } else {
    if (masked) TN_LAUNCH(false, true, (const float*)nullptr, (const float*)nullptr, 0); else TN_LAUNCH(false, false, (const float*)nullptr, (const float*)nullptr, 0);
  }
#undef TN_LAUNCH
  if (pl.splits > 1) mss_wgrad_reduce_launch(ws, dwp, slab / 4, pl.splits, s);
  return mss_launch_status();
}

static bool rowaff_eligible(const MssConvArgs& p);
// The implicit-GEMM layers on the split route: what conv_igemm_kernel takes with > 64 output channels and ONE prologue affine
bool mss_conv_bf16x3_eligible(const MssConvArgs& p) {
  if (rowaff_eligible(p)) return true;
  if (!p.w_split || p.K <= 64 || p.Kpad % 128 || p.C % BK || p.batch > 1 || p.res_mask) return false;
  const int taps = p.R * p.S;
  if ((p.S != 1 && p.S != 3) || taps > 9 || taps * (p.C / BK) < 3) return false;
  if (taps == 1 && p.stride == 1 && p.pad == 0) return false;                       // a plain GEMM: the NT route takes it
  if (p.in_scale && p.in_ss_stride) return false;                                   // per-sample affines (Dropout2d fold): native
  if (p.in_relu && !p.in_scale) return false;
  if ((unsigned long long)p.N * p.H * p.W * p.ldx * 4ull >= 0x7fffffffull) return false;   // signed 32-bit pixel offsets
  if ((unsigned long long)taps * p.Kpad * p.C * 6ull >= 0xffffffffull) return false;
  return (reinterpret_cast<uintptr_t>(p.w_split) & 15) == 0;
}

// 1x1 / stride-1 layers whose prologue affine is per sample while 128-row tiles straddle images ((OH * OW) % 128 != 0: mod6 / mod7's
// last convolutions at 16 x 700 x 700) -- the one GEMM shape mss_gemm_nt_dispatch leaves to conv_igemm_kernel's PER_SAMPLE path
static bool rowaff_eligible(const MssConvArgs& p) {
  if (!p.w_split || p.R * p.S != 1 || p.stride != 1 || p.pad != 0 || p.H != p.OH || p.W != p.OW || p.batch > 1) return false;
  if (!p.in_scale || !p.in_ss_stride || (p.OH * p.OW) % BM == 0 || p.K <= 64 || p.Kpad % 128 || p.C % BK || p.C / BK < 3) return false;
  if ((unsigned long long)p.M * p.ldx * 4ull >= 0xffffffffull || (unsigned long long)p.Kpad * p.C * 6ull >= 0xffffffffull) return false;
  return (reinterpret_cast<uintptr_t>(p.w_split) & 15) == 0;
}

int mss_conv_bf16x3_launch(MssConvArgs p, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  p.mtiles = mss_cdiv(p.M, BM);
  if (rowaff_eligible(p)) {
    p.H = p.OH * p.OW;                                   // rows per image
    p.ntiles = mss_cdiv(p.K, 128);
    return launch_split<true, 128, false, true>(p, s);
  }
  const long long tiles256 = (long long)p.mtiles * (p.K / 256);
  const bool wide = p.K % 256 == 0 && tiles256 >= 1024;
  if (wide) {
    p.ntiles = p.K / 256;
    return p.in_scale ? launch_split<true, 256, true>(p, s) : launch_split<false, 256, true>(p, s);
  }
  p.ntiles = mss_cdiv(p.K, 128);
  return p.in_scale ? launch_split<true, 128, true>(p, s) : launch_split<false, 128, true>(p, s);
}

extern "C" long long mss_gemm_split_weights_bytes(int batch, int Kpad, int C) {
  if (batch < 1 || Kpad < 128 || Kpad % 128 || C < 16 || C % 16) return 0;
  return (long long)batch * Kpad * C * 6;
}

static int split_weights(const float* w, void* planes, int batch, int taps, int Kpad, int C, long long w_bs, void* stream) {
  if (!w || !planes || batch < 1 || taps < 1 || Kpad % 128 || C % 16 || Kpad < 128 || C < 16) return MSS_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(planes)) & 15 || w_bs % 4) return MSS_ERR_BAD_ARG;
  mss_sched_init(static_cast<hipStream_t>(stream));     // the ticket pool exists before anything can launch on these planes
  const long long total = (long long)batch * Kpad * (C / 16) * taps * 2;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(split_weights_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     static_cast<unsigned char*>(planes), Kpad, C, w_bs, total, taps);
  return mss_launch_status();
}
extern "C" int mss_gemm_split_weights_bf16x3(const float* w, void* planes, int batch, int Kpad, int C, long long w_bs, void* stream) {
  return split_weights(w, planes, batch, 1, Kpad, C, w_bs, stream);
}
// w [taps][Kpad][C] (mss_conv2d_pack_weights_f32 with R * S = taps > 1) -> planes for MssConvArgs.w_split of an implicit-GEMM layer:
// taps * Kpad * C * 6 bytes, the taps folded into one reduction of taps * C
extern "C" int mss_conv_split_weights_bf16x3(const float* w, void* planes, int taps, int Kpad, int C, void* stream) {
  return split_weights(w, planes, 1, taps, Kpad, C, (long long)taps * Kpad * C, stream);
}
