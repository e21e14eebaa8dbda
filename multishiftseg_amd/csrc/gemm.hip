// Persistent NT GEMM on the fp32 matrix cores: Y[b][m][n] = sum_c act(X[b][m][c]) * W[b][n][c]  (+ epilogue).
//
// This is the 1x1 / stride-1 / padding-0 case of conv_igemm.hip without the implicit-GEMM machinery (no taps, no
// pixel decode, no dead-tap mask): the batched Winograd-domain products (winograd.hip; 36 GEMMs of
// [tiles x C] x [C x K] per layer), mask prediction of the M2F head, and the pre-activation 1x1 convolutions of the
// WideResNet trunk (wider_resnet.py:121-131,157) and the heads (deepv3.py:66,235-252).
//
// The inner loop is conv_igemm's (128x128x16 tile, 4 waves of 64x64, LDS double buffer with +4-float row padding,
// fragments double-buffered in registers, one barrier per K-step). What is new is the schedule around it: a
// workgroup is PERSISTENT and walks tiles t, t + grid, t + 2 grid, ...; the loader runs one K-step ahead ACROSS tile
// boundaries, so while the last K-step of a tile multiplies, the first K-step of the next tile is already on its way
// from HBM and lands in the other LDS buffer; the epilogue's stores are issued and the MFMAs of the next tile start
// right behind them. With short reductions (C = 128..512, i.e. 8..32 K-steps per tile) the start-up/drain of every
// tile was ~5 K-steps' worth of time in the one-tile-per-workgroup kernel.
//
// Tile order is n-fastest and an XCD owns a contiguous run of tiles, so the n-tiles that share an X tile hit in the
// same L2.
#include "mss_epilogue.h"
#include "mss_gemm_tiles.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256, BM = 128, BK = 16;
// LDS rows are 16 floats with NO padding; the four 16-byte chunks of row r are rotated by (r >> 2): a staging write
// (16 lanes = 4 rows x 4 chunks) and a fragment read (16 lanes = 16 rows x 1 chunk) then both touch 16 disjoint
// 4-bank spans. (The +4-float padding of conv_igemm is conflict-free for the reads only: PMC showed a third of the
// LDS cycles as bank conflicts from the ds_write_b128 side.)
constexpr int LDK = BK;
constexpr int WTM = 64, TM = 2;
constexpr int CPR = BK / 4;            // float4 chunks per tile row
constexpr int RPP = NT / CPR;          // rows staged per pass
constexpr int A_LD = BM / RPP;
constexpr int NKC = BK / 8;

static_assert(NKC == 2, "the step body below is written for two 8-k chunks");

// VARIANT (A/B, MSS_GEMM_VARIANT): 0 = loads for K-step k+1 issued at the top of step k and written to LDS in the same step
// (round 1); 1 = the same with the issue pinned at the loop top (the compiler otherwise sinks the four global loads behind
// the first 16 MFMAs and the LDS reads, 9 MFMAs ahead of their first use); 2 = the loader runs TWO steps ahead with one
// register set: step k first stores the registers (step k+1's data, requested a whole step ago) to LDS and immediately
// re-issues them for step k+2, so no wave waits on a load it has just issued.
// BN: output-channel extent of a tile. 128 (2x2 waves of 64x64, 4 workgroups per CU) or 256 (2x2 waves of 64x128: 64 MFMAs
// per wave and barrier instead of 32, a quarter less operand traffic per FLOP and half the per-tile prologue/epilogue
// share, at 2 workgroups per CU).
template <bool AFFINE, int VARIANT, int BN>
__global__ __launch_bounds__(NT, BN == 256 ? 2 : 3) void gemm_nt_kernel(MssConvArgs p, long long first_tile, long long total_tiles,
                                                                        int tiles_per_batch, int group_m) {
  constexpr int WTN = BN / 2, TN = WTN / 32, B_LD = BN / RPP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDK]
  float* Bs = As + 2 * BM * LDK;          // [2][BN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int chunk = tid % CPR, row0 = tid / CPR;
  const int wchunk = (chunk + (row0 >> 2)) & 3;          // RPP = 64 rows per pass: (row0 + j * 64) >> 2 has the same low bits
  const int n_it = p.C / BK;
  const long long stride = gridDim.x;
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();

  // ---- loader state: the K-step the next issue_loads() fetches ----
  // VARIANT 3 (r03): 32-bit byte offsets from the (uniform) operand bases instead of 64-bit pointers, the offsets of the
  // workgroup's NEXT tile kept beside the current ones, and a branch-free advance() (a select on "this was the tile's last
  // K-step"): the K-step body becomes ONE basic block, which is what lets the requests below interleave the loader's
  // instructions with the MFMAs instead of leaving them in a clump in front of them (a 32x32x2 f32 MFMA holds the pipe for
  // 64 cycles; in variant 2 this wave issues ~45 other instructions with no MFMA in flight every K-step).
  const float* a_ptr[A_LD];
  const float* b_ptr[B_LD];
  unsigned a_off[A_LD], b_off[B_LD], a_nxt[A_LD], b_nxt[B_LD], s_off = 0, s_nxt = 0;
  const float* s_ptr = p.in_scale;
  const float* h_ptr = p.in_shift;
  long long ld_tile = first_tile + mss_xcd_remap(blockIdx.x, gridDim.x);   // this launch walks tiles [first_tile, total_tiles)
  int ld_k = 0;
  auto setup_off = [&](long long t, unsigned* ao, unsigned* bo, unsigned& so) {
    const int b = (int)(t / tiles_per_batch);
    const int v = (int)(t - (long long)b * tiles_per_batch);
    int mt, nt; mss_tile_mn(v, p.mtiles, p.ntiles, group_m, mt, nt);
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int row = mt * BM + row0 + j * RPP;
      row = row < p.M ? row : p.M - 1;
      ao[j] = (unsigned)(((size_t)b * p.x_bs + (size_t)row * p.ldx + chunk * 4) * sizeof(float));
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      bo[j] = (unsigned)(((size_t)b * p.w_bs + (size_t)(nt * BN + row0 + j * RPP) * p.C + chunk * 4) * sizeof(float));
    if (AFFINE) so = (unsigned)(((size_t)((mt * BM) / p.H) * p.in_ss_stride + chunk * 4) * sizeof(float));
  };
  auto setup = [&](long long t) {
    if (VARIANT == 3) { setup_off(t, a_off, b_off, s_off); return; }
    const int b = (int)(t / tiles_per_batch);
    const int v = (int)(t - (long long)b * tiles_per_batch);
    int mt, nt; mss_tile_mn(v, p.mtiles, p.ntiles, group_m, mt, nt);
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int row = mt * BM + row0 + j * RPP;
      row = row < p.M ? row : p.M - 1;                  // rows past the end re-read the last row; never stored
      a_ptr[j] = p.x + (size_t)b * p.x_bs + (size_t)row * p.ldx + chunk * 4;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      b_ptr[j] = p.w + (size_t)b * p.w_bs + (size_t)(nt * BN + row0 + j * RPP) * p.C + chunk * 4;
    if (AFFINE) {
      // one affine per tile: per-sample affines (Dropout2d fold) are only routed here when tiles cannot straddle images
      const size_t so = (size_t)((mt * BM) / p.H) * p.in_ss_stride + chunk * 4;     // p.H = rows per image in this mode
      s_ptr = p.in_scale + so;
      h_ptr = p.in_shift + so;
    }
  };
  auto setup_next = [&]() {              // offsets of the tile the loader enters after its current one (past the end: the same)
    const long long t = ld_tile + stride;
    setup_off(t < total_tiles ? t : ld_tile, a_nxt, b_nxt, s_nxt);
  };
  f32x4 areg[A_LD], breg[B_LD], sreg, hreg;
  auto issue_loads = [&]() {
    if (VARIANT == 3) {
      const char* xb = reinterpret_cast<const char*>(p.x);
      const char* wb = reinterpret_cast<const char*>(p.w);
#pragma unroll
      for (int j = 0; j < A_LD; ++j) areg[j] = *reinterpret_cast<const f32x4*>(xb + a_off[j]);
#pragma unroll
      for (int j = 0; j < B_LD; ++j) breg[j] = *reinterpret_cast<const f32x4*>(wb + b_off[j]);
      if (AFFINE) {
        sreg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in_scale) + s_off);
        hreg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in_shift) + s_off);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < A_LD; ++j) areg[j] = *reinterpret_cast<const f32x4*>(a_ptr[j]);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) breg[j] = *reinterpret_cast<const f32x4*>(b_ptr[j]);
    if (AFFINE) {
      sreg = *reinterpret_cast<const f32x4*>(s_ptr);
      hreg = *reinterpret_cast<const f32x4*>(h_ptr);
    }
  };
  auto advance = [&]() {                 // next K-step of this tile, else first K-step of this workgroup's next tile
    if (VARIANT == 3) {
      const bool wrap = ++ld_k == n_it;
#pragma unroll
      for (int j = 0; j < A_LD; ++j) a_off[j] = wrap ? a_nxt[j] : a_off[j] + BK * (unsigned)sizeof(float);
#pragma unroll
      for (int j = 0; j < B_LD; ++j) b_off[j] = wrap ? b_nxt[j] : b_off[j] + BK * (unsigned)sizeof(float);
      if (AFFINE) s_off = wrap ? s_nxt : s_off + BK * (unsigned)sizeof(float);
      ld_k = wrap ? 0 : ld_k;
      return;
    }
    if (++ld_k < n_it) {
#pragma unroll
      for (int j = 0; j < A_LD; ++j) a_ptr[j] += BK;
#pragma unroll
      for (int j = 0; j < B_LD; ++j) b_ptr[j] += BK;
      if (AFFINE) { s_ptr += BK; h_ptr += BK; }
    } else {
      ld_k = 0;
      ld_tile += stride;
      setup(ld_tile < total_tiles ? ld_tile : ld_tile - stride);   // past the end: re-read the last tile, never used
    }
  };
  auto finish_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      f32x4 val = areg[j];
      if (AFFINE) {
        val = val * sreg + hreg;
        val.x = fmaxf(val.x, relu_floor); val.y = fmaxf(val.y, relu_floor);
        val.z = fmaxf(val.z, relu_floor); val.w = fmaxf(val.w, relu_floor);
      }
      *reinterpret_cast<f32x4*>(&As[(buf * BM + row0 + j * RPP) * LDK + wchunk * 4]) = val;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      *reinterpret_cast<f32x4*>(&Bs[(buf * BN + row0 + j * RPP) * LDK + wchunk * 4]) = breg[j];
  };

  const int frag_row = lane & 31, frag_h = lane >> 5;
  const int rot = frag_row >> 2;                           // tile rows are frag_row + multiples of 32: same rotation
  const float* Abase = &As[(wm * WTM + frag_row) * LDK];
  const float* Bbase = &Bs[(wn * WTN + frag_row) * LDK];
  const int koff[2] = {((frag_h + rot) & 3) * 4, ((2 + frag_h + rot) & 3) * 4};   // logical chunk kc * 2 + frag_h
  f32x4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](int set, int buf, int kc) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[set][i] = *reinterpret_cast<const f32x4*>(Abase + (buf * BM + i * 32) * LDK + koff[kc]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[set][j] = *reinterpret_cast<const f32x4*>(Bbase + (buf * BN + j * 32) * LDK + koff[kc]);
  };
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto mfma_chunk = [&](int set) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][s], fb[set][j][s], acc[i][j], 0, 0, 0);
  };
  auto epilogue = [&](long long t) {
    const int b = (int)(t / tiles_per_batch);
    const int v = (int)(t - (long long)b * tiles_per_batch);
    int mt, nt; mss_tile_mn(v, p.mtiles, p.ntiles, group_m, mt, nt);
    mss_epilogue_store<TM, TN>(acc, p, p.y + (size_t)b * p.y_bs, mt * BM + wm * WTM, nt * BN + wn * WTN, lane);
  };

  long long cur = ld_tile;               // tile being multiplied (the launch guarantees cur < total_tiles)
  setup(ld_tile);
  if (VARIANT == 3) setup_next();
  issue_loads();
  finish_store(0);
  advance();
  if (VARIANT >= 2) { issue_loads(); advance(); }      // registers now hold K-step 1
  zero_acc();
  __syncthreads();
  load_frags(0, 0, 0);
  int buf = 0, k = 0;
  while (true) {
    if (VARIANT >= 2) {
      load_frags(1, buf, 1);
      finish_store(buf ^ 1);             // K-step k+1, requested during step k-1
      issue_loads();                     // K-step k+2 (possibly of the next tile) into the registers just drained
      advance();
      mfma_chunk(0);
      if (VARIANT == 3) {
        // one fragment read / LDS write / global load / pair of VALU-SALU behind each of the first MFMAs of the chunk
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
        for (int i = 0; i < A_LD + B_LD; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
        for (int i = 0; i < A_LD + B_LD; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
#pragma unroll
        for (int i = 0; i < 4 * TM * TN - (TM + TN) - 2 * (A_LD + B_LD); ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x6, 2, 0); }
      }
      __syncthreads();
      load_frags(0, buf ^ 1, 0);
      mfma_chunk(1);
    } else {
      issue_loads();                       // K-step k+1 of this tile, or K-step 0 of the next one
      if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
      load_frags(1, buf, 1);
      finish_store(buf ^ 1);
      mfma_chunk(0);
      __syncthreads();
      load_frags(0, buf ^ 1, 0);
      mfma_chunk(1);
      advance();
    }
    buf ^= 1;
    if (++k == n_it) {
      epilogue(cur);
      cur += stride;
      if (cur >= total_tiles) break;
      zero_acc();
      k = 0;
      if (VARIANT == 3) {                // the loader entered tile `cur` at least one K-step ago (n_it >= 3): prepare the one after it
        ld_tile = cur;
        setup_next();
      }
    }
  }
}

// A handful of rows (ASPP's image-pooling branch: [N, 4096] x [4096 -> 256], deepv3.py:84-88): one MFMA tile would walk the whole
// reduction alone (256 K-steps on one or two of 256 CUs: 0.25 ms for 4 MB of weights). Here a WAVE owns one output channel: its
// weight row streams through the lanes in 16-byte pieces, the <= 8 input rows come from cache, one wave reduction per row.
template <int MAXM>
__global__ __launch_bounds__(256) void gemm_few_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            float* __restrict__ y, int ldy, int M, int C, int K) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  float acc[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) acc[m] = 0.f;
  const float* wr = w + (size_t)k * C;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
    for (int m = 0; m < MAXM; ++m)
      if (m < M) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * ldx + c);
        acc[m] += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
      }
  }
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    const float s = mss_wave_sum(acc[m]);
    if (m < M && lane == 0) y[(size_t)m * ldy + k] = s;
  }
}

template <bool AFFINE, int VARIANT, int BN>
int launch_gemm(const MssConvArgs& p, hipStream_t stream, long long first = 0, long long end = -1) {
  const int batch = p.batch > 1 ? p.batch : 1;
  const int tiles_per_batch = p.mtiles * p.ntiles;
  if (end < 0) end = (long long)tiles_per_batch * batch;
  const long long total = end - first;                 // tiles of this launch
  if (total <= 0) return MSS_OK;
  const size_t smem = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  static int per_cu_max = 0, cus = 256;  // resident workgroups per CU (BN = 128: 4 with 32 KB LDS and <= 128 registers); one static per instantiation
  if (per_cu_max == 0) {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_nt_kernel<AFFINE, VARIANT, BN>, NT, smem) != hipSuccess || n < 1) n = 3;
    per_cu_max = n;
  }
  // Every workgroup walks ceil(total / grid) tiles: pick the residency (per_cu_max or one less) whose last round is
  // fuller, e.g. 4608 tiles = 6 full rounds of 768 but 4.5 rounds of 1024.
  int grid = 0;
  double best = -1.0;
  for (int per_cu = per_cu_max; per_cu >= (per_cu_max > 1 ? per_cu_max - 1 : 1); --per_cu) {
    const long long slots = (long long)per_cu * cus;
    const long long g = total < slots ? total : slots;
    const long long rounds = (total + g - 1) / g;
    const double eff = (double)total / (double)(rounds * g);
    if (eff > best + 0.02) { best = eff; grid = (int)g; }     // (r06: re-measured, 36 x 4096 x 512 -> 512 at one per CU 133.5 vs 125 TFLOP/s forced to two; the split kernel differs)
  }
  // the square-ish tile order only for whole launches: the hybrid wide + narrow split (MSS_GEMM_TAIL) relies on tile ranges
  const int group_m = (first == 0 && end == (long long)tiles_per_batch * batch) ? MSS_ENV_INT("MSS_GEMM_GROUP_M", GEMM_GROUP_M_DEFAULT) : 0;
  hipLaunchKernelGGL((gemm_nt_kernel<AFFINE, VARIANT, BN>), dim3(grid), dim3(NT), smem, stream, p, first, end, tiles_per_batch, group_m);
  return mss_launch_status();
}

}  // namespace

bool mss_gemm_nt_bf16x3_eligible(const MssConvArgs& p);       // gemm_bf16x3.hip: the split-bf16 route (MssConvArgs.w_split)
int mss_gemm_nt_bf16x3_launch(MssConvArgs p, void* stream);

// Shapes this kernel takes from mss_conv2d_forward_f32 (conv_igemm.hip); p.M is set.
// 33..64 output channels over many rows (the 48-channel heads and bot_fine, 1 M pixels) on the persistent kernel with a 128 x 64
// tile instead of conv_igemm's one-tile-per-workgroup 256 x 64 kernel: 0.203 -> 0.184 ms (256 -> 48) and 0.116 -> 0.087 ms
// (128 -> 48) at 1 x 512 x 1024
static bool gemm_bn64_wanted(const MssConvArgs& p) {
  // r05: from K >= 32 (33 before): 256 -> 32 over 162 k rows (the tail of the pixel decoder's
  // 288-wide projection, 31 TFLOP/s on conv_igemm's 256 x 64 tile)
  // r06: batched products too (the 48-channel tail of the 304-wide Winograd-domain data gradient of final.0: 64 x 29412 x 256 -> 48,
  // conv_igemm's one-tile-per-workgroup kernel streamed its 1.9 GB of X' at 2.6 TB/s: 2.58 -> 2.47 ms native, 1.92 -> 1.75 ms on the
  // split route for the whole 304-wide product)
  return p.K <= 64 && p.K >= 32 && p.C / BK >= 3 && p.M >= 16384;
}

bool mss_gemm_nt_eligible(const MssConvArgs& p) {
  if (p.R * p.S != 1 || p.stride != 1 || p.pad != 0 || p.H != p.OH || p.W != p.OW) return false;
  if ((p.K <= 64 && !gemm_bn64_wanted(p)) || p.C % BK || p.C < 2 * BK) return false;        // narrow outputs stay on the 256x64 tile
  if (p.in_relu && !p.in_scale) return false;                      // ReLU without affine: not a shape this path sees
  if (p.in_scale && p.in_ss_stride && (p.OH * p.OW) % BM) return false;   // per-sample affine: tiles must not straddle images
  return true;
}

// Returns -1 when the shape is not handled here (the implicit-GEMM kernel takes it).
// 1x1 layers over <= 32 pixels with nothing fused (the ASPP image-pooling product: one row per image -- 16 at 16 x 768 x 768, where
// the MFMA tile path took 0.255 ms): gemm_few_rows_kernel (p.M is set)
bool mss_gemm_few_rows(const MssConvArgs& p) {
  return p.R * p.S == 1 && p.stride == 1 && p.pad == 0 && p.H == p.OH && p.W == p.OW && p.M > 0 && p.M <= 32 && p.batch <= 1 &&
         p.C % 4 == 0 && p.ldx % 4 == 0 && !p.in_scale && !p.in_relu && !p.out_scale && !p.out_relu && !p.res && !p.stats &&
         ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w)) & 15) == 0;
}

int mss_gemm_nt_dispatch(MssConvArgs p, void* stream) {
  if (mss_gemm_few_rows(p)) {
    const dim3 grid((p.K + 3) / 4);
    hipStream_t fs = static_cast<hipStream_t>(stream);
    if (p.M <= 8) hipLaunchKernelGGL(gemm_few_rows_kernel<8>, grid, dim3(256), 0, fs, p.x, p.ldx, p.w, p.y, p.ldy, p.M, p.C, p.K);
    else if (p.M <= 16) hipLaunchKernelGGL(gemm_few_rows_kernel<16>, grid, dim3(256), 0, fs, p.x, p.ldx, p.w, p.y, p.ldy, p.M, p.C, p.K);
    else hipLaunchKernelGGL(gemm_few_rows_kernel<32>, grid, dim3(256), 0, fs, p.x, p.ldx, p.w, p.y, p.ldy, p.M, p.C, p.K);
    return mss_launch_status();
  }
  if (!mss_gemm_nt_eligible(p)) return -1;
  p.H = (p.in_scale && p.in_ss_stride) ? p.OH * p.OW : (p.M > 0 ? p.M : 1);   // rows per affine group
  p.mtiles = mss_cdiv(p.M, BM);
  if (p.K <= 64) {                                       // (experiment, see gemm_bn64_wanted)
    p.ntiles = 1;
    if (p.Kpad < 64) return MSS_ERR_BAD_ARG;
    const long long nb = p.batch > 1 ? p.batch : 1;       // variant 3: 32-bit byte offsets over the whole batch
    const long long span_x = ((nb - 1) * p.x_bs + (long long)p.M * p.ldx) * 4, span_w = ((nb - 1) * p.w_bs + (long long)p.Kpad * p.C) * 4;
    if (span_x >= 0xffffffffll || span_w >= 0xffffffffll) return -1;
    return p.in_scale ? launch_gemm<true, 3, 64>(p, static_cast<hipStream_t>(stream)) : launch_gemm<false, 3, 64>(p, static_cast<hipStream_t>(stream));
  }
  p.ntiles = mss_cdiv(p.K, 128);
  if (p.Kpad < p.ntiles * 128) return MSS_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mss_gemm_nt_bf16x3_eligible(p)) return mss_gemm_nt_bf16x3_launch(p, stream);      // the split-bf16 route, chosen by the caller
  // measured (tools/bench_bgemm.py, tools/bench_1x1.py, r02): variant 2 is +3-7 % on every shape (C = 128: 82.7 -> 86.9,
  // 512: 127 -> 132, 1024 -> 2048: 132 -> 136 TFLOP/s; 1x1 2048 -> 4096 with prologue/residual/statistics: 123 -> 128)
  // r03: variant 3 (32-bit offsets, branch-free advance, loader instructions interleaved with the MFMAs) is +4-6 % from C = 256 up
  // (36 x 16384 x 256 -> 256: 125 -> 130, 512 -> 512: 128 -> 136, 1024 -> 2048: 133 -> 140, 1x1 2048 -> 4096: 139 -> 146 TFLOP/s) and
  // +0.5 % at C = 128, bit-identical results (tools/bench_gemm_variant.py, alternating A/B); shapes it does not take fall back to 2
  int variant = 3;
  variant = MSS_ENV_INT("MSS_GEMM_VARIANT", 3);
  if (variant == 1) return p.in_scale ? launch_gemm<true, 1, 128>(p, s) : launch_gemm<false, 1, 128>(p, s);
  // variant 3 addresses its operands with 32-bit byte offsets and needs >= 3 K-steps per tile (see the kernel)
  const long long nb3 = p.batch > 1 ? p.batch : 1;
  const bool v3 = variant == 3 && p.C / BK >= 3 &&
                  (unsigned long long)((nb3 - 1) * p.x_bs + (long long)p.M * p.ldx) * 4ull < 0xffffffffull &&
                  (unsigned long long)((nb3 - 1) * p.w_bs + (long long)p.Kpad * p.C) * 4ull < 0xffffffffull;
  if (variant == 0) return p.in_scale ? launch_gemm<true, 0, 128>(p, s) : launch_gemm<false, 0, 128>(p, s);
  // 256-wide tiles when the output channels split evenly, the reduction is long enough and there is work for two rounds of
  // the 512 slots. Measured (tools/bench_bgemm.py, bench_1x1.py): 1x1 2048 -> 4096
  // 127 -> 133, ASPP 4096 -> 256 123 -> 131, 1024 -> 2048 136 -> 138, C = 304 122 -> 126 TFLOP/s; C = 256: 122 -> 121 (not taken).
  constexpr int bn = 0;                                  // (the A/B switch that forced one width went in round 6)
  const long long tiles256 = (long long)p.mtiles * (p.K / 256) * (p.batch > 1 ? p.batch : 1);
  bool wide = p.K % 256 == 0 && (bn == 256 || (bn == 0 && tiles256 >= 1024 && p.C >= 256));
  if (wide && bn == 0) {
    // ... unless the last round of wide tiles is mostly idle while the narrow tiles fill theirs: 64 x 18 x 1 wide tiles (ASPP
    // dilation 12 / 24 through F(6x6): 2304 tiles x 4096 -> 256) are 2.25 rounds of the 512 slots but exactly 3 rounds of the
    // 768 narrow slots -- measured 119.5 (wide) against 133.4 TFLOP/s (narrow)
    auto eff = [](long long total, long long slots) {
      const long long rounds = (total + slots - 1) / slots;
      return (double)total / (double)(rounds * slots);
    };
    const double ew = eff(tiles256, 512), en = eff(2 * tiles256, 768);
    if (ew < 0.8 && en > ew + 0.15) wide = false;
  }
  if (wide) {
    // Hybrid last round, OPT-IN (MSS_GEMM_TAIL=1): when the wide tiles leave a partial last round that is at most 3/4
    // full, the whole rounds run on wide tiles and the remainder as twice as many narrow tiles in a second launch --
    // narrow tile 2w + {0, 1} is wide tile w's left / right half in the same n-fastest order -- e.g. 64 x 15 x 2 = 1920
    // wide tiles (mod4 through F(6x6)): 3 wide rounds + exactly one round of 768 narrow tiles instead of 3.75 -> 4.
    // Bit-identical results. Measured: isolated 0.582 -> 0.558 ms (64 x 1892 x 512 -> 512) and 1.172 -> 1.129 ms (64 x 2112 x
    // 512 -> 1024); in the step 67.3 -> 67.1 ms of gemm_nt (the narrow round is slower per FLOP and costs a second launch).
    // Off by default: 0.2 ms per step does not pay for one GEMM call becoming two kernel launches in every profile.
    // r03: a default rule for nearly-empty last rounds (rem <= 128; only one-image eval products qualify) measured +-0 on the eval
    // forward (42.15 -> 42.15 ms): it stays opt-in. The two launches use the variant-3 kernels where the shape allows.
    // r04 default rule (-1): single-position products with at most four whole rounds, where the idle part of the last round is a
    // visible share of the launch -- the pixel decoder's Linears over 16 x 10 164 tokens are 1271 wide tiles = 2.48 rounds: decoder
    // forward + backward 66.15 -> 65.78 ms with the hybrid, bit-identical; no product of the DeepLab step or of its eval forward
    // qualifies (their tile counts are whole rounds), so the per-launch accounting of bench.py is unchanged. 0: never, 1: always.
    const int tail_mode = MSS_ENV_INT("MSS_GEMM_TAIL", -1);
    const long long full = (tiles256 / 512) * 512, rem = tiles256 - full;
    const int nw = p.K / 256;
    const long long rem_max = tail_mode == 1 ? 384 : (tail_mode == -1 && p.batch <= 1 && full <= 4 * 512) ? 384 : 0;
    if (bn == 0 && full > 0 && rem > 0 && rem <= rem_max) {
      MssConvArgs q = p;
      q.ntiles = nw;
      int rc = v3 ? (p.in_scale ? launch_gemm<true, 3, 256>(q, s, 0, full) : launch_gemm<false, 3, 256>(q, s, 0, full))
                  : (p.in_scale ? launch_gemm<true, 2, 256>(q, s, 0, full) : launch_gemm<false, 2, 256>(q, s, 0, full));
      if (rc) return rc;
      q.ntiles = 2 * nw;
      if (v3) return p.in_scale ? launch_gemm<true, 3, 128>(q, s, 2 * full, 2 * tiles256) : launch_gemm<false, 3, 128>(q, s, 2 * full, 2 * tiles256);
      return p.in_scale ? launch_gemm<true, 2, 128>(q, s, 2 * full, 2 * tiles256) : launch_gemm<false, 2, 128>(q, s, 2 * full, 2 * tiles256);
    }
    p.ntiles = nw;
    if (v3) return p.in_scale ? launch_gemm<true, 3, 256>(p, s) : launch_gemm<false, 3, 256>(p, s);
    return p.in_scale ? launch_gemm<true, 2, 256>(p, s) : launch_gemm<false, 2, 256>(p, s);
  }
  if (v3) return p.in_scale ? launch_gemm<true, 3, 128>(p, s) : launch_gemm<false, 3, 128>(p, s);
  return p.in_scale ? launch_gemm<true, 2, 128>(p, s) : launch_gemm<false, 2, 128>(p, s);
}
