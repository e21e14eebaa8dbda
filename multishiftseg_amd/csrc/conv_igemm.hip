// Implicit-GEMM convolution on the fp32 matrix cores of MI355X (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the DeepWV3Plus path, every nn.Conv2d the reference runs through cuDNN
// (reference: lib/network/deepv3/deepv3.py:47-92,217-285; wider_resnet.py:64-182,288-364).
// All 54 convs are bias-free, 1x1 or 3x3, stride 1/2, dilation 1/2/4/12/24/36, padding = dilation.
//
// GEMM view: C[m][k] = sum_{tap,c} A[m][(tap,c)] * Wp[tap][k][c]
//   m   = output pixel (n, oy, ox)   -> MFMA row i
//   k   = output channel             -> MFMA column j (lane & 31): stores are 128-B segments
//   A   = NHWC input gathered at (oy*stride - pad + r*dil, ox*stride - pad + s*dil), zero outside
// Activations are NHWC with an explicit pixel stride (ldx/ldy) so a conv can read from / write
// into a channel slice of a wider (concat) tensor without a copy.
//
// Fused on the A side (prologue): per-channel affine + ReLU, i.e. eval- or train-mode
// BatchNorm+ReLU of the *previous* layer (pre-activation blocks, wider_resnet.py:43-48,169-182);
// the affine may be per-sample so that Dropout2d's channel mask folds in too.
// Fused on the C side (epilogue): per-channel affine, residual add, ReLU.
//
// Taps that are dead for the whole 128-pixel tile (all rows fall in the zero padding, common
// for dilation 12/24/36 on /8 maps) are skipped with a block-uniform decision.
#include "mss_epilogue.h"
#include <stdlib.h>

int mss_gemm_nt_dispatch(MssConvArgs p, void* stream);   // gemm.hip: persistent GEMM for the 1x1 / stride-1 shapes
bool mss_gemm_nt_eligible(const MssConvArgs& p);
bool mss_gemm_few_rows(const MssConvArgs& p);
bool mss_gemm_nt_bf16x3_eligible(const MssConvArgs& p);   // gemm_bf16x3.hip
bool mss_conv_bf16x3_eligible(const MssConvArgs& p);
int mss_conv_bf16x3_launch(MssConvArgs p, void* stream);
bool mss_wgrad_tn_bf16x3_eligible(const MssConvArgs& p, int lddy);
long long mss_wgrad_tn_bf16x3_ws_bytes(const MssConvArgs& p, int Cp);
int mss_wgrad_tn_bf16x3_launch(const MssConvArgs& p, const float* dy, int lddy, float* dwp, int Cp, float* ws, long long ws_bytes, void* stream);

namespace {

constexpr int NT = 256;

template <int BM, int BN, int BK, int WM, int WN, bool PER_SAMPLE, bool AFFINE>
__global__ __launch_bounds__(NT) void conv_igemm_kernel(MssConvArgs p) {
  constexpr int LDK = BK + 4;            // +4 floats: ds_read_b128 of 16 distinct rows is conflict-free
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int CPR = BK / 4;            // float4 chunks per tile row
  constexpr int RPP = NT / CPR;          // rows staged per pass
  constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for 256 threads");

  // all LDS in ONE dynamic array (a static __shared__ in front would shift its 16-B alignment)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* live_mask_p = reinterpret_cast<int*>(smem);  // first 16 B reserved
  float* As = smem + 4;                   // [2][BM][LDK]
  float* Bs = As + 2 * BM * LDK;          // [2][BN][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // batched mode (Winograd: 16 independent GEMMs in one launch): blockIdx.y picks the operand set
  p.x += (size_t)blockIdx.y * p.x_bs;
  p.w += (size_t)blockIdx.y * p.w_bs;
  p.y += (size_t)blockIdx.y * p.y_bs;
  const int v = mss_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = v / p.ntiles, nt = v % p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int chunk = tid % CPR;
  const int row0 = tid / CPR;

  // ---- per-thread descriptors of the A rows it stages ----
  int a_nb[A_LD], a_iy0[A_LD], a_ix0[A_LD], a_n[A_LD];
  const int ohw = p.OH * p.OW;
  int my_live = 0;
  if (tid == 0) *live_mask_p = 0;
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    int m = m0 + row0 + j * RPP;
    if (m < p.M) {
      int n = m / ohw, rem = m - n * ohw;
      int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_n[j] = n;
      a_nb[j] = n * p.H * p.W;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad;
      for (int t = 0; t < p.R * p.S; ++t) {
        int iy = a_iy0[j] + (t / p.S) * p.dil, ix = a_ix0[j] + (t % p.S) * p.dil;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) my_live |= 1 << t;
      }
    } else {
      a_n[j] = 0; a_nb[j] = 0; a_iy0[j] = -0x40000000; a_ix0[j] = -0x40000000;
    }
  }
  __syncthreads();
  if (my_live) atomicOr(live_mask_p, my_live);
  __syncthreads();
  const int live = *live_mask_p;
  const int nlive = __popc(live);
  const int cblocks = p.C / BK;
  const int n_it = nlive * cblocks;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- loader: runs one K-step ahead of the MFMA loop, split "issue early / write late":
  // issue_loads() only issues global loads (no wait, no branch: rows that fall in the zero padding
  // read a dummy valid address and are zeroed later); finish_store() runs AFTER the MFMA block of
  // the current step, applies the fused BatchNorm+ReLU prologue and writes the LDS tile. The loads'
  // latency is therefore covered by 64 MFMAs instead of stalling the wave 4x per step.
  // PER_SAMPLE: the prologue affine differs per image (Dropout2d fold) AND a tile may straddle two
  // images; then each staged row looks its affine up at LDS-write time. Otherwise one affine per
  // tile (offset by the tile's image when in_ss_stride != 0) rides along with the early loads.
  constexpr int S_LD = 1;
  f32x4 areg[A_LD], breg[B_LD], sreg[S_LD], hreg[S_LD];
  const float* a_ptr[A_LD];
  const float* b_ptr[B_LD];
  const float* s_ptr[S_LD];
  const float* h_ptr[S_LD];
  unsigned a_ok = 0;       // bit j: row j of this thread is inside the image for the current tap
  unsigned ld_ok = 0;      // a_ok of the step whose data sits in the staging registers
  int ld_cc = 0;           // its first channel (per-sample lookup at write time)
  int ld_tap = -1, ld_c0 = 0, ld_left = live;
  constexpr bool has_affine = AFFINE;
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();

  auto next_tap = [&]() {
    ld_tap = __ffs(ld_left) - 1;
    ld_left &= ld_left - 1;
    ld_c0 = 0;
    const int r = ld_tap / p.S, s = ld_tap - r * p.S;
    const int dy = r * p.dil, dx = s * p.dil;
    a_ok = 0;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      a_ok |= (ok ? 1u : 0u) << j;
      a_ptr[j] = ok ? p.x + (size_t)(a_nb[j] + iy * p.W + ix) * p.ldx + chunk * 4 : p.x;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      b_ptr[j] = p.w + ((size_t)ld_tap * p.Kpad + n0 + row0 + j * RPP) * p.C + chunk * 4;
    if (has_affine) {
#pragma unroll
      for (int j = 0; j < S_LD; ++j) {
        const size_t so = (size_t)(m0 / ohw) * p.in_ss_stride + chunk * 4;   // image of the tile's first row
        s_ptr[j] = p.in_scale + so;
        h_ptr[j] = p.in_shift + so;
      }
    }
  };

  auto issue_loads = [&]() {   // straight-line: every pointer is always a valid address
#pragma unroll
    for (int j = 0; j < A_LD; ++j) areg[j] = *reinterpret_cast<const f32x4*>(a_ptr[j]);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) breg[j] = *reinterpret_cast<const f32x4*>(b_ptr[j]);
    if (has_affine) {
#pragma unroll
      for (int j = 0; j < S_LD; ++j) {
        sreg[j] = *reinterpret_cast<const f32x4*>(s_ptr[j]);
        hreg[j] = *reinterpret_cast<const f32x4*>(h_ptr[j]);
      }
    }
    ld_cc = ld_c0 + chunk * 4;
    ld_ok = a_ok;
  };
  auto advance = [&]() {       // move the loader to the next K-step (stays put after the last one)
    if (ld_c0 + BK >= p.C) {
      if (ld_left) next_tap();
    } else {
      ld_c0 += BK;
#pragma unroll
      for (int j = 0; j < A_LD; ++j)
        if ((a_ok >> j) & 1) a_ptr[j] += BK;
#pragma unroll
      for (int j = 0; j < B_LD; ++j) b_ptr[j] += BK;
      if (has_affine) {
#pragma unroll
        for (int j = 0; j < S_LD; ++j) { s_ptr[j] += BK; h_ptr[j] += BK; }
      }
    }
  };

  auto finish_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      f32x4 val = areg[j];
      if (has_affine) {
        if (PER_SAMPLE) {
          const size_t so = (size_t)a_n[j] * p.in_ss_stride + ld_cc;
          val = val * *reinterpret_cast<const f32x4*>(p.in_scale + so) + *reinterpret_cast<const f32x4*>(p.in_shift + so);
        } else {
          val = val * sreg[0] + hreg[0];
        }
      }
      val.x = fmaxf(val.x, relu_floor); val.y = fmaxf(val.y, relu_floor);
      val.z = fmaxf(val.z, relu_floor); val.w = fmaxf(val.w, relu_floor);
      if (!((ld_ok >> j) & 1)) val = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&As[(buf * BM + row0 + j * RPP) * LDK + chunk * 4]) = val;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      *reinterpret_cast<f32x4*>(&Bs[(buf * BN + row0 + j * RPP) * LDK + chunk * 4]) = breg[j];
  };

  if (n_it > 0) {
    next_tap();
    issue_loads();
    finish_store(0);
    advance();
    issue_loads();       // the loader runs TWO K-steps ahead with one register set (see the loop): registers = step 1
    advance();
  }
  __syncthreads();

  // ---- main loop, software pipelined inside the wave -------------------------------------------
  // A K-step is NKC chunks of 8 k (= 4 MFMA k-steps x TM x TN tiles). The A/B fragments of chunk
  // kc+1 are read from LDS while chunk kc multiplies (two fragment register sets); the staging
  // registers are written to the other LDS buffer before the second-to-last chunk, the single
  // workgroup barrier sits before the last chunk, and the first fragments of the NEXT K-step are
  // read right after it -- so neither the global-load latency, nor the LDS round trip, nor the
  // barrier skew is ever exposed without MFMAs in flight.
  constexpr int NKC = BK / 8;
  const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
  const float* Abase = &As[(wm * WTM + frag_row) * LDK + frag_k];
  const float* Bbase = &Bs[(wn * WTN + frag_row) * LDK + frag_k];
  f32x4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](int set, int buf, int kc) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[set][i] = *reinterpret_cast<const f32x4*>(Abase + (buf * BM + i * 32) * LDK + kc * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[set][j] = *reinterpret_cast<const f32x4*>(Bbase + (buf * BN + j * 32) * LDK + kc * 8);
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

  if (n_it > 0) load_frags(0, 0, 0);
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    // The body is branch-free so the scheduler may interleave loads, LDS traffic, prologue math and
    // MFMAs freely: in the last step the loader re-reads its (still valid) last tile and stages it
    // into the buffer nobody reads any more.
    // The staging registers hold step it+1, requested a whole K-step ago: they go to LDS first and are re-issued at once
    // for step it+2, so no wave ever waits on a load it has just issued (gemm.hip VARIANT 2: +3-7 % on every shape).
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc + 1 < NKC) load_frags((kc + 1) & 1, buf, kc + 1);
      if (kc == NKC - 2) { finish_store(buf ^ 1); issue_loads(); advance(); }
      if (kc == NKC - 1) {
        __syncthreads();
        load_frags(NKC & 1, buf ^ 1, 0);
      }
      mfma_chunk(kc & 1);
    }
  }

  // ---- epilogue (mss_epilogue.h): affine / residual / ReLU, stores never serialised on a memory round trip ----
  mss_epilogue_store<TM, TN>(acc, p, p.y, m0 + wm * WTM, n0 + wn * WTN, lane);
}

template <int BM, int BN, int BK, int WM, int WN, bool PS, bool AFF>
int launch_conv_t(MssConvArgs& p, hipStream_t stream);

template <int BM, int BN, int BK, int WM, int WN>
int launch_conv(MssConvArgs& p, hipStream_t stream) {
  if (!p.in_scale) return launch_conv_t<BM, BN, BK, WM, WN, false, false>(p, stream);
  // a tile can only straddle two images when the image's pixel count is not a multiple of BM
  if (p.in_ss_stride && (p.OH * p.OW) % BM != 0) return launch_conv_t<BM, BN, BK, WM, WN, true, true>(p, stream);
  return launch_conv_t<BM, BN, BK, WM, WN, false, true>(p, stream);
}

template <int BM, int BN, int BK, int WM, int WN, bool PS, bool AFF>
int launch_conv_t(MssConvArgs& p, hipStream_t stream) {
  p.mtiles = mss_cdiv(p.M, BM);
  p.ntiles = mss_cdiv(p.K, BN);
  if (p.Kpad < p.ntiles * BN) return MSS_ERR_BAD_ARG;
  const size_t smem = ((size_t)2 * (BM + BN) * (BK + 4) + 4) * sizeof(float);
  auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN, PS, AFF>;
  if (smem > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles, p.batch > 1 ? p.batch : 1), dim3(NT), smem, stream, p);
  return mss_launch_status();
}

// [K][C][R][S] (PyTorch) -> [R*S][Kpad][Cp], zero padded. flip=1 builds the dgrad weights:
// a conv from K channels back to C channels with the taps rotated 180 degrees.
__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int C,
                                    int R, int S, int Kpad, int Cp, int flip) {
  const size_t total = (size_t)R * S * Kpad * Cp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int c = i % Cp;
    int k = (i / Cp) % Kpad;
    int t = i / ((size_t)Cp * Kpad);
    int r = t / S, s = t % S;
    float val = 0.f;
    if (!flip) {
      if (k < K && c < C) val = src[(((size_t)k * C + c) * R + r) * S + s];
    } else {
      // output channel index k runs over the conv's *input* channels (C of src), input index c over src's K
      if (k < C && c < K) val = src[(((size_t)c * C + k) * R + (R - 1 - r)) * S + (S - 1 - s)];
    }
    dst[i] = val;
  }
}

// [R*S][Kpad][Cp] packed gradient -> accumulate/assign into [K][C][R][S]
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int C,
                                    int R, int S, int Kpad, int Cp, int accumulate) {
  const size_t total = (size_t)K * C * R * S;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int s = i % S;
    int r = (i / S) % R;
    int c = (i / ((size_t)S * R)) % C;
    int k = i / ((size_t)S * R * C);
    float val = src[((size_t)(r * S + s) * Kpad + k) * Cp + c];
    dst[i] = accumulate ? dst[i] + val : val;
  }
}

// ------------------------------------------------------------------------------------------
// wgrad: dWp[tap][k][c] = sum_m dy[m][k] * act(x[m_tap][c])   (reduction over output pixels)
// MFMA rows = k (output channels), columns = c (input channels), contraction = pixels.
// Both operands sit in LDS as [pixel][channel] (their natural NHWC order) and are read with
// ds_read_b32: lane (i = l&31, kk = l>>5) reads row kk, column i -> 32 consecutive floats, so any
// row stride is conflict-free; the stride is a multiple of 4 floats so staging uses ds_write_b128.
// Same in-wave pipeline as the forward kernel: loads for pixel block t+1 issued first, written to
// LDS before the second-to-last chunk, one barrier before the last chunk.
// Grid: (ktiles*ctiles, taps, splits); the pixel range is split across blockIdx.z. No atomics: with one split the
// tile is stored straight into dWp; with several, split z stores its partial tile into slab z of a workspace
// ([splits][taps][Kpad][Cp], every element written by exactly one workgroup) and wgrad_reduce_kernel adds the slabs
// in split order -- the weight gradient is bit-reproducible run to run (the reference pins cudnn.deterministic,
// lib/utils/utils.py:10-13).
template <int BKO, int BCI, int BP, int WK>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(MssConvArgs p, const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ dwp, int Cp, int pix_per_split) {
  constexpr int LDA = BKO + 4;  // dy tile  [BP][BKO]
  constexpr int LDB = BCI + 4;  // x  tile  [BP][BCI]
  constexpr int WC = 4 / WK;    // waves along K x waves along C (2x2; 1x4 for the 32-row tile of the 19-channel heads)
  constexpr int WTK = BKO / WK, WTC = BCI / WC;
  constexpr int TM = WTK / 32, TN = WTC / 32;
  constexpr int A_CPR = BKO / 4, B_CPR = BCI / 4;
  constexpr bool A_PART = BP * A_CPR < NT;                // dy tile smaller than one float4 per thread: upper threads idle
  constexpr int A_LD = A_PART ? 1 : BP * A_CPR / NT, B_LD = BP * B_CPR / NT;
  constexpr int A_RPP = NT / A_CPR, B_RPP = NT / B_CPR;   // pixel rows covered per staging pass
  constexpr int NKC = BP / 8;                             // chunks of 8 pixels = 4 MFMA k-steps
  static_assert(TM >= 1 && TN >= 1 && B_LD >= 1 && NKC >= 2, "");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BP][LDA]
  float* Bs = smem + 2 * BP * LDA;     // [2][BP][LDB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WC, wn = wave % WC;
  const int ctiles = mss_cdiv(p.C, BCI);
  const int kt = blockIdx.x / ctiles, ct = blockIdx.x % ctiles;
  const int k0 = kt * BKO, c0 = ct * BCI;
  const int tap = blockIdx.y;           // output slab index: a filter tap, or (batched mode) a Winograd position
  int geo_tap = tap;
  if (p.batch > 1) {                    // 16 independent [K x T] x [T x C] products (Winograd weight gradient)
    p.x += (size_t)tap * p.x_bs;
    dy += (size_t)tap * p.y_bs;
    geo_tap = 0;
  }
  const int r = geo_tap / p.S, s = geo_tap - r * p.S;
  const int dyo = r * p.dil - p.pad, dxo = s * p.dil - p.pad;
  const int mbeg = blockIdx.z * pix_per_split;
  const int mend = min(p.M, mbeg + pix_per_split);
  const int ohw = p.OH * p.OW;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // ---- per-thread staging descriptors: pointers advance by BP pixels per step; the pixel -> (n, y, x)
  // decode (two integer divisions) is only redone when a row of this thread wraps to the next image row
  const int a_ch = (tid % A_CPR) * 4, a_pr0 = tid / A_CPR;
  const int b_ch = (tid % B_CPR) * 4, b_pr0 = tid / B_CPR;
  const bool a_full = k0 + a_ch + 3 < p.K;          // whole float4 of output channels exists
  const bool a_act = !A_PART || a_pr0 < BP;
  const bool b_in = c0 + b_ch < p.C;
  const float* a_ptr[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) a_ptr[j] = dy + (size_t)(mbeg + a_pr0 + j * A_RPP) * lddy + k0 + a_ch;
  const size_t a_step = (size_t)BP * lddy;
  const float* b_ptr[B_LD];
  int b_n[B_LD], b_oy[B_LD], b_ox[B_LD], b_ix[B_LD];
  unsigned b_rowok = 0;
  auto place_row = [&](int j) {   // (n, oy, ox) -> source pointer / validity of the image row
    const int iy = b_oy[j] * p.stride + dyo;
    b_ix[j] = b_ox[j] * p.stride + dxo;
    const bool ok = b_in && b_n[j] < p.N && (unsigned)iy < (unsigned)p.H;
    b_rowok = (b_rowok & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    b_ptr[j] = ok ? p.x + ((size_t)(b_n[j] * p.H + iy) * p.W) * p.ldx + c0 + b_ch : p.x;
  };
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int m = mbeg + b_pr0 + j * B_RPP;
    const int n = m / ohw, rem = m - n * ohw;
    b_n[j] = n; b_oy[j] = rem / p.OW; b_ox[j] = rem - b_oy[j] * p.OW;
    place_row(j);
  }
  const bool has_affine = p.in_scale != nullptr;
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();
  f32x4 areg[A_LD], breg[B_LD], sreg[B_LD], hreg[B_LD];
  unsigned ld_ok = 0;
  int ld_m = mbeg;    // first pixel of the block the loader fetches next

  auto issue_loads = [&]() {
    const bool tail = ld_m + BP > mend;   // block-uniform: only the last step of a split can be ragged
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const bool ok = a_act && (!tail || ld_m + a_pr0 + j * A_RPP < mend);
      const float* src = ok ? a_ptr[j] : dy + k0 + a_ch;
      f32x4 val;
      if (a_full) val = *reinterpret_cast<const f32x4*>(src);
      else {
        val.x = k0 + a_ch + 0 < p.K ? src[0] : 0.f;
        val.y = k0 + a_ch + 1 < p.K ? src[1] : 0.f;
        val.z = k0 + a_ch + 2 < p.K ? src[2] : 0.f;
        val.w = 0.f;
      }
      areg[j] = ok ? val : f32x4{0.f, 0.f, 0.f, 0.f};
      a_ptr[j] += a_step;
    }
    ld_ok = 0;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const bool ok = ((b_rowok >> j) & 1) && (unsigned)b_ix[j] < (unsigned)p.W &&
                      (!tail || ld_m + b_pr0 + j * B_RPP < mend);
      ld_ok |= (ok ? 1u : 0u) << j;
      const float* src = ok ? b_ptr[j] + (size_t)b_ix[j] * p.ldx : p.x;
      breg[j] = *reinterpret_cast<const f32x4*>(src);
      if (has_affine) {
        const size_t so = (ok ? (size_t)b_n[j] * p.in_ss_stride + c0 + b_ch : 0);
        sreg[j] = *reinterpret_cast<const f32x4*>(p.in_scale + so);
        hreg[j] = *reinterpret_cast<const f32x4*>(p.in_shift + so);
      }
      // advance this row's pixel by BP for the next step
      b_ox[j] += BP;
      b_ix[j] += BP * p.stride;
      if (b_ox[j] >= p.OW) {
        do { b_ox[j] -= p.OW; b_oy[j] += 1; } while (b_ox[j] >= p.OW);
        while (b_oy[j] >= p.OH) { b_oy[j] -= p.OH; b_n[j] += 1; }
        place_row(j);
      }
    }
    ld_m += BP;
  };
  auto finish_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      if (a_act) *reinterpret_cast<f32x4*>(&As[(buf * BP + a_pr0 + j * A_RPP) * LDA + a_ch]) = areg[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      f32x4 val = breg[j];
      if (has_affine) val = val * sreg[j] + hreg[j];
      val.x = fmaxf(val.x, relu_floor); val.y = fmaxf(val.y, relu_floor);
      val.z = fmaxf(val.z, relu_floor); val.w = fmaxf(val.w, relu_floor);
      if (!((ld_ok >> j) & 1)) val = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&Bs[(buf * BP + b_pr0 + j * B_RPP) * LDB + b_ch]) = val;
    }
  };

  const int fi = lane & 31, fk = lane >> 5;
  float fa[2][4][TM], fb[2][4][TN];
  auto load_frags = [&](int set, int buf, int kc) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int row = buf * BP + kc * 8 + ks * 2 + fk;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][ks][i] = As[row * LDA + wm * WTK + i * 32 + fi];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][ks][j] = Bs[row * LDB + wn * WTC + j * 32 + fi];
    }
  };
  auto mfma_chunk = [&](int set) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][ks][i], fb[set][ks][j], acc[i][j], 0, 0, 0);
  };

  const int n_it = mend > mbeg ? (mend - mbeg + BP - 1) / BP : 0;
  if (n_it > 0) { issue_loads(); finish_store(0); }
  __syncthreads();
  if (n_it > 0) load_frags(0, 0, 0);
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    // branch-free body (see the forward kernel): in the last step the loader runs past the split's
    // end, where every row is masked to a dummy address, and stages zeros nobody reads.
    // (The forward kernels' two-steps-ahead loader was measured here too: 104.8 -> 96.3 TFLOP/s. This loader's pixel
    // decode is VALU-heavy and does better at the top of the step, next to the fragment reads.)
    issue_loads();
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc + 1 < NKC) load_frags((kc + 1) & 1, buf, kc + 1);
      if (kc == NKC - 2) finish_store(buf ^ 1);
      if (kc == NKC - 1) {
        __syncthreads();
        load_frags(NKC & 1, buf ^ 1, 0);
      }
      mfma_chunk(kc & 1);
    }
  }

  const int colq = lane & 31, rowq = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    // rows in [K, Kpad) and columns in [C, Cp) were fed zeros, so their accumulators are exact zeros: storing them
    // too means every element of the [Kpad][Cp] slab is written and nobody has to clear it first
    const int col = c0 + wn * WTC + j * 32 + colq;
    if (col >= Cp) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = k0 + wm * WTK + i * 32 + (q & 3) + 8 * (q >> 2) + rowq;
        if (row < p.Kpad) dwp[(((size_t)blockIdx.z * gridDim.y + tap) * p.Kpad + row) * Cp + col] = acc[i][j][q];
      }
  }
}

// ------------------------------------------------------------------------------------------
// Batched TN GEMM for the Winograd-domain weight gradient: dU[p][k][c] = sum_t dY'[p][t][k] * X'[p][t][c].
// conv_wgrad_kernel computes the same thing with its convolution loader (pixel decode, tap geometry, per-row masks:
// 5 VALU instructions per MFMA) and one tile per workgroup; here nothing of that is left: both operands are plain
// row-major [T][channels] matrices, a PERSISTENT workgroup walks (position, k-tile, c-tile, T-range) work items, and the
// loader runs two 16-row steps ahead with one register set, across work-item boundaries (gemm.hip, VARIANT 2).
// MFMA rows = k, columns = c, contraction = t; LDS tiles [2][16][128+4]; fragments by ds_read_b32 (lane i reads column
// i of row 2s + (lane >> 5): 32 consecutive floats, conflict-free).
// Work item w = ((split * P + p) * ktiles + kt) * ctiles + ct; its 128x128 tile goes to slab `split` of dst.
constexpr int TN_BK = 128, TN_BC = 128, TN_BT = 16, TN_LD = 132;
template <bool TWO_AHEAD>
__global__ __launch_bounds__(NT, 3) void gemm_tn_wgrad_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ dst, int P, int T, int K, int C,
                                                              long long a_bs, long long b_bs, int Kpad, int Cp,
                                                              int ktiles, int ctiles, int splits, int t_per_split,
                                                              long long total) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                              // [2][16][132]
  float* Bs = smem + 2 * TN_BT * TN_LD;          // [2][16][132]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 5, lcol = (tid & 31) * 4;        // loader: rows lrow, lrow + 8; 4 consecutive channels
  const long long stride = gridDim.x;

  // ---- loader state ----
  long long ld_w = mss_xcd_remap(blockIdx.x, gridDim.x);
  const float* a_ptr = A;
  const float* b_ptr = B;
  int ld_t = 0, ld_tend = 0;
  bool a_colok = false, b_colok = false;
  auto setup = [&](long long w) {
    const int ct = (int)(w % ctiles); w /= ctiles;
    const int kt = (int)(w % ktiles); w /= ktiles;
    const int p = (int)(w % P);
    const int sp = (int)(w / P);
    ld_t = sp * t_per_split;
    ld_tend = min(T, ld_t + t_per_split);
    a_colok = kt * TN_BK + lcol < K;             // K, C are multiples of 4: a float4 is inside or outside as a whole
    b_colok = ct * TN_BC + lcol < C;
    a_ptr = A + (size_t)p * a_bs + (size_t)ld_t * K + (a_colok ? kt * TN_BK + lcol : 0);
    b_ptr = B + (size_t)p * b_bs + (size_t)ld_t * C + (b_colok ? ct * TN_BC + lcol : 0);
  };
  f32x4 areg[2], breg[2];
  const bool edge_free = K % TN_BK == 0 && C % TN_BC == 0;
  auto issue_loads = [&]() {
    if (edge_free && ld_t + TN_BT <= ld_tend) {      // workgroup-uniform: no row or channel mask needed
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        areg[j] = *reinterpret_cast<const f32x4*>(a_ptr + (size_t)(lrow + 8 * j) * K);
        breg[j] = *reinterpret_cast<const f32x4*>(b_ptr + (size_t)(lrow + 8 * j) * C);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = ld_t + lrow + 8 * j;
      const bool ok = t < ld_tend;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 va = *reinterpret_cast<const f32x4*>(ok ? a_ptr + (size_t)(lrow + 8 * j) * K : A);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(ok ? b_ptr + (size_t)(lrow + 8 * j) * C : B);
      areg[j] = (ok && a_colok) ? va : z;
      breg[j] = (ok && b_colok) ? vb : z;
    }
  };
  auto advance = [&]() {
    ld_t += TN_BT;
    if (ld_t < ld_tend) {
      a_ptr += (size_t)TN_BT * K;
      b_ptr += (size_t)TN_BT * C;
    } else {
      ld_w += stride;
      setup(ld_w < total ? ld_w : ld_w - stride);          // past the end: re-read the last item, never used
    }
  };
  auto finish_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<f32x4*>(&As[(buf * TN_BT + lrow + 8 * j) * TN_LD + lcol]) = areg[j];
      *reinterpret_cast<f32x4*>(&Bs[(buf * TN_BT + lrow + 8 * j) * TN_LD + lcol]) = breg[j];
    }
  };
  const int fi = lane & 31, fk = lane >> 5;
  float fa[2][4][2], fb[2][4][2];
  auto load_frags = [&](int set, int buf, int kc) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int row = buf * TN_BT + kc * 8 + ks * 2 + fk;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[set][ks][i] = As[row * TN_LD + wm * 64 + i * 32 + fi];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[set][ks][j] = Bs[row * TN_LD + wn * 64 + j * 32 + fi];
    }
  };
  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  auto mfma_chunk = [&](int set) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][ks][i], fb[set][ks][j], acc[i][j], 0, 0, 0);
  };
  auto epilogue = [&](long long w) {
    const int ct = (int)(w % ctiles); w /= ctiles;
    const int kt = (int)(w % ktiles); w /= ktiles;      // w = split * P + p: the slab index
    float* o = dst + (size_t)w * Kpad * Cp;
    const int colq = lane & 31, rowq = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = ct * TN_BC + wn * 64 + j * 32 + colq;
      if (col >= Cp) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = kt * TN_BK + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + rowq;
          if (row < Kpad) o[(size_t)row * Cp + col] = acc[i][j][q];     // rows >= K / cols >= C were fed zeros
        }
    }
  };

  long long cur = ld_w;                     // item being multiplied (the launch guarantees cur < total)
  int steps_left;                           // 16-row steps left in the current item
  {
    long long w = cur / ((long long)ktiles * ctiles);
    const int sp = (int)(w / P);
    const int t0 = sp * t_per_split;
    steps_left = (min(T, t0 + t_per_split) - t0 + TN_BT - 1) / TN_BT;
  }
  setup(ld_w);
  issue_loads();
  finish_store(0);
  advance();
  if (TWO_AHEAD) { issue_loads(); advance(); }
  zero_acc();
  __syncthreads();
  load_frags(0, 0, 0);
  int buf = 0;
  while (true) {
    if (TWO_AHEAD) {
      load_frags(1, buf, 1);
      finish_store(buf ^ 1);                  // step +1, requested during the previous step
      issue_loads();                          // step +2 (possibly of the next work item)
      advance();
    } else {
      issue_loads();                          // step +1, stored in this step
      load_frags(1, buf, 1);
      finish_store(buf ^ 1);
      advance();
    }
    mfma_chunk(0);
    __syncthreads();
    load_frags(0, buf ^ 1, 0);
    mfma_chunk(1);
    buf ^= 1;
    if (--steps_left == 0) {
      epilogue(cur);
      cur += stride;
      if (cur >= total) break;
      zero_acc();
      long long w = cur / ((long long)ktiles * ctiles);
      const int sp = (int)(w / P);
      const int t0 = sp * t_per_split;
      steps_left = (min(T, t0 + t_per_split) - t0 + TN_BT - 1) / TN_BT;
    }
  }
}

// Same products, operands TRANSPOSED ON THE WAY INTO LDS so that the inner loop is the NT GEMM's (gemm.hip): waves 0-1 load
// A (dY'), waves 2-3 load B (X'); a thread fetches 4 consecutive t rows x 4 channels (each a coalesced 16-byte load), and the
// 4x4 block leaves for LDS as four ds_write_b128 of [channel][4 consecutive t] -- the transpose is register naming. LDS
// tiles are [128 channels][16 t + 4] with the NT kernel's chunk rotation, so a lane reads the 4 contraction steps of its
// channel with ONE ds_read_b128 (8 LDS reads per 32 MFMAs instead of 32 ds_read_b32). Arithmetic: the 16 t of a step are
// consumed in the order (s, s + 4 | s + 8, s + 12), s = 0..3, instead of (2s, 2s + 1): sums over t are re-associated, results
// differ from gemm_tn_wgrad_kernel in the last bits and are as deterministic (fixed order).
constexpr int TN2_LDK = 20;
// BC: c extent of a tile, 128 (3 workgroups per CU) or 256 (2x2 waves of 64 x 128, 2 workgroups per CU; every thread loads a
// 4x4 block of B and threads 0-127 one of A as well).
template <int BC>
__global__ __launch_bounds__(NT, BC == 256 ? 2 : 3) void gemm_tn2_wgrad_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ dst, int P, int T, int K, int C, long long a_bs,
    long long b_bs, int Kpad, int Cp, int ktiles, int ctiles, int splits, int t_per_split, long long total) {
  constexpr int TNJ = BC / 64;                          // 32-column MFMA blocks per wave
  constexpr bool WIDE = BC == 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [2][128][20]
  float* Bs = smem + 2 * 128 * TN2_LDK;               // [2][BC][20]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long stride = gridDim.x;
  // unit 0: BC = 128: A for waves 0-1, B for waves 2-3; BC = 256: B for every thread. unit 1 (WIDE, threads 0-127): A.
  const bool u0B = WIDE || tid >= 128;                  // wave-uniform
  const int cq0 = WIDE ? (tid & 63) : (tid & 31), tg0 = WIDE ? (tid >> 6) : ((tid & 127) >> 5);
  const int cols0 = u0B ? C : K;
  const float* const base0 = u0B ? B : A;
  // chunk rotation (r >> 2) + (r >> 4) of row r = 4 cq + e: the 16 lanes of a ds_write_b128 group then cover all 64 banks
  // (with (r >> 2) alone, rows 16 apart met in the same banks: PMC showed 2/3 of the LDS cycles as bank conflicts)
  float* const ldst0 = (u0B ? Bs : As) + (4 * cq0) * TN2_LDK + ((tg0 + cq0 + (cq0 >> 2)) & 3) * 4;
  const bool has1 = WIDE && tid < 128;                  // wave-uniform
  const int cq1 = tid & 31, tg1 = (tid >> 5) & 3;
  float* const ldst1 = As + (4 * cq1) * TN2_LDK + ((tg1 + cq1 + (cq1 >> 2)) & 3) * 4;

  const bool edge_free = K % 128 == 0 && C % BC == 0;   // no channel tile hangs over the edge: no per-lane column mask
  long long ld_w = mss_xcd_remap(blockIdx.x, gridDim.x);
  const float* ptr0 = base0;
  const float* ptr1 = A;
  int ld_t = 0, ld_tend = 0;
  bool colok0 = false, colok1 = false;
  auto setup = [&](long long w) {
    const int ct = (int)(w % ctiles); w /= ctiles;
    const int kt = (int)(w % ktiles); w /= ktiles;
    const int p = (int)(w % P);
    const int sp = (int)(w / P);
    ld_t = sp * t_per_split;
    ld_tend = min(T, ld_t + t_per_split);
    const int col0 = (u0B ? ct * BC : kt * 128) + 4 * cq0;
    colok0 = col0 < cols0;
    ptr0 = base0 + (size_t)p * (u0B ? b_bs : a_bs) + (size_t)ld_t * cols0 + (colok0 ? col0 : 0);
    if (WIDE) {
      const int col1 = kt * 128 + 4 * cq1;
      colok1 = col1 < K;
      ptr1 = A + (size_t)p * a_bs + (size_t)ld_t * K + (colok1 ? col1 : 0);
    }
  };
  f32x4 reg0[4], reg1[WIDE ? 4 : 1];
  auto issue_loads = [&]() {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (edge_free && ld_t + 16 <= ld_tend) {        // workgroup-uniform: whole 16-row step inside, no ragged channel tile
#pragma unroll
      for (int e = 0; e < 4; ++e) reg0[e] = *reinterpret_cast<const f32x4*>(ptr0 + (size_t)(4 * tg0 + e) * cols0);
      if (WIDE && has1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) reg1[WIDE ? e : 0] = *reinterpret_cast<const f32x4*>(ptr1 + (size_t)(4 * tg1 + e) * K);
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * tg0 + e;
      const bool ok = ld_t + r < ld_tend;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? ptr0 + (size_t)r * cols0 : base0);
      reg0[e] = (ok && colok0) ? v : z;
    }
    if (WIDE && has1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * tg1 + e;
        const bool ok = ld_t + r < ld_tend;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? ptr1 + (size_t)r * K : A);
        reg1[WIDE ? e : 0] = (ok && colok1) ? v : z;
      }
    }
  };
  auto advance = [&]() {
    ld_t += 16;
    if (ld_t < ld_tend) {
      ptr0 += (size_t)16 * cols0;
      if (WIDE) ptr1 += (size_t)16 * K;
    } else {
      ld_w += stride;
      setup(ld_w < total ? ld_w : ld_w - stride);
    }
  };
  auto finish_store = [&](int buf) {
    float* d = ldst0 + buf * (u0B ? BC : 128) * TN2_LDK;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 w = {reg0[0][e], reg0[1][e], reg0[2][e], reg0[3][e]};     // channel 4 cq + e, t = 4 tg .. 4 tg + 3
      *reinterpret_cast<f32x4*>(d + e * TN2_LDK) = w;
    }
    if (WIDE && has1) {
      float* d1 = ldst1 + buf * 128 * TN2_LDK;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 w = {reg1[0][e], reg1[WIDE ? 1 : 0][e], reg1[WIDE ? 2 : 0][e], reg1[WIDE ? 3 : 0][e]};
        *reinterpret_cast<f32x4*>(d1 + e * TN2_LDK) = w;
      }
    }
  };
  const int frag_row = lane & 31, frag_h = lane >> 5;
  const int rot = (frag_row >> 2) + (frag_row >> 4);      // + 2 per 32-row block (the block base's (r >> 4) mod 4)
  const float* Abase = &As[(wm * 64 + frag_row) * TN2_LDK];
  const float* Bbase = &Bs[(wn * (BC / 2) + frag_row) * TN2_LDK];
  // logical chunk kc * 2 + frag_h of a row in an even / odd 32-row block
  const int koff[2][2] = {{((frag_h + rot) & 3) * 4, ((frag_h + rot + 2) & 3) * 4},
                          {((2 + frag_h + rot) & 3) * 4, ((2 + frag_h + rot + 2) & 3) * 4}};
  f32x4 fa[2][2], fb[2][TNJ];
  auto load_frags = [&](int set, int buf, int kc) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(Abase + (buf * 128 + i * 32) * TN2_LDK + koff[kc][i & 1]);
#pragma unroll
    for (int j = 0; j < TNJ; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(Bbase + (buf * BC + j * 32) * TN2_LDK + koff[kc][j & 1]);
  };
  f32x16 acc[2][TNJ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TNJ; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  auto mfma_chunk = [&](int set) {
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][s2], fb[set][j][s2], acc[i][j], 0, 0, 0);
  };
  auto epilogue = [&](long long w) {
    const int ct = (int)(w % ctiles); w /= ctiles;
    const int kt = (int)(w % ktiles); w /= ktiles;      // w = split * P + p: the slab index
    float* o = dst + (size_t)w * Kpad * Cp;
    const int colq = lane & 31, rowq = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TNJ; ++j) {
      const int col = ct * BC + wn * (BC / 2) + j * 32 + colq;
      if (col >= Cp) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = kt * 128 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + rowq;
          if (row < Kpad) o[(size_t)row * Cp + col] = acc[i][j][q];
        }
    }
  };

  long long cur = ld_w;
  int steps_left;
  {
    long long w = cur / ((long long)ktiles * ctiles);
    const int sp = (int)(w / P);
    const int t0 = sp * t_per_split;
    steps_left = (min(T, t0 + t_per_split) - t0 + 15) / 16;
  }
  setup(ld_w);
  issue_loads();
  finish_store(0);
  advance();
  zero_acc();
  __syncthreads();
  load_frags(0, 0, 0);
  int buf = 0;
  while (true) {
    issue_loads();                          // step +1, stored in this step
    load_frags(1, buf, 1);
    finish_store(buf ^ 1);
    advance();
    mfma_chunk(0);
    __syncthreads();
    load_frags(0, buf ^ 1, 0);
    mfma_chunk(1);
    buf ^= 1;
    if (--steps_left == 0) {
      epilogue(cur);
      cur += stride;
      if (cur >= total) break;
      zero_acc();
      long long w = cur / ((long long)ktiles * ctiles);
      const int sp = (int)(w / P);
      const int t0 = sp * t_per_split;
      steps_left = (min(T, t0 + t_per_split) - t0 + 15) / 16;
    }
  }
}

// dwp[tap][row][col] = sum over splits (ascending) of ws[split][tap][row][col], float4 over col (Cp % 4 == 0)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dwp,
                                                           long long slab4, int splits) {
  const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
  f32x4* d4 = reinterpret_cast<f32x4*>(dwp);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slab4;
       i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = w4[i];
    int sp = 1;
    for (; sp + 8 <= splits; sp += 8) {        // eight loads in flight, added in split order (the sum is bit-reproducible)
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = w4[(long long)(sp + u) * slab4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; sp < splits; ++sp) a += w4[(long long)sp * slab4 + i];
    d4[i] = a;
  }
}

// The same sum when there are MANY splits of a SMALL slab (the narrow head gradients: 1536 splits of 20 KB; a Linear of the pixel
// decoder: 256 splits of 256 KB): one thread per output float4 walks the splits one dependent load chain after the other
// (1536 / 8 round trips = 90 us for 30 MB). Here G threads share an output element: thread g adds the splits g, g + G, g + 2G, ...
// in ascending order, and the G partial sums are added in the order g = 0 .. G-1 through LDS -- a fixed tree, so the result is as
// reproducible as the sequential sum (it is a different rounding of the same sum).
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_par_kernel(const float* __restrict__ ws, float* __restrict__ dwp, long long slab4,
                                                               int splits) {
  constexpr int EPB = 256 / G;                       // output float4s per workgroup
  __shared__ f32x4 part[G][EPB];
  const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
  const int el = threadIdx.x % EPB, g = threadIdx.x / EPB;
  const long long i = (long long)blockIdx.x * EPB + el;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < slab4) {
    int sp = g;
    for (; sp + 3 * G < splits; sp += 4 * G) {       // four loads in flight, added in ascending split order
      const f32x4 v0 = w4[(long long)sp * slab4 + i], v1 = w4[(long long)(sp + G) * slab4 + i];
      const f32x4 v2 = w4[(long long)(sp + 2 * G) * slab4 + i], v3 = w4[(long long)(sp + 3 * G) * slab4 + i];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; sp < splits; sp += G) a += w4[(long long)sp * slab4 + i];
  }
  part[g][el] = a;
  __syncthreads();
  if (g == 0 && i < slab4) {
    f32x4 t = part[0][el];
#pragma unroll
    for (int k = 1; k < G; ++k) t += part[k][el];
    reinterpret_cast<f32x4*>(dwp)[i] = t;
  }
}

inline void launch_wgrad_reduce(const float* ws, float* dwp, long long slab4, int splits, hipStream_t stream) {
  // threads wanted: ~64 k; G split-lanes per element while each lane still has >= 4 splits
  int G = 1;
  while (G < 32 && slab4 * G < 65536 && splits >= 8 * G) G *= 2;
  if (G == 1) {
    long long blocks = (slab4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, ws, dwp, slab4, splits);
    return;
  }
#define RED(G_) hipLaunchKernelGGL(wgrad_reduce_par_kernel<G_>, dim3((unsigned)((slab4 + 256 / G_ - 1) / (256 / G_))), dim3(256), 0, stream, ws, dwp, slab4, splits)
  switch (G) {
    case 2: RED(2); break;
    case 4: RED(4); break;
    case 8: RED(8); break;
    case 16: RED(16); break;
    default: RED(32); break;
  }
#undef RED
}

struct WgradPlan { int ktiles, ctiles, taps, splits, pps; };

template <int BKO, int BCI, int BP>
WgradPlan wgrad_plan(const MssConvArgs& p) {
  WgradPlan pl;
  pl.ktiles = mss_cdiv(p.K, BKO); pl.ctiles = mss_cdiv(p.C, BCI); pl.taps = p.batch > 1 ? p.batch : p.R * p.S;
  // Pixel splits: pick the smallest split count whose grid fills its last round of resident workgroups to >= 95 %
  // (a 1152-block grid on 768 slots runs 2 rounds for 1.5 rounds of work); more splits only add partial-slab traffic.
  // Small slabs (the 19 x 256 head gradient: 2 tiles) need hundreds of splits to fill the chip.
  const int base = pl.ktiles * pl.ctiles * pl.taps;
  const int slots = 768;   // 3 workgroups per CU (34 KB LDS, 154 registers)
  int max_splits = mss_cdiv(p.M, BP * 8);
  if (max_splits > 1024) max_splits = 1024;
  if (max_splits < 1) max_splits = 1;
  int splits = 1;
  double best = 0.0;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const long long total = (long long)base * sp;
    const double eff = (double)total / (double)(((total + slots - 1) / slots) * slots);
    if (eff > best + 1e-9) { best = eff; splits = sp; }
    if (eff >= 0.95 && total >= slots) break;
  }
  pl.pps = mss_cdiv(mss_cdiv(p.M, splits), BP) * BP;
  pl.splits = mss_cdiv(p.M, pl.pps);
  return pl;
}

// bytes of partial-slab workspace the launch needs (0 when the pixel range is not split)
template <int BKO, int BCI, int BP>
long long wgrad_ws_bytes(const MssConvArgs& p, int Cp) {
  const WgradPlan pl = wgrad_plan<BKO, BCI, BP>(p);
  return pl.splits > 1 ? (long long)pl.splits * pl.taps * p.Kpad * Cp * 4 : 0;
}

template <int BKO, int BCI, int BP, int WK>
int launch_wgrad(MssConvArgs& p, const float* dy, int lddy, float* dwp, int Cp, float* ws, long long ws_bytes,
                 hipStream_t stream) {
  const WgradPlan pl = wgrad_plan<BKO, BCI, BP>(p);
  const long long slab = (long long)pl.taps * p.Kpad * Cp;
  if (pl.splits > 1 && (!ws || ws_bytes < (long long)pl.splits * slab * 4)) return MSS_ERR_BAD_ARG;
  const size_t smem = (size_t)2 * BP * (BKO + 4 + BCI + 4) * sizeof(float);
  // staging after the whole MFMA block, and BP=32, were measured: within 1-4 % slower
  auto kern = conv_wgrad_kernel<BKO, BCI, BP, WK>;
  hipLaunchKernelGGL(kern, dim3(pl.ktiles * pl.ctiles, pl.taps, pl.splits), dim3(NT), smem, stream, p, dy, lddy,
                     pl.splits > 1 ? ws : dwp, Cp, pl.pps);
  if (pl.splits > 1) {
    // every element of every partial slab was written (see the kernel's epilogue), so whole slabs are swept
    launch_wgrad_reduce(ws, dwp, slab / 4, pl.splits, stream);
  }
  return mss_launch_status();
}

// ---- TN weight gradient WITHOUT LDS (r04 experiment -> see launch_wgrad_tn, MSS_WGRAD_TN=5 / default rule) --------------------------
// dW[k][c] = sum_r dy[r][k] * x[r][c]. In the 32x32x2 fp32 MFMA, operand A is [m][kk] with lane = m + 32 * kk: for THIS product the
// contraction index kk is the ROW of both operands, so the 32 lanes of one kk read 32 consecutive floats of one row -- the layout the
// tensors already have in memory. A lane therefore loads 16 bytes (4 consecutive columns) of row r0 + (lane >> 5) of dy and of x
// straight into registers and its four components feed four MFMAs each way: block (a, b) accumulates the output elements
// (k = k0 + 4 m + a, c = c0 + 4 n + b), i.e. the 128 x 128 tile of a WAVE is computed as 16 interleaved 32 x 32 blocks with
// 2 global loads per 16 MFMAs, no LDS staging, no transposing reads (gemm_tn_wgrad_kernel: one ds_read_b32 per MFMA and operand)
// and no workgroup barrier. The price: 256 accumulator registers per wave, so one wave per SIMD, and each operand row piece is
// fetched by every wave that needs it (from L2: 4 waves of a workgroup are the 4 column tiles of one k tile over the same rows).
// A ring of TND row pairs is in flight per wave. Output: whole 128 x 128 tiles of the (split, position) slab, 16-byte stores.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ const float tn_zero_row[4096] = {0.f};              // the A operand of rows past the end of a split
// KB: 32-column blocks of dy per wave. 4: a 128 x 128 tile, 256 accumulator registers, ONE wave per SIMD. 2: a 64 x 128 tile, 128
// accumulators, TWO waves per SIMD (the second wave fills the matrix pipe while the first issues its loads and address arithmetic)
// at 1.5x the operand traffic per MFMA.
// AFFINE: x enters as relu(x * scale[c] + shift[c]) (the forward's BatchNorm + ReLU prologue, one affine for all rows), applied to the
// registers at consume time -- bot_aspp's 1280 -> 256 weight gradient (deepv3.py:235-240 reads the BN+ReLU of the five ASPP branches)
template <int KB, bool AFFINE = false>
__global__ __launch_bounds__(256, KB == 4 ? 1 : 2) void gemm_tn_direct_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                float* __restrict__ out, int P, int M, int K, int C, long long a_bs,
                                                                long long b_bs, int Kpad, int Cp, int ktiles, int ctiles, int splits,
                                                                int tps, long long total, long long full, float* __restrict__ tail_ws,
                                                                const float* __restrict__ scale = nullptr,
                                                                const float* __restrict__ shift = nullptr, int relu = 0, int lda = 0) {
  typedef typename std::conditional<KB == 4, f32x4, f32x2>::type avec;
  if (lda <= 0) lda = K;               // row stride of A (dy): larger when dy is a channel slice of a wider buffer
  constexpr int TND = KB == 4 ? 8 : 5;                         // row pairs per register block (two blocks: one consumed, one in flight)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long job = (long long)blockIdx.x * 4 + wave;
  if (job >= total) return;                                   // no barrier anywhere in this kernel
  // Two job layouts. full < 0: every (position, tile) is cut into `splits` row ranges (slab per split in `out`, reduced afterwards).
  // full >= 0 (TAIL plan, r04: more tiles than wave slots and not a multiple of them -- 36 x 2 x 32 = 2304 tiles on 1024 SIMDs are
  // 2.25 rounds): the first `full` jobs are whole tiles written straight to the result; the remaining tiles are cut into `splits`
  // row ranges each, so that the last round is 1/splits as long; their partial tiles go to tail_ws [split][tail tile][128][128].
  long long t = job;
  int sp = 0, nsp = 1;
  long long tail_tile = -1;
  if (full >= 0) {
    if (job >= full) {
      const long long ntail = (total - full) / splits;
      sp = (int)((job - full) / ntail);
      tail_tile = (job - full) - (long long)sp * ntail;
      t = full + tail_tile;
      nsp = splits;
    }
  } else {
    nsp = splits;
  }
  const int ct = (int)(t % ctiles); t /= ctiles;
  const int kt = (int)(t % ktiles); t /= ktiles;
  int pb;
  if (full >= 0) pb = (int)t;
  else { sp = (int)(t % splits); pb = (int)(t / splits); }
  const int half = lane >> 5, j = lane & 31;
  const int r0 = nsp > 1 ? sp * tps : 0, r1 = nsp > 1 ? (r0 + tps < M ? r0 + tps : M) : M;
  const float* a = A + (size_t)pb * a_bs + (size_t)(kt * (32 * KB) + KB * j);
  const float* b = B + (size_t)pb * b_bs + (size_t)(ct * 128 + 4 * j);
  f32x16 acc[KB][4];
#pragma unroll
  for (int i = 0; i < KB; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
  // Two register blocks of TND row pairs each: while the 4 * KB * TND MFMAs of one block run, the 2 * TND loads of the other are in
  // flight. Rows past the end of the split take their A operand from a row of zeros (tn_zero_row) and their B operand from the last
  // valid row: no mask arithmetic on loaded values (a use at fetch time would wait on the load; a multiply at consume time costs 4
  // VALU + hazard nops per 16 MFMAs). The loop body handles both blocks in straight-line code, so no
  // register of the ring is ever copied while its load is pending; the scheduling fences keep hipcc from sinking the loads down to
  // their uses (it does: shorter live ranges).
  avec a0[TND], a1[TND];
  f32x4 b0[TND], b1[TND];
  const int last = r1 - 1;
  const float* az = tn_zero_row + (kt * (32 * KB) + KB * j);    // K <= 4096 (host check)
  auto fetch = [&](int row, avec& va, f32x4& vb) {
    const bool ok = row <= last;
    const size_t rr = (size_t)(ok ? row : last);
    va = *reinterpret_cast<const avec*>(ok ? a + rr * lda : az);  // a row past the end contributes A = 0: the product is zero
    vb = *reinterpret_cast<const f32x4*>(b + rr * C);
  };
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (AFFINE) {
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + ct * 128 + 4 * j);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + ct * 128 + 4 * j);
  }
  const float fl = relu ? 0.f : -__builtin_huge_valf();
  auto compute = [&](const avec (&va)[TND], const f32x4 (&vb)[TND]) {
#pragma unroll
    for (int d = 0; d < TND; ++d) {
      const avec ca = va[d];
      f32x4 cb = vb[d];
      if (AFFINE) {
        cb = cb * sc + sh;
        cb.x = fmaxf(cb.x, fl); cb.y = fmaxf(cb.y, fl); cb.z = fmaxf(cb.z, fl); cb.w = fmaxf(cb.w, fl);
      }
#pragma unroll
      for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i], cb[q], acc[i][q], 0, 0, 0);
    }
  };
  // (a mask-free main loop with a separate remainder loop was tried: a second loop that touches the accumulators makes hipcc keep
  // part of them in architectural registers and spill)
#pragma unroll
  for (int d = 0; d < TND; ++d) fetch(r0 + 2 * d + half, a0[d], b0[d]);
  for (int r = r0; r < r1; r += 4 * TND) {
#pragma unroll
    for (int d = 0; d < TND; ++d) fetch(r + 2 * TND + 2 * d + half, a1[d], b1[d]);
    __builtin_amdgcn_sched_barrier(0);
    compute(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < TND; ++d) fetch(r + 4 * TND + 2 * d + half, a0[d], b0[d]);
    __builtin_amdgcn_sched_barrier(0);
    compute(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* o;
  size_t ostride;
  if (tail_tile >= 0) {          // partial tile of the tail plan: compact [128][128]
    const long long ntail = (total - full) / splits;
    o = tail_ws + ((size_t)sp * ntail + tail_tile) * (128 * 128) + (size_t)(4 * j);
    ostride = 128;
  } else {
    o = out + ((size_t)(full >= 0 ? 0 : sp) * P + pb) * Kpad * Cp + (size_t)(kt * (32 * KB)) * Cp + (size_t)(ct * 128 + 4 * j);
    ostride = Cp;
  }
#pragma unroll
  for (int i = 0; i < KB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
      const f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      *reinterpret_cast<f32x4*>(o + (size_t)(KB * m + i) * ostride) = v;
    }
}

// tail plan: result tile = sum over splits (ascending) of its partial tiles; one workgroup per (tail tile, 16 rows)
__global__ __launch_bounds__(256) void tn_tail_reduce_kernel(const float* __restrict__ tail_ws, float* __restrict__ dwp, long long full,
                                                             long long ntail, int splits, int ktiles, int ctiles, int Kpad, int Cp) {
  const long long ti = blockIdx.x / 16;
  const int part = blockIdx.x % 16;
  long long t = full + ti;
  const int ct = (int)(t % ctiles); t /= ctiles;
  const int kt = (int)(t % ktiles); t /= ktiles;
  const int pb = (int)t;
  const int e = part * 1024 + threadIdx.x * 4;                       // element of the 128 x 128 tile (4 consecutive columns)
  const int row = e >> 7, col = e & 127;
  const float* src = tail_ws + (size_t)ti * (128 * 128) + e;
  f32x4 a = *reinterpret_cast<const f32x4*>(src);
  for (int sp = 1; sp < splits; ++sp) a += *reinterpret_cast<const f32x4*>(src + (size_t)sp * ntail * (128 * 128));
  *reinterpret_cast<f32x4*>(dwp + (size_t)pb * Kpad * Cp + (size_t)(kt * 128 + row) * Cp + ct * 128 + col) = a;
}

// ---- narrow weight gradients (K <= 64 output channels: the 19-channel heads, bot_fine's 48) without LDS (r04) -----------------------
// dW[k][c] = sum_r dy[r][k] * act(x[r][c]) with a handful of output channels is a STREAM over x (1.07 GB for the 256-channel head
// input at 2 x 512 x 1024) with 2 K FLOP per element: conv_wgrad_kernel<32 / 64, 128, 16> stages 16 pixels at a time through LDS
// and runs at 2.4-3.0 TB/s. Here, as in gemm_tn_direct_kernel, the operands go from global memory straight into the MFMA layout:
// lane (j, half) loads KB2 floats of dy row r + half (columns KB2 * j ..; lanes past K read a row of zeros) and 16 bytes of x
// (channels ct * 128 + 4 j ..), 4 * KB2 MFMAs per row pair, a wave owns a [32 * KB2] x 128 tile (64 / 128 accumulators), two or three
// waves per SIMD keep >= 16 KB per wave in flight. The BatchNorm + ReLU prologue of the forward (deepv3.py:235-252) is applied to the
// x registers at consume time. Partial tiles per row split in slabs, added in split order by wgrad_reduce_kernel: deterministic.
__device__ float tn_zeros_rt[64];          // zero-initialised and never written; NOT const, so that the compiler keeps `cond ? row : zeros`
                                           // a select of two addresses in front of ONE load (a const array of zeros folds to a branch around the load)
template <int KB2, bool AFFINE>
__global__ __launch_bounds__(256, 2) void gemm_tn_narrow_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int relu, float* __restrict__ out, int M, int K, int Kpad, int Cp,
                                                                int ctiles, int tps, long long total) {
  typedef typename std::conditional<KB2 == 2, f32x2, float>::type avec;
  constexpr int TND = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long job = (long long)blockIdx.x * 4 + wave;
  if (job >= total) return;
  const int ct = (int)(job % ctiles), sp = (int)(job / ctiles);
  const int half = lane >> 5, j = lane & 31;
  const int r0 = sp * tps, r1 = r0 + tps < M ? r0 + tps : M;
  const bool a_ok = KB2 * j + KB2 - 1 < K;
  const float* a = A + KB2 * j;
  const float* b = B + (size_t)(ct * 128 + 4 * j);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (AFFINE) {
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + ct * 128 + 4 * j);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + ct * 128 + 4 * j);
  }
  const float fl = relu ? 0.f : -__builtin_huge_valf();
  f32x16 acc[KB2][4];
#pragma unroll
  for (int i = 0; i < KB2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
  avec a0[TND], a1[TND];
  f32x4 b0[TND], b1[TND];
  const int last = r1 - 1;
  auto fetch = [&](int row, avec& va, f32x4& vb) {
    const bool ok = row <= last;
    const size_t rr = (size_t)(ok ? row : last);
    va = *reinterpret_cast<const avec*>((ok && a_ok) ? a + rr * lda : tn_zeros_rt);
    vb = *reinterpret_cast<const f32x4*>(b + rr * ldb);
  };
  auto compute = [&](const avec (&va)[TND], const f32x4 (&vb)[TND]) {
#pragma unroll
    for (int d = 0; d < TND; ++d) {
      f32x4 cb = vb[d];
      if (AFFINE) {
        cb = cb * sc + sh;
        cb.x = fmaxf(cb.x, fl); cb.y = fmaxf(cb.y, fl); cb.z = fmaxf(cb.z, fl); cb.w = fmaxf(cb.w, fl);
      }
#pragma unroll
      for (int i = 0; i < KB2; ++i) {
        float ai;
        if constexpr (KB2 == 2) ai = va[d][i]; else ai = va[d];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, cb[q], acc[i][q], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < TND; ++d) fetch(r0 + 2 * d + half, a0[d], b0[d]);
  for (int r = r0; r < r1; r += 4 * TND) {
#pragma unroll
    for (int d = 0; d < TND; ++d) fetch(r + 2 * TND + 2 * d + half, a1[d], b1[d]);
    __builtin_amdgcn_sched_barrier(0);
    compute(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < TND; ++d) fetch(r + 4 * TND + 2 * d + half, a0[d], b0[d]);
    __builtin_amdgcn_sched_barrier(0);
    compute(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* o = out + (size_t)sp * Kpad * Cp + (size_t)(ct * 128 + 4 * j);
#pragma unroll
  for (int i = 0; i < KB2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = KB2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + i;
      const f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      if (k < Kpad) *reinterpret_cast<f32x4*>(o + (size_t)k * Cp) = v;       // rows K .. Kpad-1: zeros (their A lanes read the zero row)
    }
}

struct NarrowPlan { int ctiles, splits, tps; long long total; };
inline bool narrow_shape_ok(const MssConvArgs& p, int Cp) {      // the part of the rule that needs no pointers (workspace query)
  if (MSS_ENV_INT("MSS_WGRAD_NARROW", 1) == 0) return false;
  if (p.R * p.S != 1 || p.stride != 1 || p.pad != 0 || p.batch > 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.K > 64 || (p.K > 32 && p.K % 2) || p.C % 128 || Cp != p.C || p.ldx % 4) return false;
  if ((p.in_scale || p.in_shift) && p.in_ss_stride != 0) return false;      // per-sample affines (Dropout2d folds): the LDS kernel
  return p.M >= 16384;                                                     // below that the launch is all ramp
}
inline bool narrow_eligible(const MssConvArgs& p, const float* dy, int lddy, int Cp) {
  if (!narrow_shape_ok(p, Cp)) return false;
  if (p.K > 32 && (lddy % 2 || (reinterpret_cast<uintptr_t>(dy) & 7))) return false;
  if (reinterpret_cast<uintptr_t>(p.x) & 15) return false;
  if ((p.in_scale && (reinterpret_cast<uintptr_t>(p.in_scale) & 15)) || (p.in_shift && (reinterpret_cast<uintptr_t>(p.in_shift) & 15))) return false;
  return true;
}
inline NarrowPlan narrow_plan(const MssConvArgs& p) {
  NarrowPlan pl;
  pl.ctiles = p.C / 128;
  int splits = (p.K <= 32 ? 3072 : 2048) / pl.ctiles;   // one round of the resident waves: 3 per SIMD (166 registers), 2 for the 64-row tile (248)
  const int max_splits = mss_cdiv(p.M, 512);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  pl.tps = mss_cdiv(mss_cdiv(p.M, splits), 2) * 2;
  pl.splits = mss_cdiv(p.M, pl.tps);
  pl.total = (long long)pl.ctiles * pl.splits;
  return pl;
}
inline long long narrow_ws_bytes(const MssConvArgs& p, int Cp) {
  if (!narrow_shape_ok(p, Cp)) return 0;
  const NarrowPlan pl = narrow_plan(p);
  return pl.splits > 1 ? (long long)pl.splits * p.Kpad * Cp * 4 : 0;
}
int launch_wgrad_narrow(const MssConvArgs& p, const float* dy, int lddy, float* dwp, int Cp, float* ws, long long ws_bytes,
                        hipStream_t stream) {
  const NarrowPlan pl = narrow_plan(p);
  const long long slab = (long long)p.Kpad * Cp;
  if (pl.splits > 1 && (!ws || ws_bytes < (long long)pl.splits * slab * 4)) return MSS_ERR_BAD_ARG;
  float* out = pl.splits > 1 ? ws : dwp;
  const bool aff = p.in_scale || p.in_shift || p.in_relu;
  const dim3 grid((unsigned)((pl.total + 3) / 4));
#define NARROW(KB2_, AFF_)                                                                                                          \
  hipLaunchKernelGGL((gemm_tn_narrow_kernel<KB2_, AFF_>), grid, dim3(256), 0, stream, dy, lddy, p.x, p.ldx, p.in_scale, p.in_shift,  \
                     p.in_relu, out, p.M, p.K, p.Kpad, Cp, pl.ctiles, pl.tps, pl.total)
  if (p.K <= 32) { if (aff) NARROW(1, true); else NARROW(1, false); }
  else { if (aff) NARROW(2, true); else NARROW(2, false); }
#undef NARROW
  if (pl.splits > 1) {
    launch_wgrad_reduce(ws, dwp, slab / 4, pl.splits, stream);
  }
  return mss_launch_status();
}

// ---- the TN route of the batched (Winograd-domain) weight gradient ----
struct TnPlan { int ktiles, ctiles, splits, tps; long long total; long long full = -1; };   // full >= 0: the tail plan of gemm_tn_direct_kernel
inline int tn_batch(const MssConvArgs& p) { return p.batch > 1 ? p.batch : 1; }
inline long long tn_tail_bytes(const TnPlan& pl) { return pl.full >= 0 ? (pl.total - pl.full) * (128ll * 128 * 4) : 0; }
inline bool tn_direct(const MssConvArgs& p);
inline int tn_mode();
inline bool tn_eligible(const MssConvArgs& p, int lddy) {
  const bool off = MSS_ENV_INT("MSS_WGRAD_TN", 5) == 0;     // A/B switch
  if (off || p.R * p.S != 1 || p.K % 4 || p.C % 4 || p.ldx != p.C || lddy < p.K || lddy % 4) return false;
  if (lddy != p.K && (p.batch > 1 || !tn_direct(p) || tn_mode() == 6)) return false;      // a slice of a wider dy: the LDS-free kernel takes a row stride
  if (p.in_scale || p.in_shift || p.in_relu) {
    // a prologue on x: only the LDS-free kernel applies one (a single affine for all rows, 16-byte aligned vectors), one position
    if (p.batch > 1 || p.in_ss_stride != 0 || !tn_direct(p) || MSS_ENV_INT("MSS_WGRAD_TN_AFFINE", 1) == 0) return false;
    if ((p.in_scale && (reinterpret_cast<uintptr_t>(p.in_scale) & 15)) || (p.in_shift && (reinterpret_cast<uintptr_t>(p.in_shift) & 15))) return false;
  }
  if (p.batch > 1) return p.x_bs % 4 == 0 && p.y_bs % 4 == 0 && p.N == 1 && p.H == 1;   // Winograd-domain products
  // a plain 1x1 / stride-1 layer over dense rows (ASPP 4096 -> 256: 95 -> see DESIGN 3.3): the same GEMM with one position;
  // narrow outputs (<= 64 channels: bot_fine, the heads) keep conv_wgrad_kernel's 64- / 32-row tiles
  return p.stride == 1 && p.pad == 0 && p.K >= 128 && p.C >= 128 && p.OH == p.H && p.OW == p.W;
}
inline TnPlan tn_plan(const MssConvArgs& p, int bc = TN_BC, int slots = 768) {
  TnPlan pl;
  pl.ktiles = mss_cdiv(p.K, TN_BK); pl.ctiles = mss_cdiv(p.C, bc);
  const long long base = (long long)tn_batch(p) * pl.ktiles * pl.ctiles;
  int max_splits = mss_cdiv(p.M, TN_BT * 8);
  // one position with few output tiles and very many rows (the decoder's Linear layers: 162 624 tokens x 256 -> 256 is 4 tiles):
  // 64 splits would fill a third of the slots
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
  pl.tps = mss_cdiv(mss_cdiv(p.M, splits), TN_BT) * TN_BT;
  pl.splits = mss_cdiv(p.M, pl.tps);
  pl.total = base * pl.splits;
  return pl;
}
// MSS_WGRAD_TN: 5 (default, r04) the LDS-free gemm_tn_direct_kernel<4> wherever K and C are multiples of 128 -- measured against
// the kernels below (tools/bench_wgrad_tn.py): ASPP F(6x6) 64 x 2304 x 4096 -> 256 116.9 -> 121.9 TFLOP/s, F(4x4) 36 x 5184 118.1 ->
// 123.9, decoder F(6x6) 64 x 29412 x 256 -> 256 119.9 -> 131.4, the pixel decoder's Linears 162624 x 256 -> 256 / 1024 -> 256 /
// 256 -> 1024 107.9 / 121.5 / 121.8 -> 111.9 / 129.7 / 129.5, 1x1 65536 x 4096 -> 256 123.5 -> 132.6 -- when its one-wave jobs fill at
// least 3/4 of the SIMDs, and rule 3 elsewhere; 7: that kernel at any size (tests); 6: its 64 x 128-tile, two-waves-per-SIMD form
// (102-120: slower everywhere, kept for the tests). Forcing the older kernels:
// 0 the convolution-loader kernel, 1 gemm_tn_wgrad_kernel, 2 gemm_tn2_wgrad_kernel<128>, 3 (the round-3 default) tn2 with
// 256-wide c tiles where they need no pixel split and fill their rounds (C % 256 == 0 and at least 2 rounds of 512 slots) and
// gemm_tn_wgrad_kernel elsewhere (with unmasked loads it is 2 % ahead of tn2<128>: 119.5 / 122.2 against 117.3 / 119.9 TFLOP/s),
// 4 the wide kernel whenever C % 256 == 0 (tests)
inline int tn_mode() {
  return MSS_ENV_INT("MSS_WGRAD_TN", 5);
}
inline bool tn_wide(const MssConvArgs& p) {
  if (tn_mode() == 4) return p.C % 256 == 0;            // tests: the wide kernel at any size, pixel splits included
  if ((tn_mode() != 3 && tn_mode() < 5) || p.C % 256) return false;
  const TnPlan w = tn_plan(p, 256, 512);
  const double eff = (double)w.total / (double)(((w.total + 511) / 512) * 512);
  return w.splits == 1 && w.total >= 1024 && eff >= 0.9;
}
inline bool tn_direct(const MssConvArgs& p);
inline TnPlan tn_plan_direct(const MssConvArgs& p);
inline TnPlan tn_plan_for(const MssConvArgs& p) { return tn_direct(p) ? tn_plan_direct(p) : tn_wide(p) ? tn_plan(p, 256, 512) : tn_plan(p); }
inline TnPlan tn_plan_direct(const MssConvArgs& p);
inline bool tn_direct(const MssConvArgs& p) {
  const int mode = tn_mode();
  if ((mode != 5 && mode != 6 && mode != 7) || p.K % 128 || p.C % 128 || p.K > 4096) return false;
  if (mode != 5) return true;                                  // 6 / 7 (tests, A/B): the direct kernels at any size
  // one wave per job and at least 256 rows per split: a product with few rows (the pixel decoder at ONE image: 10 164 tokens x
  // 256 -> 256 is 4 tiles x 39 splits = 156 waves for 1024 SIMDs; forward + backward 8.4 -> 9.2 ms) keeps the workgroup-tile kernels
  const TnPlan pl = tn_plan_direct(p);
  const long long slots = tn_mode() == 6 ? 2048 : 1024;
  return pl.total * 4 >= slots * 3;
}
// plan of the LDS-free kernel: one WAVE per (position, split, tile); mode 5: 128 x 128 tiles, 1024 wave slots (one per SIMD);
// mode 6: 64 x 128 tiles, 2048 slots
inline TnPlan tn_plan_direct(const MssConvArgs& p) {
  TnPlan pl;
  const int kb = tn_mode() == 6 ? 64 : 128;
  pl.ktiles = p.K / kb; pl.ctiles = p.C / 128;
  const long long base = (long long)tn_batch(p) * pl.ktiles * pl.ctiles;
  const int slots = tn_mode() == 6 ? 2048 : 1024;
  int max_splits = mss_cdiv(p.M, 256);
  // more tiles than slots, and the last round mostly empty: whole tiles for the full rounds, the rest cut so that they fill one
  // short round (the ASPP F(4x4) product: 2304 tiles = 2048 whole + 256 x 4 quarter jobs; only the 256 tail tiles are reduced)
  if (tn_mode() != 6 && MSS_ENV_INT("MSS_WGRAD_TN_TAIL", 1) != 0 && base > slots && base % slots != 0 &&
      (double)base / (double)(((base + slots - 1) / slots) * slots) < 0.95) {
    const long long tail = base % slots;
    int ts = (int)(slots / tail);
    if (ts > max_splits) ts = max_splits;
    if (ts > 16) ts = 16;
    if (ts >= 2) {
      pl.full = base - tail;
      pl.tps = mss_cdiv(mss_cdiv(p.M, ts), 2) * 2;
      pl.splits = mss_cdiv(p.M, pl.tps);
      pl.total = pl.full + tail * pl.splits;
      return pl;
    }
  }
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
  pl.tps = mss_cdiv(mss_cdiv(p.M, splits), 2) * 2;
  pl.splits = mss_cdiv(p.M, pl.tps);
  pl.total = base * pl.splits;
  return pl;
}
int launch_wgrad_tn(const MssConvArgs& p, const float* dy, float* dwp, int Cp, float* ws, long long ws_bytes,
                    hipStream_t stream, int lddy = 0) {
  if (tn_direct(p)) {
    const TnPlan pl = tn_plan_direct(p);
    const int P = tn_batch(p);
    const long long a_bs = p.batch > 1 ? p.y_bs : 0, b_bs = p.batch > 1 ? p.x_bs : 0;
    const long long slab = (long long)P * p.Kpad * Cp;
    if (pl.full >= 0) {
      const long long ntail = (pl.total - pl.full) / pl.splits;
      if (!ws || ws_bytes < tn_tail_bytes(pl)) return MSS_ERR_BAD_ARG;
      if (p.in_scale || p.in_shift || p.in_relu)
        hipLaunchKernelGGL((gemm_tn_direct_kernel<4, true>), dim3((unsigned)((pl.total + 3) / 4)), dim3(256), 0, stream, dy, p.x, dwp, P, p.M,
                           p.K, p.C, a_bs, b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total, pl.full, ws, p.in_scale,
                           p.in_shift, p.in_relu, lddy);
      else
        hipLaunchKernelGGL(gemm_tn_direct_kernel<4>, dim3((unsigned)((pl.total + 3) / 4)), dim3(256), 0, stream, dy, p.x, dwp, P, p.M, p.K,
                           p.C, a_bs, b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total, pl.full, ws, (const float*)nullptr,
                           (const float*)nullptr, 0, lddy);
      hipLaunchKernelGGL(tn_tail_reduce_kernel, dim3((unsigned)(ntail * 16)), dim3(256), 0, stream, ws, dwp, pl.full, ntail, pl.splits,
                         pl.ktiles, pl.ctiles, p.Kpad, Cp);
      return mss_launch_status();
    }
    if (pl.splits > 1 && (!ws || ws_bytes < (long long)pl.splits * slab * 4)) return MSS_ERR_BAD_ARG;
    float* out = pl.splits > 1 ? ws : dwp;
    if (tn_mode() == 6)
      hipLaunchKernelGGL(gemm_tn_direct_kernel<2>, dim3((unsigned)((pl.total + 3) / 4)), dim3(256), 0, stream, dy, p.x, out, P, p.M, p.K,
                         p.C, a_bs, b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total, -1ll, (float*)nullptr);
    else if (p.in_scale || p.in_shift || p.in_relu)
      hipLaunchKernelGGL((gemm_tn_direct_kernel<4, true>), dim3((unsigned)((pl.total + 3) / 4)), dim3(256), 0, stream, dy, p.x, out, P, p.M,
                         p.K, p.C, a_bs, b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total, -1ll, (float*)nullptr,
                         p.in_scale, p.in_shift, p.in_relu, lddy);
    else
      hipLaunchKernelGGL(gemm_tn_direct_kernel<4>, dim3((unsigned)((pl.total + 3) / 4)), dim3(256), 0, stream, dy, p.x, out, P, p.M, p.K,
                         p.C, a_bs, b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total, -1ll, (float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, 0, lddy);
    if (pl.splits > 1) {
      launch_wgrad_reduce(ws, dwp, slab / 4, pl.splits, stream);
    }
    return mss_launch_status();
  }
  const bool wide = tn_wide(p);
  const TnPlan pl = tn_plan_for(p);
  const int P = tn_batch(p);
  const long long a_bs = p.batch > 1 ? p.y_bs : 0, b_bs = p.batch > 1 ? p.x_bs : 0;
  const long long slab = (long long)P * p.Kpad * Cp;
  if (pl.splits > 1 && (!ws || ws_bytes < (long long)pl.splits * slab * 4)) return MSS_ERR_BAD_ARG;
  const size_t smem = (size_t)4 * TN_BT * TN_LD * sizeof(float);
  static int per_cu = 0, cus = 256;
  if (per_cu == 0) {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_tn_wgrad_kernel<false>, NT, smem) != hipSuccess || n < 1) n = 3;
    per_cu = n > 3 ? 3 : n;
  }
  const long long slots = (long long)(wide ? 2 : per_cu) * cus;
  const int grid = (int)(pl.total < slots ? pl.total : slots);
  float* out = pl.splits > 1 ? ws : dwp;
  const int mode = tn_mode();
  if (wide) {
    const size_t smem2 = (size_t)2 * (128 + 256) * TN2_LDK * sizeof(float);
    hipLaunchKernelGGL(gemm_tn2_wgrad_kernel<256>, dim3(grid), dim3(NT), smem2, stream, dy, p.x, out, P, p.M, p.K, p.C, a_bs,
                       b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total);
  } else if (mode == 2) {
    const size_t smem2 = (size_t)4 * 128 * TN2_LDK * sizeof(float);
    hipLaunchKernelGGL(gemm_tn2_wgrad_kernel<128>, dim3(grid), dim3(NT), smem2, stream, dy, p.x, out, P, p.M, p.K, p.C, a_bs,
                       b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total);
  } else {
    hipLaunchKernelGGL(gemm_tn_wgrad_kernel<false>, dim3(grid), dim3(NT), smem, stream, dy, p.x, out, P, p.M, p.K, p.C, a_bs,
                       b_bs, p.Kpad, Cp, pl.ktiles, pl.ctiles, pl.splits, pl.tps, pl.total);
  }
  if (pl.splits > 1) {
    launch_wgrad_reduce(ws, dwp, slab / 4, pl.splits, stream);
  }
  return mss_launch_status();
}

}  // namespace

void mss_wgrad_reduce_launch(const float* ws, float* dwp, long long slab4, int splits, hipStream_t stream) {
  launch_wgrad_reduce(ws, dwp, slab4, splits, stream);
}

extern "C" {

int mss_conv2d_forward_f32(MssConvArgs* args, void* stream) {
  MssConvArgs p = *args;
  if (!p.x || !p.w || !p.y) return MSS_ERR_BAD_ARG;
  if (p.C % 16 || p.ldx % 4 || p.R * p.S > 9 || p.R * p.S < 1) return MSS_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w)) & 15) return MSS_ERR_BAD_ARG;
  if (p.in_scale && ((reinterpret_cast<uintptr_t>(p.in_scale) | reinterpret_cast<uintptr_t>(p.in_shift)) & 15))
    return MSS_ERR_BAD_ARG;
  if (p.batch > 1 && (p.res || p.stats || p.x_bs % 4 || p.w_bs % 4 || p.batch > 65535)) return MSS_ERR_BAD_ARG;
  if (p.res_mask && !p.res) return MSS_ERR_BAD_ARG;
  p.M = p.N * p.OH * p.OW;
  if (p.M <= 0) return MSS_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // read per call (not cached): the parity tests switch routes inside one process (MSS_GEMM=0: every layer on the
  // implicit-GEMM kernel; tests/test_gpu_fullsize.py compares it with the GEMM/Winograd routes)
  const int use_gemm = MSS_ENV_INT("MSS_GEMM", 1);
  if (use_gemm) {
    const int rc = mss_gemm_nt_dispatch(p, stream);
    if (rc >= 0) return rc;
  }
  if (mss_conv_bf16x3_eligible(p)) return mss_conv_bf16x3_launch(p, stream);     // the split-bf16 route (args->w_split): gemm_bf16x3.hip, CONV
  // K-step: 16 (41 KB LDS, 144 registers -> 3 workgroups/CU, 3 waves/SIMD) is the faster choice except
  // for the ASPP shape (4096 input channels, 256 output channels), where the 32-deep step wins
  // (measured: 128 vs 119 TFLOP/s).
  const bool k32 = p.C % 32 == 0 && p.C >= 2048 && p.K <= 256;
  if (p.K <= 64) {
    (void)k32;
    return launch_conv<256, 64, 16, 4, 1>(p, s);
  }
  return k32 ? launch_conv<128, 128, 32, 2, 2>(p, s) : launch_conv<128, 128, 16, 2, 2>(p, s);
}

// Which kernel mss_conv2d_forward_f32 runs for these arguments: 1 = gemm_nt_kernel (gemm.hip), 0 = conv_igemm_kernel, 2 = gemm_few_rows_kernel,
// 3 = gemm_nt_bf16x3_kernel, 4 = its implicit-GEMM / per-sample-affine instantiations (args->w_split set).
int mss_conv2d_forward_route(const MssConvArgs* args) {
  MssConvArgs p = *args;
  p.M = p.N * p.OH * p.OW;
  if (!MSS_ENV_INT("MSS_GEMM", 1)) return mss_conv_bf16x3_eligible(p) ? 4 : 0;      // as the forward: MSS_GEMM=0 only skips the NT dispatch
  if (mss_gemm_few_rows(p)) return 2;                  // (0 implicit-GEMM kernel, 1 gemm_nt_kernel, 2 gemm_few_rows_kernel)
  if (!mss_gemm_nt_eligible(p)) return mss_conv_bf16x3_eligible(p) ? 4 : 0;
  p.mtiles = (p.M + 127) / 128;
  return mss_gemm_nt_bf16x3_eligible(p) ? 3 : 1;
}

// Kpad the packed layout must use for a conv with K output channels (multiple of the N tile).
int mss_conv2d_kpad(int K) { return K <= 64 ? 64 : ((K + 127) / 128) * 128; }

int mss_conv2d_pack_weights_f32(const float* w, float* packed, int K, int C, int R, int S, int Kpad, int Cp,
                                int flip, void* stream) {
  if (!w || !packed) return MSS_ERR_BAD_ARG;
  const size_t total = (size_t)R * S * Kpad * Cp;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     packed, K, C, R, S, Kpad, Cp, flip);
  return mss_launch_status();
}

int mss_conv2d_unpack_wgrad_f32(const float* packed, float* grad, int K, int C, int R, int S, int Kpad, int Cp,
                                int accumulate, void* stream) {
  if (!grad || !packed) return MSS_ERR_BAD_ARG;
  const size_t total = (size_t)K * C * R * S;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), packed,
                     grad, K, C, R, S, Kpad, Cp, accumulate);
  return mss_launch_status();
}

// K = 128 j + r output channels with 0 < r <= 64 over many pixels (the pixel decoder's merged 288-wide projection = 256 + 32: as ONE
// product its third 128-row tile is 3/4 padding, 71 - 79 TFLOP/s): the first 128 j channels as one product on the wide kernels and
// the last r on the narrow streaming kernel, each writing its own rows of dwp. Returns the wide part's channel count, 0 = no split.
static int wgrad_wide_part(const MssConvArgs& p, int Cp) {
  const int r = p.K % 128;
  if (p.batch > 1 || p.K <= 128 || r == 0 || r > 64 || Cp != p.C || p.Kpad < p.K) return 0;
  MssConvArgs t = p;
  t.K = r; t.Kpad = p.Kpad - (p.K - r);
  return narrow_shape_ok(t, Cp) && (r <= 32 || r % 2 == 0) ? p.K - r : 0;
}

// 1 when mss_conv2d_wgrad_f32 evaluates these arguments with the split-bf16 TN kernel (args->route == 1 and the shape eligible), else 0.
int mss_conv2d_wgrad_route(const MssConvArgs* args, int lddy) {
  MssConvArgs p = *args;
  p.M = p.N * p.OH * p.OW;
  if (p.M <= 0) return 0;
  const int wide = wgrad_wide_part(p, p.C);
  if (wide) p.K = p.Kpad = wide;
  return p.Kpad == p.K && mss_wgrad_tn_bf16x3_eligible(p, lddy) ? 1 : 0;     // (and Cp == C at the call)
}

// Bytes of scratch mss_conv2d_wgrad_f32 needs for these arguments (0: the pixel range is not split).
long long mss_conv2d_wgrad_workspace_bytes(const MssConvArgs* args, int Cp) {
  MssConvArgs p = *args;
  p.M = p.N * p.OH * p.OW;
  if (p.M <= 0) return 0;
  if (const int wide = wgrad_wide_part(p, Cp)) {           // both parts run one after the other on the same scratch
    MssConvArgs a = p, b = p;
    a.K = a.Kpad = wide;
    b.K = p.K - wide; b.Kpad = p.Kpad - wide;
    const long long wa = mss_conv2d_wgrad_workspace_bytes(&a, Cp), wb = mss_conv2d_wgrad_workspace_bytes(&b, Cp);
    // ... or, when the narrow part's pointers turn out misaligned at launch, the unsplit product instead: enough for that too
    // (the fall-back to `whole` used to fail with BAD_ARG on the smaller scratch)
    const long long parts = wa > wb ? wa : wb;
    long long whole = wgrad_ws_bytes<128, 128, 16>(p, Cp);
    if (tn_eligible(p, p.K)) {
      const TnPlan pl = tn_plan_for(p);
      const long long tb = pl.full >= 0 ? tn_tail_bytes(pl) : pl.splits > 1 ? (long long)pl.splits * tn_batch(p) * p.Kpad * Cp * 4 : 0;
      if (tb > whole) whole = tb;
    }
    return parts > whole ? parts : whole;
  }
  long long tn_bytes = 0;
  if (Cp == p.C && p.Kpad == p.K && mss_wgrad_tn_bf16x3_eligible(p, p.K)) return mss_wgrad_tn_bf16x3_ws_bytes(p, Cp);   // args->route == 1: the split-bf16 TN kernel
  if (tn_eligible(p, p.K)) {                 // lddy == K is assumed here and checked again at launch
    const TnPlan pl = tn_plan_for(p);
    tn_bytes = pl.full >= 0 ? tn_tail_bytes(pl) : pl.splits > 1 ? (long long)pl.splits * tn_batch(p) * p.Kpad * Cp * 4 : 0;
    if (p.batch > 1) return tn_bytes;        // Winograd-domain products always have lddy == K
  }
  // a plain 1x1 layer whose dy is a channel slice of a wider buffer falls back to conv_wgrad_kernel at launch: enough for both
  long long cw;
  const long long nb = narrow_ws_bytes(p, Cp);     // eligibility depends on pointers known only at launch: enough for either route
  if (nb > tn_bytes) tn_bytes = nb;
  if (p.K <= 32) cw = wgrad_ws_bytes<32, 128, 16>(p, Cp);
  else if (p.K <= 64) cw = wgrad_ws_bytes<64, 128, 16>(p, Cp);
  else cw = wgrad_ws_bytes<128, 128, 16>(p, Cp);
  return cw > tn_bytes ? cw : tn_bytes;
}

// dwp ([R*S][Kpad][Cp], Kpad >= K, Cp >= C multiples of 4) is fully overwritten (padding = 0); args
// describes the *forward* conv (x, geometry, optional prologue on x); dy is the NHWC output gradient with pixel
// stride lddy; ws: scratch of mss_conv2d_wgrad_workspace_bytes bytes (may be NULL when that is 0), contents
// irrelevant on entry. Deterministic: no atomics, fixed summation order.
int mss_conv2d_wgrad_f32(MssConvArgs* args, const float* dy, int lddy, float* dwp, int Cp, float* ws,
                         long long ws_bytes, void* stream) {
  MssConvArgs p = *args;
  if (!p.x || !dy || !dwp) return MSS_ERR_BAD_ARG;
  if (p.C % 4 || p.ldx % 4 || lddy % 4 || p.R * p.S > 9 || Cp % 4) return MSS_ERR_UNSUPPORTED;
  p.M = p.N * p.OH * p.OW;
  if (p.M <= 0) return MSS_OK;
  if (p.batch > 1 && (p.R * p.S != 1 || p.batch > 65535)) return MSS_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int wide = wgrad_wide_part(p, Cp)) {
    MssConvArgs a = *args, b = *args;
    a.K = a.Kpad = wide;
    b.K = p.K - wide; b.Kpad = p.Kpad - wide;
    b.route = 0;                                           // (the narrow kernel is a streaming kernel: nothing to split)
    b.M = p.M;
    if (!narrow_eligible(b, dy + wide, lddy, Cp)) goto whole;      // pointer alignment, known only here
    int rc = mss_conv2d_wgrad_f32(&a, dy, lddy, dwp, Cp, ws, ws_bytes, stream);
    if (rc != MSS_OK) return rc;
    return mss_conv2d_wgrad_f32(&b, dy + wide, lddy, dwp + (size_t)wide * Cp, Cp, ws, ws_bytes, stream);
  }
whole:
  // (whole 128 x 256 tiles only: a caller that pads dwp beyond K x C keeps the native kernels, which clear the padding)
  if (Cp == p.C && p.Kpad == p.K && mss_wgrad_tn_bf16x3_eligible(p, lddy) && ws_bytes >= mss_wgrad_tn_bf16x3_ws_bytes(p, Cp))
    return mss_wgrad_tn_bf16x3_launch(p, dy, lddy, dwp, Cp, ws, ws_bytes, stream);
  if (tn_eligible(p, lddy)) return launch_wgrad_tn(p, dy, dwp, Cp, ws, ws_bytes, s, lddy);
  if (narrow_eligible(p, dy, lddy, Cp)) return launch_wgrad_narrow(p, dy, lddy, dwp, Cp, ws, ws_bytes, s);
  // output-channel tile: 32 rows (1x4 waves) for the 19-channel heads, 64 for bot_fine's 48, else 128
  if (p.K <= 32) return launch_wgrad<32, 128, 16, 1>(p, dy, lddy, dwp, Cp, ws, ws_bytes, s);
  if (p.K <= 64) return launch_wgrad<64, 128, 16, 2>(p, dy, lddy, dwp, Cp, ws, ws_bytes, s);
  return launch_wgrad<128, 128, 16, 2>(p, dy, lddy, dwp, Cp, ws, ws_bytes, s);
}

}  // extern "C"
