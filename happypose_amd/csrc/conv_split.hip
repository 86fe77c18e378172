// fp32 convolution on the fp16 matrix path: every fp32 operand is split into two halves,
//   x = x_hi + x_lo,  x_hi = fp16(x),  x_lo = fp16(x - x_hi)          (22 significant bits),
// and a product is formed from three fp16 MFMAs accumulated in fp32,
//   x * w ~= x_hi * w_hi + x_hi * w_lo + x_lo * w_hi                   (x_lo * w_lo ~ 2^-22 dropped).
// v_mfma_f32_32x32x16_f16 does 16 K-steps in 32 cycles, v_mfma_f32_32x32x2_f32 two in 64: three of the
// former replace eight of the latter, 5.3x less matrix-pipe time for the same contraction at fp32-level
// accuracy (measured against fp64 in tests/test_gpu_kernels.py::test_conv3x3_kernel_families: the same
// bound as the exact-fp32 direct kernels).  Replaces the same reference ops as conv.hip (ATen conv +
// BatchNorm + ReLU [+ identity], MP/models/torchvision_resnet.py:110-126, MP/models/wide_resnet.py:59-65)
// for the 3x3 / stride-1 / pad-1 layers; activations stay fp32 NHWC in HBM, so the rest of the network
// plan (residuals, max-pool, heads, the other conv kernels) is unchanged.
//
// What makes the split exact enough:
//   * gfx950's f16 MFMA honours fp16 subnormals (tools/probes/mfma_f16_denorm.hip), so the low half of a
//     small activation is kept down to 2^-24: |x - x_hi - x_lo| <= max(2^-22 |x|, 2^-25);
//   * weights are scaled per output channel by a power of two so that max |w| lands in [2^13, 2^14)
//     (plan time, exact) and the accumulator is scaled back in the epilogue (exact): their low halves are
//     normal numbers whatever the magnitude of the folded-BN weights;
//   * activations must stay below the fp16 range (|x| < 65504): larger values become inf and poison the
//     output visibly instead of silently losing accuracy (hp_conv_select_algo(HP_CONV_ALGO_DIRECT) or
//     HP_CONV_NO_SPLIT=1 selects the exact-fp32 kernels).
//
// Kernel = the patch-staged direct 3x3 scheme of conv_patch.hip / conv_f16.hip: a block owns 128 output
// pixels x BN output channels; per 32-channel chunk it stages the pixel range [m0 - W - 1, m0 + 127 + W + 1]
// ONCE (fp32 buffer loads; pre-activation BN + ReLU in fp32; split; LDS rows [32 hi | 32 lo] halves = 128 B,
// padded to 144 B) and forms the nine taps as row-shifted fragment reads with a per-lane border mask.
// Weights are pre-split at plan time into the same [32 hi | 32 lo] rows, ordered (chunk, tap), double
// buffered in LDS one tap ahead, their loads two taps ahead in alternating register sets.
// Per tap and wave (64x64 wave tile): 16 ds_read_b128 feed 24 MFMAs.
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "conv.h"
#include "conv_epilogue.h"
#include "conv_splitk.h"

namespace hp {

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 512;  // 8 waves: two per SIMD, ONE workgroup per CU
constexpr int CK = 32;         // channels per chunk
constexpr int LDH = 64 + 8;    // LDS row: 32 hi + 32 lo halves, padded to 36 dwords
constexpr unsigned kOob = 0xFFFFFFF0u;

__device__ __forceinline__ floatx4 loadf4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ halfx8 loadh8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(halfx8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// ---- plan time: fp32 packed weights [Cout][9 * Cin] (K = (tap, c), BN folded) -> split rows
// [Cout][Cin / 32 chunks][9 taps][32 hi | 32 lo] halves + per-cout scale-back factors [Cout] (fp32) behind them
__global__ __launch_bounds__(256) void split_weights_kernel(const float* w, _Float16* ws, float* unscale, int cin, int Kpad) {
  const int o = blockIdx.x, K = 9 * cin;
  const float* row = w + (size_t)o * Kpad;
  __shared__ float red[256];
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(row[k]));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);  // mx = m 2^e, m in [0.5, 1)
  const int s = mx > 0.f ? 14 - e : 0;                  // mx 2^s in [2^13, 2^14)
  if (threadIdx.x == 0) unscale[o] = ldexpf(1.f, -s);
  _Float16* out = ws + (size_t)o * K * 2;
  for (int k = threadIdx.x; k < K; k += 256) {
    const int tap = k / cin, c = k - tap * cin, cc = c / CK, j = c - cc * CK;
    const float v = ldexpf(row[k], s);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    _Float16* d = out + ((size_t)(cc * 9 + tap)) * 64;
    d[j] = hi;
    d[32 + j] = lo;
  }
}

// Block tile BM x BN = (64 WAVES_M) x (64 WAVES_N), WAVES_M * WAVES_N = 8, every wave a 64 x 64 tile:
// 256 x 128 for the layers with >= 128 output channels, 512 x 64 for the 64-channel ones.
// NPC: 128-row passes of the patch staging (8 channels = two 16-B loads per thread and pass).
template <int WAVES_M, int WAVES_N>
struct SplitTile {
  static constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N;
  static int P(int W) { return BM + 2 * W + 2; }
  static constexpr int NTHR = 64 * WAVES_M * WAVES_N;  // 8 waves: one workgroup per CU; 4 waves: two
  static int npc(int W) { return (P(W) + NTHR / 4 - 1) / (NTHR / 4); }
  static size_t lds_bytes(int W) {
    const size_t loop = ((size_t)P(W) * LDH + 2 * (size_t)BN * LDH + LDH) * 2;  // patch, weights x 2, a zero row
    const size_t epi = (size_t)BM * (BN + 4) * 4;
    return loop < epi ? epi : loop;
  }
};

template <int WAVES_M, int WAVES_N, bool PRE, int NPC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_split_f32(ConvArgs a, int P) {
  static_assert(WAVES_M * WAVES_N == 8 || WAVES_M * WAVES_N == 4, "8 waves (one workgroup per CU) or 4 (two)");
  constexpr int kThreads = 64 * WAVES_M * WAVES_N;  // shadows the file-level constant
  constexpr int PROWS = kThreads / 4, BROWS = kThreads / 8;  // rows per staging pass: patch / weights
  constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N, MT = 2, NT = 2;
  constexpr int NB = BN * 8 / kThreads;  // 16-B weight chunks per thread and tap (2 or 1)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const patch = reinterpret_cast<_Float16*>(lds_raw);  // [P][LDH]
  _Float16* const Bs = patch + P * LDH;                            // [2][BN][LDH]
  _Float16* const zrow = Bs + 2 * BN * LDH;                        // [LDH] zeros: what a masked tap reads
  float act_sx, act_inv;  // ConvArgs::amax_in: power-of-two scale of the staged activations and its inverse
  conv_act_scale(a, act_sx, act_inv);

  // work item = a whole tile, or a (tile, K slice) of the last, partially filled round (conv_splitk.h): one
  // workgroup per CU makes a partial round as long as a full one, so its tiles are cut along K into slices
  // that fill the CUs and meet through the "last arriver reduces" protocol (deterministic slice order)
  int lin, slice;
  bool split;
  if (!splitk_decode(a, lin, slice, split)) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H, Cin = a.Cin;
  const int ncc_all = Cin / CK, ntaps = ncc_all * 9;
  const int cc_begin = split ? slice * ncc_all / a.sk_S : 0;
  const int ncc = split ? (slice + 1) * ncc_all / a.sk_S : ncc_all;  // end of this item's chunk range

  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(a.M * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((size_t)a.Cout * 9 * Cin * 4), 0x00020000);

  // patch staging: row = pr0 + 128 j, channels 8 pk .. 8 pk + 7 of the chunk
  const int pk = tid & 3, pr0 = tid >> 2;
  const int64_t gp0 = m0 - (W + 1) + pr0;
  auto patch_voff = [&](int j) -> unsigned {
    const int64_t gp = gp0 + PROWS * j;
    return (pr0 + PROWS * j < P && gp >= 0 && gp < a.M) ? (unsigned)((gp * Cin + 8 * pk) * 4) : kOob;
  };
  _Float16* const Pst = patch + pr0 * LDH + 8 * pk;
  // weight staging: row = br0 + 64 i, 16-B chunk bk of the 128-B row
  const int bk = tid & 7, br0 = tid >> 3;
  unsigned wvoff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(n0 + br0 + BROWS * i) * (18 * Cin) + 8 * bk) * 2);
  _Float16* const Bst = Bs + br0 * LDH + 8 * bk;
  if (tid < LDH / 2) reinterpret_cast<unsigned*>(zrow)[tid] = 0u;

  // fragment bases + validity of the 9 taps per fragment row
  const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * 64;
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  const _Float16* const Bfr = Bs + (wn + frow) * LDH + fk;
  const _Float16* Afr[MT];
  unsigned vmask[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    Afr[mt] = patch + (wm + mt * 32 + frow + W + 1) * LDH + fk;
    const int64_t g = m0 + wm + mt * 32 + frow;
    unsigned mk = 0;
    if (g < a.M) {
      const int rem = (int)g - fdiv((int)g, a.fd_howo) * (H * W);  // stride 1: Ho x Wo = H x W
      const int oh = fdiv(rem, a.fd_wo), ow = rem - oh * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
        mk |= ((((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W)) ? 1u : 0u) << t;
      }
    }
    vmask[mt] = mk;
  }
  const _Float16* const Zfr = zrow + fk;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  floatx4 pr[NPC][2];  // next channel chunk of the patch (fp32)
  halfx8 rb[2][NB];    // weights of taps t+1 / t+2 (alternating sets)
  auto load_patch = [&](int j, int cc) {
    const unsigned vo = patch_voff(j);
    pr[j][0] = loadf4(xrsrc, vo, (unsigned)(cc * CK * 4));
    pr[j][1] = loadf4(xrsrc, vo, (unsigned)(cc * CK * 4 + 16));
  };
  auto store_patch = [&](int cc) {
    floatx4 ps[2], pb[2];
    if (PRE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // the activation scale rides on the prologue: sx relu(x s + b) = relu(x (s sx) + b sx), exactly
        ps[h] = *reinterpret_cast<const floatx4*>(a.pre_scale + cc * CK + 8 * pk + 4 * h) * act_sx;
        pb[h] = *reinterpret_cast<const floatx4*>(a.pre_shift + cc * CK + 8 * pk + 4 * h) * act_sx;
      }
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      if (pr0 + PROWS * j < P) {
        const bool real = patch_voff(j) != kOob;  // pixels outside the tensor stay zero
        halfx4 hi[2], lo[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          floatx4 v = pr[j][h];
          if (PRE) {
            v = __builtin_elementwise_max(v * ps[h] + pb[h], floatx4{0.f, 0.f, 0.f, 0.f});
            if (!real) v = floatx4{0.f, 0.f, 0.f, 0.f};
          } else {
            v = v * act_sx;
          }
          hi[h] = __builtin_convertvector(v, halfx4);
          lo[h] = __builtin_convertvector(v - __builtin_convertvector(hi[h], floatx4), halfx4);
        }
        *reinterpret_cast<halfx8*>(Pst + PROWS * j * LDH) = __builtin_shufflevector(hi[0], hi[1], 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<halfx8*>(Pst + PROWS * j * LDH + 32) = __builtin_shufflevector(lo[0], lo[1], 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  };
  auto load_b = [&](int set, int tt) {  // tt = cc * 9 + tap: the order of the split weight rows
    const int t2 = tt < ntaps ? tt : ntaps - 1;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[set][i] = loadh8(wrsrc, wvoff[i], (unsigned)(t2 * 128));
  };
  auto store_b = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<halfx8*>(Bst + buf * BN * LDH + BROWS * i * LDH) = rb[set][i];
  };

  // prologue: patch of chunk 0, weights of tap 0 (-> LDS), 1 and 2 (-> registers)
#pragma unroll
  for (int j = 0; j < NPC; ++j) load_patch(j, cc_begin);
  load_b(0, cc_begin * 9);
  load_b(1, cc_begin * 9 + 1);
  store_patch(cc_begin);
  store_b(0, 0);
  load_b(0, cc_begin * 9 + 2);
  __syncthreads();

#ifndef HP_SPLIT_OLD_TAP
  // ---- software-pipelined tap body.  A tap is six groups of MT x NT MFMAs (hi.hi, hi.lo, lo.hi for each of the two
  // 16-channel k-steps).  The fragments of group g + 1 are read BEFORE the MFMAs of group g are issued, the first
  // group's fragments of the NEXT tap before the last group of this one, and the per-tap barrier sits between groups
  // 4 and 5 (every B read of the tap has completed by then: the weight buffer may be overwritten, and the buffer of
  // the next tap -- stored at the head of this one -- is visible).  With both waves of a SIMD in lockstep behind the
  // barrier, an exposed LDS round trip at the head of every tap and one in its middle idled the matrix pipe for
  // 15-25 % of the tap (the compiler clustered the reads in front of their first use); sched_barrier pins the order.
  halfx8 pfa[MT], pfb[NT];  // q0 fragments of the tap about to run (prefetched)
  auto a_base = [&](int tap, const _Float16* (&Ab)[MT]) {
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) Ab[i] = ((vmask[i] >> tap) & 1u) ? Afr[i] + d * LDH : Zfr;
  };
  auto rd_a = [&](halfx8 (&f)[MT], const _Float16* const (&Ab)[MT], int q) {
#pragma unroll
    for (int i = 0; i < MT; ++i) f[i] = *reinterpret_cast<const halfx8*>(Ab[i] + q * 16);
  };
  auto rd_b = [&](halfx8 (&f)[NT], int buf, int q) {
#pragma unroll
    for (int i = 0; i < NT; ++i) f[i] = *reinterpret_cast<const halfx8*>(Bfr + buf * BN * LDH + i * 32 * LDH + q * 16);
  };
  auto mm = [&](const halfx8 (&fa)[MT], const halfx8 (&fb)[NT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
  };
#define HP_PIN() __builtin_amdgcn_sched_barrier(0)
  auto prefetch_first = [&](int tap, int buf) {  // q0 fragments of `tap` (weights in LDS buffer `buf`)
    const _Float16* Ab[MT];
    a_base(tap, Ab);
    rd_a(pfa, Ab, 0);
    rd_b(pfb, buf, 0);
  };
  auto tap_step = [&](int tt, int cc, int tap, auto par) {
    constexpr int Pb = decltype(par)::value;  // tt & 1: LDS weight buffer of this tap
    const _Float16* Ab[MT];
    a_base(tap, Ab);
    const bool next_chunk = cc + 1 < ncc;
    const bool swap = tap == 8 && next_chunk;  // the patch is replaced after this tap
    halfx8 ah[MT], bh[NT], bl[NT], al[MT], ah1[MT], bh1[NT], bl1[NT], al1[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ah[i] = pfa[i];
#pragma unroll
    for (int i = 0; i < NT; ++i) bh[i] = pfb[i];
    rd_b(bl, Pb, 2);                       // group 1
    store_b(1 - Pb, 1 - Pb);               // weights of tap tt+1 (register set (tt+1) & 1) ...
    load_b(1 - Pb, tt + 3);                // ... and that set takes tap tt+3
    HP_PIN();
    mm(ah, bh);                            // group 0: hi.hi, k-step 0
    HP_PIN();
    rd_a(al, Ab, 2);                       // group 2
    HP_PIN();
    mm(ah, bl);                            // group 1: hi.lo
    HP_PIN();
    rd_a(ah1, Ab, 1);                      // group 3
    rd_b(bh1, Pb, 1);
    HP_PIN();
    mm(al, bh);                            // group 2: lo.hi
    HP_PIN();
    rd_b(bl1, Pb, 3);                      // group 4
#pragma unroll
    for (int j = 0; j < NPC; ++j)          // next chunk's patch: one pass per tap (taps 0 .. NPC-1)
      if (j == tap) load_patch(j, cc + 1 < ncc ? cc + 1 : cc);
    HP_PIN();
    mm(ah1, bh1);                          // group 3: hi.hi, k-step 1
    HP_PIN();
    rd_a(al1, Ab, 3);                      // group 5
    HP_PIN();
    mm(ah1, bl1);                          // group 4: hi.lo
    HP_PIN();
    __syncthreads();                       // every wave has finished the B reads of this tap and stored tap tt+1's weights
    if (!swap) prefetch_first(tap == 8 ? 0 : tap + 1, 1 - Pb);
    else rd_b(pfb, 1 - Pb, 0);
    HP_PIN();
    mm(al1, bh1);                          // group 5: lo.hi
    HP_PIN();
    if (swap) {  // every wave is done with this chunk's patch: swap in the next one
      __syncthreads();
      store_patch(cc + 1);
      __syncthreads();
      const _Float16* An[MT];
      a_base(0, An);
      rd_a(pfa, An, 0);
    }
  };
#undef HP_PIN
#else
  auto tap_step = [&](int tt, int cc, int tap, auto par) {
    constexpr int Pb = decltype(par)::value;  // tt & 1: LDS weight buffer of this tap
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
    // rows whose shifted pixel lies across an image border read the zero row instead
    const _Float16* Ab[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#ifndef HP_SABL_NOMASK
      Ab[i] = ((vmask[i] >> tap) & 1u) ? Afr[i] + d * LDH : Zfr;
#else
      Ab[i] = Afr[i] + d * LDH;
#endif
    }
    const _Float16* Bb = Bfr + Pb * BN * LDH;
    const bool next_chunk = cc + 1 < ncc;
    // fragment q of a row: 0 / 1 = hi halves of channels 0-15 / 16-31, 2 / 3 = their lo halves
    auto read_a = [&](halfx8 (&f)[MT], int q) {
#pragma unroll
#ifdef HP_SABL_NOREAD
      for (int i = 0; i < MT; ++i) f[i] = __builtin_bit_cast(halfx8, floatx4{(float)tt, (float)q, 1.f, 2.f});
#else
      for (int i = 0; i < MT; ++i) f[i] = *reinterpret_cast<const halfx8*>(Ab[i] + q * 16);
#endif
    };
    auto read_b = [&](halfx8 (&f)[NT], int q) {
#pragma unroll
#ifdef HP_SABL_NOREAD
      for (int i = 0; i < NT; ++i) f[i] = __builtin_bit_cast(halfx8, floatx4{(float)tt, (float)q, 3.f, 4.f});
#else
      for (int i = 0; i < NT; ++i) f[i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH + q * 16);
#endif
    };
    auto mm = [&](const halfx8 (&fa)[MT], const halfx8 (&fb)[NT]) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    };
    // six groups of MT x NT MFMAs; a fragment set is read one group before its first use and is dead after
    // its second (hi) or only (lo) use: at most five sets are live
    halfx8 ah[MT], al[MT], bh[NT], bl[NT], ah1[MT], bh1[NT];
    // (the staging below is unconditional -- clamped addresses, a dead LDS buffer after the last tap -- so
    // that the tap body is straight-line code: with branches the compiler waits for ALL outstanding loads)
    read_a(ah, 0);
    read_b(bh, 0);
    read_b(bl, 2);
#ifndef HP_SABL_NOBSTAGE
#ifndef HP_SABL_NOBSTORE
    store_b(1 - Pb, 1 - Pb);               // weights of tap tt+1 (register set (tt+1) & 1)
#endif
#ifndef HP_SABL_NOBLOAD
    load_b(1 - Pb, tt + 3);                // ... and that set takes tap tt+3
#endif
#endif
#ifdef HP_SPLIT_2BATCH
    halfx8 al1[MT], bl1[NT];
    read_a(al, 2);
    mm(ah, bh);
    read_a(ah1, 1);
    read_b(bh1, 1);
    read_b(bl1, 3);
    read_a(al1, 3);
    mm(ah, bl);
    mm(al, bh);
#ifndef HP_SABL_NOPATCH
#pragma unroll
    for (int j = 0; j < NPC; ++j)          // next chunk's patch: one pass per tap (taps 0 .. NPC-1)
      if (j == tap) load_patch(j, cc + 1 < ncc ? cc + 1 : cc);
#endif
    mm(ah1, bh1);
    mm(ah1, bl1);
    mm(al1, bh1);
#else
    mm(ah, bh);
    read_a(al, 2);
    mm(ah, bl);
    read_a(ah1, 1);
    read_b(bh1, 1);
    mm(al, bh);
    read_b(bl, 3);
#ifndef HP_SABL_NOPATCH
#pragma unroll
    for (int j = 0; j < NPC; ++j)          // next chunk's patch: one pass per tap (taps 0 .. NPC-1)
      if (j == tap) load_patch(j, cc + 1 < ncc ? cc + 1 : cc);
#endif
    mm(ah1, bh1);
    read_a(al, 3);
    mm(ah1, bl);
    mm(al, bh1);
#endif
#ifndef HP_SABL_NOBARRIER
    __syncthreads();
#endif
#ifndef HP_SABL_NOPATCH
    if (tap == 8 && next_chunk) {  // every wave is done with this chunk's patch: swap in the next one
      store_patch(cc + 1);
      __syncthreads();
    }
#endif
  };
#endif
  // tap t of the item's chunk number rc uses weight buffer (rc + t) & 1: two chunks per loop iteration make
  // that a compile-time value
  auto chunk = [&](int cc, auto c0) {
    constexpr int C0 = decltype(c0)::value;
    using E = std::integral_constant<int, C0>;      // even taps
    using O = std::integral_constant<int, 1 - C0>;  // odd taps
    tap_step(cc * 9 + 0, cc, 0, E{}); tap_step(cc * 9 + 1, cc, 1, O{}); tap_step(cc * 9 + 2, cc, 2, E{});
    tap_step(cc * 9 + 3, cc, 3, O{}); tap_step(cc * 9 + 4, cc, 4, E{}); tap_step(cc * 9 + 5, cc, 5, O{});
    tap_step(cc * 9 + 6, cc, 6, E{}); tap_step(cc * 9 + 7, cc, 7, O{}); tap_step(cc * 9 + 8, cc, 8, E{});
  };
  int cc = cc_begin;
#ifndef HP_SPLIT_OLD_TAP
  prefetch_first(0, 0);
#endif
#ifdef HP_SABL_NOLOOP
  cc = ncc;
#endif
  for (; cc + 1 < ncc; cc += 2) {
    chunk(cc, std::integral_constant<int, 0>{});
    chunk(cc + 1, std::integral_constant<int, 1>{});
  }
  if (cc < ncc) chunk(cc, std::integral_constant<int, 0>{});

#ifdef HP_SPLIT_FENCED_SLABS
  if (split && !splitk_reduce<BM, BN, MT, NT, kThreads>(a, acc, lin - a.sk_regular, slice)) return;
#else
  if (split && !splitk_reduce_sc1<BM, BN, MT, NT, kThreads>(a, acc, lin - a.sk_regular, slice)) return;
#endif
#ifdef HP_SABL_NOEPI
  if (acc[0][0][0] != 12345.f) return;
#endif

  // ---- scale back (a lane holds one output channel per N tile), then the shared fp32 epilogue
  const float* const unscale = reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(a.w) + (size_t)a.Cout * 18 * Cin);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float s = unscale[n0 + wn + nt * 32 + (lane & 31)] * act_inv;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= s;
  }
  conv_epilogue<BM, BN, MT, NT, kThreads>(a, reinterpret_cast<float*>(lds_raw), acc, m0, n0, wm, wn);
}

// ---- 3x3 / stride-2 / pad-1 layers (the first conv of layer2 / 3 / 4) on the same scheme.  In the
// space-to-depth view of the input -- pixel (i, j) of phase (p, q) = input pixel (2i + p, 2j + q) -- a
// stride-2 3x3 filter is a stride-1 filter with taps at row shifts {0} (phase row 0: kh = 1) or {-1, 0}
// (phase row 1: kh = 0, 2), the same for columns: 1 / 2 / 2 / 4 taps for the phases (0,0) (0,1) (1,0) (1,1),
// nine in all.  The K loop walks (phase, 32-channel chunk, tap); the patch of a (phase, chunk) is the
// output-pixel-linear range [m0 - Wo - 1, m0 + BM - 1] gathered from the ordinary NHWC input (one 128-B
// run per pixel and chunk), so nothing upstream changes layout.  Weights are pre-split in loop order.
__device__ __forceinline__ void s2_entry(int e, int n, int& ph, int& c, int& t, int& T) {
  if (e < n) { ph = 0; c = e; t = 0; T = 1; }
  else if (e < 3 * n) { ph = 1; c = (e - n) >> 1; t = (e - n) & 1; T = 2; }
  else if (e < 5 * n) { ph = 2; c = (e - 3 * n) >> 1; t = (e - 3 * n) & 1; T = 2; }
  else { ph = 3; c = (e - 5 * n) >> 2; t = (e - 5 * n) & 3; T = 4; }
}

__device__ __forceinline__ void s2_tap(int ph, int t, int& di, int& dj, int& kh, int& kw) {
  const int p = ph >> 1, q = ph & 1;
  const int ti = (p && q) ? t >> 1 : t, tj = (p && q) ? t & 1 : t;
  di = p ? ti - 1 : 0; kh = p ? 2 * ti : 1;  // phase row 1 holds the input rows of kh = 0 (one row up) and kh = 2
  dj = q ? tj - 1 : 0; kw = q ? 2 * tj : 1;
}

__global__ __launch_bounds__(256) void split_weights_s2_kernel(const float* w, _Float16* ws, float* unscale, int cin, int Kpad) {
  const int o = blockIdx.x, K = 9 * cin, n = cin / CK;
  const float* row = w + (size_t)o * Kpad;
  __shared__ float red[256];
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(row[k]));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  int ex = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &ex);
  const int s = mx > 0.f ? 14 - ex : 0;
  if (threadIdx.x == 0) unscale[o] = ldexpf(1.f, -s);
  _Float16* out = ws + (size_t)o * K * 2;
  for (int idx = threadIdx.x; idx < K; idx += 256) {
    const int e = idx / CK, j = idx - e * CK;
    int ph, c, t, T;
    s2_entry(e, n, ph, c, t, T);
    int di, dj, kh, kw;
    s2_tap(ph, t, di, dj, kh, kw);
    const float v = ldexpf(row[(kh * 3 + kw) * cin + c * CK + j], s);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    out[(size_t)e * 64 + j] = hi;
    out[(size_t)e * 64 + 32 + j] = lo;
  }
}

template <int WAVES_M, int WAVES_N, bool PRE, int NPC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3s2_split_f32(ConvArgs a, int P) {
  static_assert(WAVES_M * WAVES_N == 8, "8 waves");
  constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N, MT = 2, NT = 2;
  constexpr int NB = BN * 8 / kThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const patch = reinterpret_cast<_Float16*>(lds_raw);  // [P][LDH]
  _Float16* const Bs = patch + P * LDH;                            // [2][BN][LDH]
  _Float16* const zrow = Bs + 2 * BN * LDH;
  float act_sx, act_inv;  // ConvArgs::amax_in (see conv3x3_split_f32)
  conv_act_scale(a, act_sx, act_inv);

  const int nblk = a.tiles_m * a.tiles_n;
  const int per_xcd = (nblk + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= nblk) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H, C = a.Cin, Wo = a.Wo, Ho = a.Ho;
  const int n = C / CK, nent = 9 * n;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((a.M / (Ho * Wo)) * (int64_t)H * W * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((size_t)a.Cout * 9 * C * 4), 0x00020000);

  // patch staging: row = pr0 + 128 j <-> output-grid pixel g = m0 - (Wo + 1) + row; phase (p, q) of it is
  // input pixel (2 oh + p, 2 ow + q)
  const int pk = tid & 3, pr0 = tid >> 2;
  unsigned pbase[NPC];  // byte offset of input pixel (2 oh, 2 ow), channel 8 pk
  unsigned pflag[NPC];  // bit 0: pixel inside the tensor, bit 1: row 2 oh + 1 exists, bit 2: column 2 ow + 1 exists
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int64_t g = m0 - (Wo + 1) + pr0 + 128 * j;
    pbase[j] = 0; pflag[j] = 0;
    if (pr0 + 128 * j < P && g >= 0 && g < a.M) {
      const int img = fdiv((int)g, a.fd_howo);
      const int rem = (int)g - img * (Ho * Wo);
      const int oh = fdiv(rem, a.fd_wo), ow = rem - oh * Wo;
      pbase[j] = (unsigned)(((((int64_t)img * H + 2 * oh) * W + 2 * ow) * C + 8 * pk) * 4);
      pflag[j] = 1u | (2 * oh + 1 < H ? 2u : 0u) | (2 * ow + 1 < W ? 4u : 0u);
    }
  }
  _Float16* const Pst = patch + pr0 * LDH + 8 * pk;
  const int bk = tid & 7, br0 = tid >> 3;
  unsigned wvoff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(n0 + br0 + 64 * i) * (18 * C) + 8 * bk) * 2);
  _Float16* const Bst = Bs + br0 * LDH + 8 * bk;
  if (tid < LDH / 2) reinterpret_cast<unsigned*>(zrow)[tid] = 0u;

  const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * 64;
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  const _Float16* const Bfr = Bs + (wn + frow) * LDH + fk;
  const _Float16* Afr[MT];
  unsigned vmask[MT];  // bit kh * 3 + kw: that tap of this output pixel is inside the image
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    Afr[mt] = patch + (wm + mt * 32 + frow + Wo + 1) * LDH + fk;
    const int64_t g = m0 + wm + mt * 32 + frow;
    unsigned mk = 0;
    if (g < a.M) {
      const int rem = (int)g - fdiv((int)g, a.fd_howo) * (Ho * Wo);
      const int oh = fdiv(rem, a.fd_wo), ow = rem - oh * Wo;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ih = 2 * oh + t / 3 - 1, iw = 2 * ow + t % 3 - 1;
        mk |= ((((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W)) ? 1u : 0u) << t;
      }
    }
    vmask[mt] = mk;
  }
  const _Float16* const Zfr = zrow + fk;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  floatx4 pr[NPC][2];
  halfx8 rb[2][NB];
  // (phase, chunk) number u = ph * n + c of the patch sequence
  auto patch_voff = [&](int j, int ph) -> unsigned {
    const unsigned need = 1u | ((ph & 2) ? 2u : 0u) | ((ph & 1) ? 4u : 0u);
    return (pflag[j] & need) == need ? pbase[j] : kOob;
  };
  auto load_patch = [&](int u) {
    const int ph = u / n, c = u - ph * n;
    const unsigned soff = (unsigned)((((ph >> 1) * W + (ph & 1)) * C + c * CK) * 4);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const unsigned vo = patch_voff(j, ph);
      pr[j][0] = loadf4(xrsrc, vo, soff);
      pr[j][1] = loadf4(xrsrc, vo, soff + 16);
    }
  };
  auto store_patch = [&](int u) {
    const int ph = u / n, c = u - ph * n;
    floatx4 ps[2], pb[2];
    if (PRE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ps[h] = *reinterpret_cast<const floatx4*>(a.pre_scale + c * CK + 8 * pk + 4 * h) * act_sx;
        pb[h] = *reinterpret_cast<const floatx4*>(a.pre_shift + c * CK + 8 * pk + 4 * h) * act_sx;
      }
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      if (pr0 + 128 * j < P) {
        const bool real = patch_voff(j, ph) != kOob;
        halfx4 hi[2], lo[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          floatx4 v = pr[j][h];
          if (PRE) {
            v = __builtin_elementwise_max(v * ps[h] + pb[h], floatx4{0.f, 0.f, 0.f, 0.f});
            if (!real) v = floatx4{0.f, 0.f, 0.f, 0.f};
          } else {
            v = v * act_sx;
          }
          hi[h] = __builtin_convertvector(v, halfx4);
          lo[h] = __builtin_convertvector(v - __builtin_convertvector(hi[h], floatx4), halfx4);
        }
        *reinterpret_cast<halfx8*>(Pst + 128 * j * LDH) = __builtin_shufflevector(hi[0], hi[1], 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<halfx8*>(Pst + 128 * j * LDH + 32) = __builtin_shufflevector(lo[0], lo[1], 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  };
  auto load_b = [&](int set, int e) {
    const int e2 = e < nent ? e : nent - 1;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[set][i] = loadh8(wrsrc, wvoff[i], (unsigned)(e2 * 128));
  };
  auto store_b = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<halfx8*>(Bst + buf * BN * LDH + 64 * i * LDH) = rb[set][i];
  };

  load_patch(0);
  load_b(0, 0);
  load_b(1, 1);
  store_patch(0);
  store_b(0, 0);
  load_b(0, 2);
  __syncthreads();

  auto entry_step = [&](int e, auto par) {
    constexpr int Pb = decltype(par)::value;
    int ph, c, t, T;
    s2_entry(e, n, ph, c, t, T);
    int di, dj, kh, kw;  // tap t of the phase: row / column shift in the phase grid, filter tap
    s2_tap(ph, t, di, dj, kh, kw);
    const int d = di * Wo + dj, bit = kh * 3 + kw;
    const int u = ph * n + c;
    const bool first = t == 0, last = t == T - 1, more = u + 1 < 4 * n;
    const _Float16* Ab[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) Ab[i] = ((vmask[i] >> bit) & 1u) ? Afr[i] + d * LDH : Zfr;
    const _Float16* Bb = Bfr + Pb * BN * LDH;
    auto read_a = [&](halfx8 (&f)[MT], int qq) {
#pragma unroll
      for (int i = 0; i < MT; ++i) f[i] = *reinterpret_cast<const halfx8*>(Ab[i] + qq * 16);
    };
    auto read_b = [&](halfx8 (&f)[NT], int qq) {
#pragma unroll
      for (int i = 0; i < NT; ++i) f[i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH + qq * 16);
    };
    auto mm = [&](const halfx8 (&fa)[MT], const halfx8 (&fb)[NT]) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    };
    halfx8 ah[MT], al[MT], bh[NT], bl[NT], ah1[MT], bh1[NT];
    read_a(ah, 0);
    read_b(bh, 0);
    read_b(bl, 2);
    store_b(1 - Pb, 1 - Pb);
    load_b(1 - Pb, e + 3);
    mm(ah, bh);
    read_a(al, 2);
    mm(ah, bl);
    read_a(ah1, 1);
    read_b(bh1, 1);
    mm(al, bh);
    read_b(bl, 3);
    if (first && more) load_patch(u + 1);  // wave-uniform
    mm(ah1, bh1);
    read_a(al, 3);
    mm(ah1, bl);
    mm(al, bh1);
    __syncthreads();
    if (last && more) {
      store_patch(u + 1);
      __syncthreads();
    }
  };
  for (int e = 0; e < nent; e += 2) {  // 9 n entries, n even
    entry_step(e, std::integral_constant<int, 0>{});
    entry_step(e + 1, std::integral_constant<int, 1>{});
  }

  const float* const unscale = reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(a.w) + (size_t)a.Cout * 18 * C);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float s = unscale[n0 + wn + nt * 32 + (lane & 31)] * act_inv;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= s;
  }
  conv_epilogue<BM, BN, MT, NT, kThreads>(a, reinterpret_cast<float*>(lds_raw), acc, m0, n0, wm, wn);
}

template <int WAVES_M, int WAVES_N>
struct SplitTileS2 {
  static constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N;
  static int P(int Wo) { return BM + Wo + 1; }
  static int npc(int Wo) { return (P(Wo) + 127) / 128; }
  static size_t lds_bytes(int Wo) {
    const size_t loop = ((size_t)P(Wo) * LDH + 2 * (size_t)BN * LDH + LDH) * 2;
    const size_t epi = (size_t)BM * (BN + 4) * 4;
    return loop < epi ? epi : loop;
  }
};

template <int WAVES_M, int WAVES_N, bool PRE, int NPC>
int launch_split_s2_variant(ConvArgs args, hipStream_t stream) {
  using T = SplitTileS2<WAVES_M, WAVES_N>;
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& s) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_split_f32<WAVES_M, WAVES_N, PRE, NPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        s.spills = note_kernel(reinterpret_cast<const void*>(&conv3x3s2_split_f32<WAVES_M, WAVES_N, PRE, NPC>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + T::BM - 1) / T::BM);
  args.tiles_n = args.Cout / T::BN;
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int nblk = args.tiles_m * args.tiles_n;
  hipLaunchKernelGGL((conv3x3s2_split_f32<WAVES_M, WAVES_N, PRE, NPC>), dim3(8 * ((nblk + 7) / 8)), dim3(kThreads),
                     T::lds_bytes(args.Wo), stream, args, T::P(args.Wo));
  return check_launch("conv3x3s2_split_f32");
}

// tail split: T tiles on `slots` CUs (one workgroup each); the tiles of the last partial round are cut into S slices
struct SplitWs { float* slabs = nullptr; size_t slab_bytes = 0; int* counters = nullptr; size_t counter_bytes = 0; int slots = 0; };

int plan_tail_split(ConvArgs& a, int T, int ncc, size_t tile_floats, int wg_per_cu, hipStream_t stream) {
  // one workspace per (device, stream): launches on different streams (the lanes of a two-lane predictor) run
  // concurrently, and the default stream of two devices must not share slabs; the map is guarded because callers on
  // several host threads plan concurrently (ctypes releases the GIL)
  static std::map<std::pair<int, hipStream_t>, SplitWs> wss;
  static std::mutex wss_mutex;
  std::lock_guard<std::mutex> lock(wss_mutex);
  int dev = 0;
  HP_CHECK_HIP(hipGetDevice(&dev));
  SplitWs& ws = wss[std::make_pair(dev, stream)];
  if (ws.slots == 0) {
    int cus = 256;
    HP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    ws.slots = cus - cus % 8;
  }
  const bool no_split = dbg(DBG_CONV_NO_SPLITK) != 0;
  const int slots = ws.slots * wg_per_cu;
  int regular = (T / slots) * slots, S = 1;
  int tail = T - regular;
  // while two lanes share the GPU (tail split off) a launch is still sliced when it and its twin on the other lane
  // together cannot fill the GPU: it then plans against half of the CUs
  const bool shared = a.no_tail_split != 0;
  const int fill = shared ? slots / 2 : slots;
  if (tail > 0 && ncc > 1 && !no_split && (!shared || (regular == 0 && tail <= fill))) {
    // cost in units of a whole tile: rounds x longest slice + parking / re-reading the slabs (~1.3 us per 128 KB
    // against ~1 us per tap of the K loop)
    double best = 1.0;
    for (int s = 2; s <= ncc && s <= 4; ++s) {
      const double c = (double)((tail * s + fill - 1) / fill) * ((ncc + s - 1) / s) / ncc + 1.3 * (1 + s) / (9.0 * ncc);
      if (c < 0.92 * best) { best = c; S = s; }
    }
  }
  a.sk_regular = regular;
  a.sk_S = S;
  a.sk_tail_items = tail * S;
  a.sk_slabs = nullptr;
  a.sk_counters = nullptr;
  if (S > 1) {
    const size_t need_slab = (size_t)tail * S * tile_floats * sizeof(float), need_cnt = (size_t)tail * sizeof(int);
    if (ws.slab_bytes < need_slab || ws.counter_bytes < need_cnt) {  // must not grow under capture: the first (eager) call of a signature sizes it
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (stream) (void)hipStreamIsCapturing(stream, &cap);
      HP_REQUIRE(cap == hipStreamCaptureStatusNone, "conv tail split: the K-slice workspace of this stream would have to grow during stream capture");
    }
    if (ws.slab_bytes < need_slab) {
      if (ws.slabs) (void)hipFree(ws.slabs);
      ws.slabs = nullptr; ws.slab_bytes = 0;
      HP_CHECK_HIP(hipMalloc((void**)&ws.slabs, need_slab));
      ws.slab_bytes = need_slab;
    }
    if (ws.counter_bytes < need_cnt) {
      if (ws.counters) (void)hipFree(ws.counters);
      ws.counters = nullptr; ws.counter_bytes = 0;
      HP_CHECK_HIP(hipMalloc((void**)&ws.counters, need_cnt));
      HP_CHECK_HIP(hipMemsetAsync(ws.counters, 0, need_cnt, stream));  // kernels leave them at zero
      ws.counter_bytes = need_cnt;
    }
    a.sk_slabs = ws.slabs;
    a.sk_counters = ws.counters;
  }
  return HP_OK;
}

template <int WAVES_M, int WAVES_N, bool PRE, int NPC>
int launch_split_variant(ConvArgs args, hipStream_t stream) {
  using T = SplitTile<WAVES_M, WAVES_N>;
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& s) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_f32<WAVES_M, WAVES_N, PRE, NPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024 /* + the static ticket word */));
        s.spills = note_kernel(reinterpret_cast<const void*>(&conv3x3_split_f32<WAVES_M, WAVES_N, PRE, NPC>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + T::BM - 1) / T::BM);
  args.tiles_n = args.Cout / T::BN;
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int rc = plan_tail_split(args, args.tiles_m * args.tiles_n, args.Cin / CK, (size_t)T::BM * T::BN, 512 / T::NTHR, stream);
  if (rc) return rc;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8;
  hipLaunchKernelGGL((conv3x3_split_f32<WAVES_M, WAVES_N, PRE, NPC>), dim3(8 * per_xcd), dim3(T::NTHR),
                     T::lds_bytes(args.W), stream, args, T::P(args.W));
  return check_launch("conv3x3_split_f32");
}

template <int WAVES_M, int WAVES_N, bool PRE>
int launch_split_npc(const ConvArgs& a, hipStream_t stream) {
  const int npc = SplitTile<WAVES_M, WAVES_N>::npc(a.W);
  if (npc <= 3) return launch_split_variant<WAVES_M, WAVES_N, PRE, 3>(a, stream);
  if (npc == 4) return launch_split_variant<WAVES_M, WAVES_N, PRE, 4>(a, stream);
  if (npc <= 6) return launch_split_variant<WAVES_M, WAVES_N, PRE, 6>(a, stream);
  return launch_split_variant<WAVES_M, WAVES_N, PRE, 7>(a, stream);
}

}  // namespace

int conv_split_plan_tail(ConvArgs& a, int T, int ncc, size_t tile_floats, int wg_per_cu, hipStream_t stream) {
  return plan_tail_split(a, T, ncc, tile_floats, wg_per_cu, stream);
}

bool conv_split_applicable(const ConvArgs& a, int kh, int kw) {
  if (kh != 3 || kw != 3 || a.pad != 1 || a.Cin % CK != 0 || a.Cout % 64 != 0) return false;
  if (a.stride == 2)  // space-to-depth walk: an even number of 32-channel chunks, 256 x 128 tiles, <= 3 staging passes
    return a.Cin % (2 * CK) == 0 && a.Cout % 128 == 0 && SplitTileS2<4, 2>::npc(a.Wo) <= 3 &&
           a.Ho == (a.H - 1) / 2 + 1 && a.Wo == (a.W - 1) / 2 + 1;
  if (a.stride != 1) return false;
  if (a.Cout % 128 == 0) return SplitTile<4, 2>::npc(a.W) <= 6 && SplitTile<4, 2>::lds_bytes(a.W) <= 159 * 1024;
  return SplitTile<8, 1>::npc(a.W) <= 6 && SplitTile<8, 1>::lds_bytes(a.W) <= 159 * 1024;
}

bool conv_split_launchable(const ConvArgs& a) {
  // 32-bit buffer offsets; the activation may not be a pre-scale-only (squeeze-excitation) input
  return a.M * a.Cin * 4 * a.stride * a.stride < (1ll << 31) && a.M < (1ll << 31) && (a.pre_shift || !a.pre_scale) && a.relu != HP_ACT_SWISH;
}

size_t conv_split_weight_bytes(int cout, int cin) { return (size_t)cout * 18 * cin * 2 + (size_t)cout * 4; }

int conv_split_transform_weights(const float* d_w, void* d_ws, int cout, int cin, int Kpad, int stride, hipStream_t stream) {
  _Float16* ws = reinterpret_cast<_Float16*>(d_ws);
  float* unscale = reinterpret_cast<float*>(ws + (size_t)cout * 18 * cin);
  if (stride == 2) hipLaunchKernelGGL(split_weights_s2_kernel, dim3(cout), dim3(256), 0, stream, d_w, ws, unscale, cin, Kpad);
  else hipLaunchKernelGGL(split_weights_kernel, dim3(cout), dim3(256), 0, stream, d_w, ws, unscale, cin, Kpad);
  return check_launch("split_weights_kernel");
}

int launch_conv_split(const ConvArgs& a, hipStream_t stream) {
  const bool pre = a.pre_scale != nullptr;
  if (a.stride == 2) {
    if (conv_pp_s2_applicable(a, 3, 3)) return launch_conv_pp_s2_split(a, stream);  // ping-pong skeleton (conv_pp.hip)
    return pre ? launch_split_s2_variant<4, 2, true, 3>(a, stream) : launch_split_s2_variant<4, 2, false, 3>(a, stream);
  }
  // >= 128 output channels: the ping-pong kernel (conv_pp.hip) wherever its double-buffered patch fits the LDS
  if (conv_pp_split_applicable(a, 3, 3)) return launch_conv_pp_split(a, stream);
  // (128 x 128 tiles in 4-wave workgroups, two per CU, measured 3-10 % slower than 256 x 128 on the >= 128-channel layers)
  if (a.Cout % 128 == 0) return pre ? launch_split_npc<4, 2, true>(a, stream) : launch_split_npc<4, 2, false>(a, stream);
  // 64-channel layers (60x80 maps, K = 18 taps): 256 x 64 tiles in 4-wave workgroups, TWO per CU -- with one 512 x 64
  // workgroup per CU nothing runs under its prologue, patch restaging and epilogue, a third of such a short tile
  if (SplitTile<4, 1>::npc(a.W) <= 7 && 2 * SplitTile<4, 1>::lds_bytes(a.W) + 1024 <= 160 * 1024)
    return pre ? launch_split_npc<4, 1, true>(a, stream) : launch_split_npc<4, 1, false>(a, stream);
  return pre ? launch_split_npc<8, 1, true>(a, stream) : launch_split_npc<8, 1, false>(a, stream);
}

}  // namespace hp
