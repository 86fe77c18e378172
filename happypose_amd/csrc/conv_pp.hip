// 3x3 / stride-1 / pad-1 convolution, Cout % 128 == 0, as a PING-PONG kernel: the two waves that share a SIMD never
// compete for the matrix pipe.
//
// Where conv3x3_split_f32 (conv_split.hip) lost its time: both waves of a SIMD ran the same tap at the same moment, each
// interleaving 16 fragment reads, staging stores and 24 MFMAs, all eight waves meeting at a barrier per tap.  The SQ
// counters of that kernel (profiles/r01h_split_pmc_summary.txt) show the matrix pipe busy 65 % of the time waves are
// resident; pulling the fragment reads a group ahead of their MFMAs (software pipelining, still in conv_split.hip) moved
// that by 1-4 %, i.e. the loss is not an exposed LDS round trip but the arbitration between two waves that BOTH want
// the pipe and BOTH stall on lgkmcnt in the middle of their MFMA stream (an in-order wave cannot slip an MFMA into idle
// pipe time unless the idle slot is at that point of ITS program order: MI355X_MICROARCH.md, "Two waves per SIMD").
//
// Here a tap is two segments per wave, separated by workgroup barriers:
//   L  stage the weights of tap t+1 (registers -> LDS), issue the global loads of tap t+3 / of the next channel chunk,
//      read all 16 operand fragments of tap t from LDS into registers, wait for them;
//   C  24 (split-fp16) or 16 (fp16) MFMAs back to back, operands in registers, nothing else in the instruction stream.
// Waves 0-3 (one per SIMD) run L in even phases and C in odd phases, waves 4-7 the other way round -- the same code,
// shifted by ONE extra s_barrier at the head of waves 4-7 (and one at the tail of waves 0-3).  A SIMD therefore always
// has exactly one wave in its MFMA burst while the other one does every LDS / VMEM / VALU instruction of the tap.
// LDS: the input patch of a 32- (fp32) or 64-channel (fp16) chunk is double buffered, so the chunk swap needs no phase
// of its own: [2][P][72] + weights [2][128][72] halves (+ a zero row), 134 KB at W = 40.
//
// Two arithmetic modes share the skeleton:
//   MODE_SPLIT  fp32 activations / weights as fp16 hi + lo halves, three MFMAs per product, fp32 accumulate, fp32 out
//               (conv_split.hip's scheme, same weight layout: conv_split_transform_weights);
//   MODE_F16    fp16 activations / weights (the fp16 plan of configuration C5), one MFMA per product, fp32 accumulate,
//               fp16 out; weights in the plan's packed layout [Cout][(tap, c)].
// Replaces the same reference ops as conv.hip: ATen conv2d + BatchNorm + ReLU (+ identity) of
// MP/models/torchvision_resnet.py:110-126 and MP/models/wide_resnet.py:59-65.
#include <cstdlib>
#include <type_traits>

#include "conv.h"
#include "conv_epilogue.h"
#include "conv_splitk.h"

namespace hp {

typedef _Float16 pp_halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 pp_halfx4 __attribute__((ext_vector_type(4)));
typedef float pp_floatx16 __attribute__((ext_vector_type(16)));
typedef float pp_floatx4 __attribute__((ext_vector_type(4)));

#ifdef HP_PP_STAMPS  // diagnostics build: shader cycles and 100-MHz ticks spent in the K loops (-> the shader clock)
__device__ unsigned long long g_pp_stamps[8];
#endif

namespace {

enum { MODE_SPLIT = 0, MODE_F16 = 1 };
constexpr int kPPThreads = 512;
constexpr int LDH = 64 + 8;  // LDS row: 64 halves (32 hi + 32 lo, or 64 channels), padded to 36 dwords
constexpr unsigned kOob = 0xFFFFFFF0u;
constexpr int BM = 256, BN = 128, MT = 2, NT = 2;

__device__ __forceinline__ pp_floatx4 ldf4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(pp_floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ pp_halfx8 ldh8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(pp_halfx8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int MODE>
struct PP {
  static constexpr int CKC = MODE == MODE_SPLIT ? 32 : 64;  // channels per chunk (= one 128-B LDS row)
  static constexpr int TPR = MODE == MODE_SPLIT ? 4 : 8;    // threads per patch row (8 channels each)
  static constexpr int PROWS = kPPThreads / TPR;            // patch rows per staging pass
  static constexpr int ESZ = MODE == MODE_SPLIT ? 4 : 2;    // bytes per activation element
  static int P(int W) { return BM + 2 * W + 2; }
  static int npc(int W) { return (P(W) + PROWS - 1) / PROWS; }
  static constexpr int kPreFloats = 2 * 512;  // pre-activation BN scale / shift of up to 512 input channels, staged once
  static size_t lds_bytes(int W) {
    const size_t loop = ((size_t)2 * P(W) * LDH + 2 * (size_t)BN * LDH + LDH) * 2 + kPreFloats * 4;
    const size_t epi = (size_t)BM * (BN + 4) * 4;
    return loop < epi ? epi : loop;
  }
};

// fp16 epilogue: accumulators -> LDS [row][BN + 4] floats -> bias (fp32), residual (fp16), ReLU -> 8 halves per store
__device__ __forceinline__ void pp_epilogue_f16(const ConvArgs& a, float* cl, pp_floatx16 (&acc)[MT][NT], int64_t m0, int n0,
                                                int wm, int wn) {
  constexpr int LDC = BN + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        cl[row * LDC + wn + nt * 32 + (lane & 31)] = acc[mt][nt][r];
      }
  __syncthreads();
  constexpr int C8 = BN / 8;
  constexpr int ITERS = BM * C8 / kPPThreads;
  const int c8 = tid % C8;
  const int n = n0 + 8 * c8;
  const _Float16* const res = reinterpret_cast<const _Float16*>(a.residual);
  _Float16* const y = reinterpret_cast<_Float16*>(a.y);
  pp_floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (a.bias) {
    b0 = *reinterpret_cast<const pp_floatx4*>(a.bias + n);
    b1 = *reinterpret_cast<const pp_floatx4*>(a.bias + n + 4);
  }
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const int row = tid / C8 + k * (kPPThreads / C8);
    const int64_t m = m0 + row;
    if (m < a.M) {
      pp_floatx4 v0 = *reinterpret_cast<const pp_floatx4*>(cl + row * LDC + 8 * c8) + b0;
      pp_floatx4 v1 = *reinterpret_cast<const pp_floatx4*>(cl + row * LDC + 8 * c8 + 4) + b1;
      if (res) {
        const pp_halfx8 rr = *reinterpret_cast<const pp_halfx8*>(res + m * a.Cout + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] += (float)rr[q]; v1[q] += (float)rr[4 + q]; }
      }
      if (a.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
      }
      pp_halfx8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q] = (_Float16)v0[q]; o[4 + q] = (_Float16)v1[q]; }
      *reinterpret_cast<pp_halfx8*>(y + m * a.Cout + n) = o;
    }
  }
}

// a.Kpad: MODE_F16 = elements per weight row ([Cout][Kpad] halves, K = (tap, c)); MODE_SPLIT unused (rows are 18 Cin halves)
template <int MODE, bool PRE, int NPC>
__global__ __launch_bounds__(kPPThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_pp(ConvArgs a, int P) {
  using T = PP<MODE>;
  constexpr int CKC = T::CKC, TPR = T::TPR, PROWS = T::PROWS, ESZ = T::ESZ;
  constexpr int NB = BN * 8 / kPPThreads;  // 16-B weight pieces per thread and tap (2)
  constexpr int BROWS = kPPThreads / 8;    // weight rows per staging pass (64)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const patch = reinterpret_cast<_Float16*>(lds_raw);  // [2][P][LDH]
  _Float16* const Bs = patch + 2 * P * LDH;                        // [2][BN][LDH]
  _Float16* const zrow = Bs + 2 * BN * LDH;                        // [LDH] zeros: what a masked tap reads
  float act_sx = 1.f, act_inv = 1.f;  // ConvArgs::amax_in: power-of-two scale of the staged activations (split mode)
  if constexpr (MODE == MODE_SPLIT) conv_act_scale(a, act_sx, act_inv);
  // pre-activation BN + ReLU vectors in LDS (fetched from global memory inside the L segment of a chunk's last tap they
  // stalled the whole phase for an HBM / L2 round trip: 1918 instead of 1690 cycles per tap on the PRE layers)
  float* const pre_lds = reinterpret_cast<float*>(zrow + LDH);      // split: [Cin] scale, [Cin] shift (fp32); f16: halves

#ifdef HP_PP_STAMPS
  const unsigned long long st_k0 = __builtin_readcyclecounter();
#endif
  int lin, slice;
  bool split;
  if (!splitk_decode(a, lin, slice, split)) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H, Cin = a.Cin;
  const int ncc_all = Cin / CKC, ntaps = ncc_all * 9;
  const int cc_begin = split ? slice * ncc_all / a.sk_S : 0;
  const int ncc = split ? (slice + 1) * ncc_all / a.sk_S : ncc_all;  // end of this item's chunk range

  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(a.M * Cin * ESZ), 0x00020000);
  const int64_t wrow = MODE == MODE_SPLIT ? (int64_t)18 * Cin : (int64_t)a.Kpad;  // halves per cout
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((int64_t)a.Cout * wrow * 2), 0x00020000);

  // patch staging: row = pr0 + PROWS j, channels 8 pk .. 8 pk + 7 of the chunk
  const int pk = tid % TPR, pr0 = tid / TPR;
  const int64_t gp0 = m0 - (W + 1) + pr0;
  auto patch_voff = [&](int j) -> unsigned {
    const int64_t gp = gp0 + PROWS * j;
    return (pr0 + PROWS * j < P && gp >= 0 && gp < a.M) ? (unsigned)((gp * Cin + 8 * pk) * ESZ) : kOob;
  };
  // split: this thread's 8 channels land at halves [8 pk, 8 pk + 8) (hi) and 32 + [8 pk, ...) (lo); f16: at [8 pk, ...)
  _Float16* const Pst = patch + pr0 * LDH + 8 * pk;
  // weight staging: row = br0 + 64 i, 16-B piece bk of the 128-B row
  const int bk = tid & 7, br0 = tid >> 3;
  unsigned wvoff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(n0 + br0 + BROWS * i) * wrow + 8 * bk) * 2);
  _Float16* const Bst = Bs + br0 * LDH + 8 * bk;
  if (tid < LDH / 2) reinterpret_cast<unsigned*>(zrow)[tid] = 0u;
  if (PRE) {
    if constexpr (MODE == MODE_SPLIT) {
      for (int i = tid; i < Cin; i += kPPThreads) { pre_lds[i] = a.pre_scale[i] * act_sx; pre_lds[Cin + i] = a.pre_shift[i] * act_sx; }  // sx relu(x s + b) = relu(x s sx + b sx)
    } else {
      _Float16* const ph = reinterpret_cast<_Float16*>(pre_lds);
      for (int i = tid; i < Cin; i += kPPThreads) {
        ph[i] = reinterpret_cast<const _Float16*>(a.pre_scale)[i];
        ph[Cin + i] = reinterpret_cast<const _Float16*>(a.pre_shift)[i];
      }
    }
    __syncthreads();  // the prologue's store_patch reads them
  }
  auto w_soff = [&](int cc, int tap) -> unsigned {  // byte offset of the (chunk, tap) row piece inside a cout's weights
    return MODE == MODE_SPLIT ? (unsigned)((cc * 9 + tap) * 128) : (unsigned)((tap * Cin + cc * CKC) * 2);
  };

  // fragment bases + validity of the 9 taps per fragment row
  const int wm = ((wave & 3) >> 1) * 64 + (wave >> 2) * 128, wn = (wave & 1) * 64;  // waves w and w + 4 share a SIMD
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

  pp_floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging registers: the next chunk's patch rows of this thread, the weights of taps t+1 / t+2 (alternating sets)
  pp_floatx4 prf[MODE == MODE_SPLIT ? NPC : 1][2];
  pp_halfx8 prh[MODE == MODE_F16 ? NPC : 1];
  pp_halfx8 rb[2][NB];
  auto load_patch = [&](int j, int cc) {
    const unsigned vo = patch_voff(j);
    if constexpr (MODE == MODE_SPLIT) {
      prf[j][0] = ldf4(xrsrc, vo, (unsigned)(cc * CKC * 4));
      prf[j][1] = ldf4(xrsrc, vo, (unsigned)(cc * CKC * 4 + 16));
    } else {
      prh[j] = ldh8(xrsrc, vo, (unsigned)(cc * CKC * 2));
    }
  };
  auto store_patch = [&](int cc, int pbuf) {
    _Float16* const dst = Pst + pbuf * P * LDH;
    if constexpr (MODE == MODE_SPLIT) {
      pp_floatx4 ps[2], pb[2];
      if (PRE) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          ps[h] = *reinterpret_cast<const pp_floatx4*>(pre_lds + cc * CKC + 8 * pk + 4 * h);
          pb[h] = *reinterpret_cast<const pp_floatx4*>(pre_lds + Cin + cc * CKC + 8 * pk + 4 * h);
        }
      }
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        if (pr0 + PROWS * j < P) {
          const bool real = patch_voff(j) != kOob;  // pixels outside the tensor stay zero
          pp_halfx4 hi[2], lo[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            pp_floatx4 v = prf[j][h];
            if (PRE) {
              v = __builtin_elementwise_max(v * ps[h] + pb[h], pp_floatx4{0.f, 0.f, 0.f, 0.f});
              if (!real) v = pp_floatx4{0.f, 0.f, 0.f, 0.f};
            } else {
              v = v * act_sx;
            }
            hi[h] = __builtin_convertvector(v, pp_halfx4);
            lo[h] = __builtin_convertvector(v - __builtin_convertvector(hi[h], pp_floatx4), pp_halfx4);
          }
          *reinterpret_cast<pp_halfx8*>(dst + PROWS * j * LDH) = __builtin_shufflevector(hi[0], hi[1], 0, 1, 2, 3, 4, 5, 6, 7);
          *reinterpret_cast<pp_halfx8*>(dst + PROWS * j * LDH + 32) = __builtin_shufflevector(lo[0], lo[1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    } else {
      pp_halfx8 ps, pb;
      if (PRE) {
        ps = *reinterpret_cast<const pp_halfx8*>(reinterpret_cast<const _Float16*>(pre_lds) + cc * CKC + 8 * pk);
        pb = *reinterpret_cast<const pp_halfx8*>(reinterpret_cast<const _Float16*>(pre_lds) + Cin + cc * CKC + 8 * pk);
      }
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        if (pr0 + PROWS * j < P) {
          pp_halfx8 v = prh[j];
          if (PRE) {
            const pp_halfx8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            v = __builtin_elementwise_max(v * ps + pb, zero);
            if (patch_voff(j) == kOob) v = zero;
          }
          *reinterpret_cast<pp_halfx8*>(dst + PROWS * j * LDH) = v;
        }
      }
    }
  };
  auto load_b = [&](int set, int cc, int tap) {  // clamped to the layer's last tap: the tail of an item re-reads it
    if (cc >= ncc_all) { cc = ncc_all - 1; tap = 8; }
    const unsigned so = w_soff(cc, tap);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[set][i] = ldh8(wrsrc, wvoff[i], so);
  };
  auto store_b = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<pp_halfx8*>(Bst + buf * BN * LDH + BROWS * i * LDH) = rb[set][i];
  };
  auto tap_at = [&](int cc, int tap, int d, int& c2, int& t2) {  // (cc, tap) + d taps
    t2 = tap + d; c2 = cc;
    if (t2 >= 9) { t2 -= 9; ++c2; }
  };

  // prologue: patch of the first chunk -> buffer 0, weights of tap 0 -> LDS buffer 0, taps 1 and 2 -> registers
#pragma unroll
  for (int j = 0; j < NPC; ++j) load_patch(j, cc_begin);
  load_b(0, cc_begin, 0);
  load_b(1, cc_begin, 1);
  store_patch(cc_begin, 0);
  store_b(0, 0);
  load_b(0, cc_begin, 2);
  __syncthreads();

  const bool odd = wave >= 4;  // waves 4-7 run one phase behind waves 0-3
#ifdef HP_PP_STAMPS
  const unsigned long long st_t0 = __builtin_readcyclecounter(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (odd) __builtin_amdgcn_s_barrier();

  // one tap: L segment, barrier, C segment, barrier.  Pb = LDS weight buffer (tap parity within the item), PQ = patch
  // buffer (chunk parity within the item) -- compile-time through the two-chunk unrolling below.
  auto tap_step = [&](int cc, int tap, auto par, auto ppar) {
    constexpr int Pb = decltype(par)::value;
    constexpr int PQ = decltype(ppar)::value;
    const bool next_chunk = cc + 1 < ncc;
    // ---- L: staging + every operand fragment of this tap
    store_b(1 - Pb, 1 - Pb);  // weights of the next tap (register set (t + 1) & 1) ...
    {
      int c3, t3;
      tap_at(cc, tap, 3, c3, t3);
      load_b(1 - Pb, c3, t3);   // ... and that set takes the tap three ahead
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j)  // next chunk's patch: one staging pass per tap (taps 0 .. NPC - 1)
      if (j == tap) load_patch(j, next_chunk ? cc + 1 : cc);
    if (tap == 8 && next_chunk) store_patch(cc + 1, 1 - PQ);
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
    pp_halfx8 fa[4][MT], fb[4][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const _Float16* const Ab = ((vmask[i] >> tap) & 1u) ? Afr[i] + PQ * P * LDH + d * LDH : Zfr;
#pragma unroll
      for (int q = 0; q < 4; ++q) fa[q][i] = *reinterpret_cast<const pp_halfx8*>(Ab + q * 16);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) fb[q][i] = *reinterpret_cast<const pp_halfx8*>(Bfr + Pb * BN * LDH + i * 32 * LDH + q * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers, staging stores in LDS
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- C: the MFMA burst
    auto mm = [&](const pp_halfx8 (&x)[MT], const pp_halfx8 (&y)[NT]) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[mi], y[ni], acc[mi][ni], 0, 0, 0);
    };
    if constexpr (MODE == MODE_SPLIT) {
      // fragment q of a row: 0 / 1 = hi halves of channels 0-15 / 16-31, 2 / 3 = their lo halves
      mm(fa[0], fb[0]); mm(fa[0], fb[2]); mm(fa[2], fb[0]);
      mm(fa[1], fb[1]); mm(fa[1], fb[3]); mm(fa[3], fb[1]);
    } else {
      mm(fa[0], fb[0]); mm(fa[1], fb[1]); mm(fa[2], fb[2]); mm(fa[3], fb[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // chunk number rc of the item: patch buffer rc & 1; its tap t uses weight buffer (rc + t) & 1 (9 taps per chunk)
  auto chunk = [&](int cc, auto c0) {
    constexpr int C0 = decltype(c0)::value;
    using E = std::integral_constant<int, C0>;      // even taps
    using O = std::integral_constant<int, 1 - C0>;  // odd taps
    using Q = std::integral_constant<int, C0>;      // patch buffer
    tap_step(cc, 0, E{}, Q{}); tap_step(cc, 1, O{}, Q{}); tap_step(cc, 2, E{}, Q{});
    tap_step(cc, 3, O{}, Q{}); tap_step(cc, 4, E{}, Q{}); tap_step(cc, 5, O{}, Q{});
    tap_step(cc, 6, E{}, Q{}); tap_step(cc, 7, O{}, Q{}); tap_step(cc, 8, E{}, Q{});
  };
  int cc = cc_begin;
  for (; cc + 1 < ncc; cc += 2) {
    chunk(cc, std::integral_constant<int, 0>{});
    chunk(cc + 1, std::integral_constant<int, 1>{});
  }
  if (cc < ncc) chunk(cc, std::integral_constant<int, 0>{});
  if (!odd) __builtin_amdgcn_s_barrier();
  (void)ntaps;
#ifdef HP_PP_STAMPS
  if (tid == 0) {
    atomicAdd(&g_pp_stamps[0], __builtin_readcyclecounter() - st_t0);
    atomicAdd(&g_pp_stamps[1], __builtin_amdgcn_s_memrealtime() - st_r0);
    atomicAdd(&g_pp_stamps[2], (unsigned long long)((ncc - cc_begin) * 9));
    atomicAdd(&g_pp_stamps[3], 1ull);
    atomicAdd(&g_pp_stamps[4], st_t0 - st_k0);  // prologue
  }
  const unsigned long long st_e0 = __builtin_readcyclecounter();
#endif

  if (split && !splitk_reduce_sc1<BM, BN, MT, NT, kPPThreads>(a, acc, lin - a.sk_regular, slice)) return;

  if constexpr (MODE == MODE_SPLIT) {
    // scale back (a lane holds one output channel per N tile), then the shared fp32 epilogue
    const float* const unscale = reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(a.w) + (size_t)a.Cout * 18 * Cin);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float s = unscale[n0 + wn + nt * 32 + (lane & 31)] * act_inv;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= s;
    }
    conv_epilogue<BM, BN, MT, NT, kPPThreads>(a, reinterpret_cast<float*>(lds_raw), acc, m0, n0, wm, wn);
  } else {
    pp_epilogue_f16(a, reinterpret_cast<float*>(lds_raw), acc, m0, n0, wm, wn);
  }
#ifdef HP_PP_STAMPS
  __syncthreads();
  if (tid == 0) atomicAdd(&g_pp_stamps[5], __builtin_readcyclecounter() - st_e0);  // slab hand-off + epilogue (full tiles / last arrivers)
#endif
}

template <int MODE, bool PRE, int NPC>
int launch_pp_variant(ConvArgs args, hipStream_t stream) {
  using T = PP<MODE>;
  static bool opted = false, spills = false;
  if (!opted) {
    HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp<MODE, PRE, NPC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));  // + the static ticket word
    spills = note_kernel(reinterpret_cast<const void*>(&conv3x3_pp<MODE, PRE, NPC>));
    opted = true;
  }
  if (spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = args.Cout / BN;
  args.fd_howo = make_fastdiv((unsigned)(args.H * args.W));
  args.fd_wo = make_fastdiv((unsigned)args.W);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int rc = conv_split_plan_tail(args, args.tiles_m * args.tiles_n, args.Cin / T::CKC, (size_t)BM * BN, 1, stream);
  if (rc) return rc;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8;
  hipLaunchKernelGGL((conv3x3_pp<MODE, PRE, NPC>), dim3(8 * per_xcd), dim3(kPPThreads), T::lds_bytes(args.W), stream, args,
                     T::P(args.W));
  return check_launch("conv3x3_pp");
}

template <int MODE, bool PRE>
int launch_pp_npc(const ConvArgs& a, hipStream_t stream) {
  const int npc = PP<MODE>::npc(a.W);
  if (npc <= 3) return launch_pp_variant<MODE, PRE, 3>(a, stream);
  if (npc == 4) return launch_pp_variant<MODE, PRE, 4>(a, stream);
  if (npc <= 6) return launch_pp_variant<MODE, PRE, 6>(a, stream);
  return fail(HP_ERR_ARG, "conv3x3_pp: map too wide for the staged patch");
}

template <int MODE>
bool pp_shape_ok(int W, int Cin, int Cout, int stride, int pad, int kh, int kw) {
  static const bool off = std::getenv("HP_CONV_NO_PP") != nullptr;
  return !off && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Cout % BN == 0 && Cin % PP<MODE>::CKC == 0 && Cin <= 512 &&
         PP<MODE>::npc(W) <= 6 && PP<MODE>::lds_bytes(W) <= 159 * 1024;
}

}  // namespace

// fp32 (split-fp16) entry: a.w = weights split by conv_split_transform_weights
bool conv_pp_split_applicable(const ConvArgs& a, int kh, int kw) {
  return pp_shape_ok<MODE_SPLIT>(a.W, a.Cin, a.Cout, a.stride, a.pad, kh, kw);
}

int launch_conv_pp_split(const ConvArgs& a, hipStream_t stream) {
  return a.pre_scale ? launch_pp_npc<MODE_SPLIT, true>(a, stream) : launch_pp_npc<MODE_SPLIT, false>(a, stream);
}

// fp16 entry (the fp16 plan): packed weights [Cout][Kpad] with K = (tap, c)
bool conv_pp_f16_applicable(const ConvArgsH& a) {
  return pp_shape_ok<MODE_F16>(a.W, a.Cin, a.Cout, a.stride, a.pad, a.kh, a.kw) && a.Ho == a.H && a.Wo == a.W;
}

int launch_conv_pp_f16(const ConvArgsH& h, hipStream_t stream) {
  ConvArgs a{};
  a.x = reinterpret_cast<const float*>(h.x); a.w = reinterpret_cast<const float*>(h.w); a.bias = h.bias;
  a.residual = reinterpret_cast<const float*>(h.residual);
  a.pre_scale = reinterpret_cast<const float*>(h.pre_scale); a.pre_shift = reinterpret_cast<const float*>(h.pre_shift);
  a.y = reinterpret_cast<float*>(h.y);
  a.M = h.M; a.H = h.H; a.W = h.W; a.Cin = h.Cin; a.Ho = h.Ho; a.Wo = h.Wo; a.Cout = h.Cout; a.stride = 1; a.pad = 1;
  a.Kpad = h.Kpad; a.relu = h.relu; a.no_tail_split = h.no_tail_split;
  return h.pre_scale ? launch_pp_npc<MODE_F16, true>(a, stream) : launch_pp_npc<MODE_F16, false>(a, stream);
}

}  // namespace hp

#ifdef HP_PP_STAMPS
extern "C" int hp_debug_pp_stamps(double* out4) {  // cycles, 100-MHz ticks, taps, workgroups, prologue cycles, epilogue cycles
  unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(hp::g_pp_stamps), sizeof(h)) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hp::g_pp_stamps), z, sizeof(z));
  for (int i = 0; i < 6; ++i) out4[i] = (double)h[i];
  return 0;
}
#endif
