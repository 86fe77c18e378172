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
#include <algorithm>
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

int conv_num_cus() {  // of the current device (queried once: one device per process, as everywhere in this library)
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 8 ? n : 256;
  }();
  return cus;
}

namespace {

enum { MODE_SPLIT = 0, MODE_F16 = 1 };
constexpr int kPPThreads = 512;
constexpr int LDH = 64 + 8;  // LDS row: 64 halves (32 hi + 32 lo, or 64 channels), padded to 36 dwords
constexpr unsigned kOob = 0xFFFFFFF0u;
constexpr int BM = 256, MT = 2;  // block tile 256 x (64 NT): NT = 2 (Cout % 128 == 0) or 1 (the 64-channel layers)

__device__ __forceinline__ pp_floatx4 ldf4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(pp_floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ pp_halfx8 ldh8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(pp_halfx8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// WIDE (round 4, fp32 / split mode only): 512 x 64 block tile for the 64-channel layers -- eight waves of 64 x 64, stacked in
// M (waves 0-3 rows 0-255, their SIMD partners 4-7 rows 256-511), ONE patch buffer (two would not fit) swapped between
// chunks in a two-phase bubble.
template <int MODE, int NT, bool WIDE = false>
struct PP {
  static constexpr int BN = WIDE ? 64 : 64 * NT;
  static constexpr int BMT = WIDE ? 512 : BM;   // rows of the block tile
  static constexpr int NPB = WIDE ? 1 : 2;      // patch buffers
  static constexpr int CKC = MODE == MODE_SPLIT ? 32 : 64;  // channels per chunk (= one 128-B LDS row)
  static constexpr int TPR = MODE == MODE_SPLIT ? 4 : 8;    // threads per patch row (8 channels each)
  static constexpr int PROWS = kPPThreads / TPR;            // patch rows per staging pass
  static constexpr int ESZ = MODE == MODE_SPLIT ? 4 : 2;    // bytes per activation element
  static int P(int W) { return BMT + 2 * W + 2; }
  static int npc(int W) { return (P(W) + PROWS - 1) / PROWS; }
  static constexpr int kPreFloats = 2 * 512;  // pre-activation BN scale / shift of up to 512 input channels, staged once
  static size_t lds_bytes(int W) {
    return ((size_t)NPB * P(W) * LDH + 2 * (size_t)BN * LDH + LDH) * 2 + kPreFloats * 4;  // the epilogue needs none (direct stores)
  }
};

// Direct epilogue (round 4).  The MFMAs run TRANSPOSED -- weights as the A operand, pixels as B -- so a lane holds ONE pixel
// (column l & 31) and 16 output channels per 32 x 32 tile; the weight rows are read in the order sigma(i) = 16 ((i >> 2) & 1) +
// 4 (i >> 3) + (i & 3), which makes those 16 channels CONSECUTIVE: acc[mt][nt][r] = (pixel m0 + wm + 32 mt + (l & 31), channel
// n0 + wn + 32 nt + 16 (l >> 5) + r).  Bias, unscale, residual, activation and the NHWC store are 16-B accesses straight from
// the accumulator registers: no LDS transpose, no workgroup barrier -- each wave leaves on its own, its stores drain while
// the next tile's prologue runs (the old epilogue: 64 ds_write_b32 + 16 ds_read_b128 per lane between two barriers, ~18 k
// cycles per 256 x 128 tile, a quarter of a 36-tap tile).  The residual pieces of the whole wave tile are fetched first, into
// the registers the operand fragments just vacated.
// 4 x 4 transpose inside a lane quad: lane j's (r0, r1, r2, r3) become element j of lanes 0 .. 3 (two DPP butterfly stages)
__device__ __forceinline__ void quad_transpose4(int qj, float& r0, float& r1, float& r2, float& r3) {
  {  // lanes differing in bit 0 exchange across the register pairs (0, 1) and (2, 3)
    const bool b = qj & 1;
    const float s01 = b ? r0 : r1, s23 = b ? r2 : r3;
    const float g01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
    const float g23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
    if (b) { r0 = g01; r2 = g23; } else { r1 = g01; r3 = g23; }
  }
  {  // lanes differing in bit 1: pairs (0, 2) and (1, 3)
    const bool b = qj & 2;
    const float s02 = b ? r0 : r2, s13 = b ? r1 : r3;
    const float g02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
    const float g13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
    if (b) { r0 = g02; r1 = g13; } else { r2 = g02; r3 = g13; }
  }
}

// `between()` runs once, after the first of the wave's two M tiles has been stored (its 32 accumulator registers are dead by
// then): the persistent loop issues the NEXT item's first loads there, so that they land under the rest of this epilogue.
template <int MODE, int NT, int MTT = MT, typename Hook>
__device__ __forceinline__ void pp_epilogue_direct(const ConvArgs& a, pp_floatx16 (&acc)[MTT][NT], int64_t m0, int n0, int wm, int wn,
                                                   float act_inv, int lane, Hook&& between) {
  const int px = lane & 31, h16 = 16 * (lane >> 5);
  float chk = 0.f, amax = 0.f;
  const float* const unscale = MODE == MODE_SPLIT
      ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(a.w) + (size_t)a.Cout * 18 * a.Cin) : nullptr;
  if constexpr (MODE == MODE_SPLIT) {
    // A lane holds 64 contiguous bytes of ONE pixel per 32 x 32 tile; stored as they lie, the 32 lanes of a half-wave would
    // touch 32 different 512-B rows per instruction (measured: 8-12 % slower than the LDS-transposed epilogue).  A 4 x 4
    // transpose inside each lane quad (two DPP butterfly stages) gives lane j of a quad piece j (couts 4j .. 4j + 3) of the
    // quad's four pixels instead: store k then writes pixel k of every quad, 4 lanes x 16 B = 64 contiguous bytes, and the
    // two half-waves complete the 128-B line -- 8 whole lines per instruction.
    const int qj = lane & 3, qp = px & ~3;
    auto quad_transpose = [&](pp_floatx4 (&t)[4]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float r0 = t[0][c], r1 = t[1][c], r2 = t[2][c], r3 = t[3][c];
        quad_transpose4(qj, r0, r1, r2, r3);
        t[0][c] = r0; t[1][c] = r1; t[2][c] = r2; t[3][c] = r3;
      }
    };
    // after the transpose: t[k] = couts [n + 4 qj, + 4) of pixel row m0 + wm + 32 mt + qp + k
    pp_floatx4 sc[NT], bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn + nt * 32 + h16 + 4 * qj;
      sc[nt] = *reinterpret_cast<const pp_floatx4*>(unscale + n) * act_inv;
      bias[nt] = pp_floatx4{0.f, 0.f, 0.f, 0.f};
      if (a.bias) bias[nt] = *reinterpret_cast<const pp_floatx4*>(a.bias + n);
    }
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt) {
      pp_floatx4 res[NT][4];
      if (a.residual) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + wm + mt * 32 + qp + k;
          const float* const rp = a.residual + (m < a.M ? m : 0) * a.Cout + n0 + wn + h16 + 4 * qj;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) res[nt][k] = *reinterpret_cast<const pp_floatx4*>(rp + nt * 32);
        }
      }
      if (mt == 1) between();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn + nt * 32 + h16 + 4 * qj;
        pp_floatx4 t[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) t[g] = pp_floatx4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        quad_transpose(t);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + wm + mt * 32 + qp + k;
          pp_floatx4 v = t[k] * sc[nt] + bias[nt];
          if (a.residual) v += res[nt][k];
          if (m < a.M) {
            chk += (v[0] + v[1]) + (v[2] + v[3]);
            if (a.relu == HP_ACT_RELU) v = __builtin_elementwise_max(v, pp_floatx4{0.f, 0.f, 0.f, 0.f});
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#if defined(HP_PP_NT_STORES)   // experiment: non-temporal output stores
            __builtin_nontemporal_store(v, reinterpret_cast<pp_floatx4*>(a.y + m * a.Cout + n));
#elif defined(HP_PP_SC1_STORES)  // experiment: write-through (sc1) output stores
            {
              typedef unsigned int pp_uintx4 __attribute__((ext_vector_type(4)));
              const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0x7FFFFFFF, 0x00020000);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_uintx4, v), yr, (int)((m * a.Cout + n) * 4), 0, 16);
            }
#else
            *reinterpret_cast<pp_floatx4*>(a.y + m * a.Cout + n) = v;
#endif
          }
        }
      }
    }
    conv_report_nonfinite(a, chk);
    if (a.amax_out) {  // as conv_epilogue.h: wave maximum, look before the atomic, words spread over L2 channels
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
      if (lane == 0 && amax > 0.f) {
        const unsigned mine = __float_as_uint(amax);
        unsigned* const slot = a.amax_out + (blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride;
        if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
      }
    }
  } else {
    // fp16 output: a lane's wave tile row is 4 pieces of 8 couts (nt, g) = 16 B each; the same quad transpose (in fp32, so
    // that the single rounding to fp16 stays where it was) gives lane j piece j of the quad's four pixels: store k writes 8
    // whole 128-B lines (pieces 0 / 1 of the two half-waves = couts wn .. wn + 31, pieces 2 / 3 = wn + 32 .. wn + 63)
    static_assert(NT == 2, "the piece map below assumes two N tiles per wave (the fp16 plan launches NT = 2 only)");
    const int qj = lane & 3, qp = px & ~3;
    const _Float16* const resp = reinterpret_cast<const _Float16*>(a.residual);
    _Float16* const y = reinterpret_cast<_Float16*>(a.y);
    const int n = n0 + wn + (qj >> 1) * 32 + h16 + 8 * (qj & 1);  // this lane's 8 couts after the transpose
    pp_floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (a.bias) {
      b0 = *reinterpret_cast<const pp_floatx4*>(a.bias + n);
      b1 = *reinterpret_cast<const pp_floatx4*>(a.bias + n + 4);
    }
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt) {
      pp_halfx8 res[4];
      if (resp) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + wm + mt * 32 + qp + k;
          res[k] = *reinterpret_cast<const pp_halfx8*>(resp + (m < a.M ? m : 0) * a.Cout + n);
        }
      }
      if (mt == 1) between();
      float t[4][8];  // [piece j = 2 nt + g][8 couts]
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) t[j][c] = acc[mt][j >> 1][8 * (j & 1) + c];
#pragma unroll
      for (int c = 0; c < 8; ++c) quad_transpose4(qj, t[0][c], t[1][c], t[2][c], t[3][c]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // t[k] = this lane's 8 couts of pixel row qp + k
        const int64_t m = m0 + wm + mt * 32 + qp + k;
        pp_floatx4 v0 = {t[k][0], t[k][1], t[k][2], t[k][3]}, v1 = {t[k][4], t[k][5], t[k][6], t[k][7]};
        v0 += b0; v1 += b1;
        if (resp) {
          const pp_halfx8 rr = res[k];
#pragma unroll
          for (int q = 0; q < 4; ++q) { v0[q] += (float)rr[q]; v1[q] += (float)rr[4 + q]; }
        }
        if (a.relu) {
          v0 = __builtin_elementwise_max(v0, pp_floatx4{0.f, 0.f, 0.f, 0.f});
          v1 = __builtin_elementwise_max(v1, pp_floatx4{0.f, 0.f, 0.f, 0.f});
        }
        pp_halfx8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[q] = (_Float16)v0[q]; o[4 + q] = (_Float16)v1[q]; }
        if (m < a.M) *reinterpret_cast<pp_halfx8*>(y + m * a.Cout + n) = o;
      }
    }
  }
}

// a.Kpad: MODE_F16 = elements per weight row ([Cout][Kpad] halves, K = (tap, c)); MODE_SPLIT unused (rows are 18 Cin halves)
template <int MODE, bool PRE, int NPC, int NT, bool WIDE = false>
__global__ __launch_bounds__(kPPThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_pp(ConvArgs a, int P) {
  using T = PP<MODE, NT, WIDE>;
  constexpr int BN = T::BN, BMT = T::BMT, NPB = T::NPB;
  static_assert(!WIDE || NT == 2, "the 512 x 64 tile: every wave covers all 64 couts (two N tiles)");
  constexpr int CKC = T::CKC, TPR = T::TPR, PROWS = T::PROWS, ESZ = T::ESZ;
  constexpr int NB = BN * 8 / kPPThreads;  // 16-B weight pieces per thread and tap (2)
  constexpr int BROWS = kPPThreads / 8;    // weight rows per staging pass (64)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const patch = reinterpret_cast<_Float16*>(lds_raw);  // [2][P][LDH]
  _Float16* const Bs = patch + NPB * P * LDH;                      // [2][BN][LDH]
  _Float16* const zrow = Bs + 2 * BN * LDH;                        // [LDH] zeros: what a masked tap reads
  float act_sx = 1.f, act_inv = 1.f;  // ConvArgs::amax_in: power-of-two scale of the staged activations (split mode)
  if constexpr (MODE == MODE_SPLIT) conv_act_scale(a, act_sx, act_inv);
  // pre-activation BN + ReLU vectors in LDS (fetched from global memory inside the L segment of a chunk's last tap they
  // stalled the whole phase for an HBM / L2 round trip: 1918 instead of 1690 cycles per tap on the PRE layers)
  float* const pre_lds = reinterpret_cast<float*>(zrow + LDH);      // split: [Cin] scale, [Cin] shift (fp32); f16: halves

#ifdef HP_PP_STAMPS
  const unsigned long long st_k0 = __builtin_readcyclecounter();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H, Cin = a.Cin;
  const int ncc_all = Cin / CKC;

  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(a.M * Cin * ESZ), 0x00020000);
  const int64_t wrow = MODE == MODE_SPLIT ? (int64_t)18 * Cin : (int64_t)a.Kpad;  // halves per cout
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((int64_t)a.Cout * wrow * 2), 0x00020000);

  // ---- thread constants that do not depend on the tile
  // patch staging: row = pr0 + PROWS j, channels 8 pk .. 8 pk + 7 of the chunk
  const int pk = tid % TPR, pr0 = tid / TPR;
  // split: this thread's 8 channels land at halves [8 pk, 8 pk + 8) (hi) and 32 + [8 pk, ...) (lo); f16: at [8 pk, ...)
  _Float16* const Pst = patch + pr0 * LDH + 8 * pk;
  // weight staging: row = br0 + 64 i, 16-B piece bk of the 128-B row
  const int bk = tid & 7, br0 = tid >> 3;
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
  // fragment bases
  // waves w and w + 4 share a SIMD; WIDE: eight 64-row slabs of the 512-row tile, every wave all 64 couts
  const int wm = WIDE ? (wave & 3) * 64 + (wave >> 2) * 256 : ((wave & 3) >> 1) * 64 + (wave >> 2) * 128;
  const int wn = WIDE ? 0 : (wave & 1) * (BN / 2);
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  // MFMA row i of a 32-cout block multiplies weight row sigma(i): the lane's 16 accumulator rows are then 16 consecutive couts
  const int srow = 16 * ((frow >> 2) & 1) + 4 * (frow >> 3) + (frow & 3);
  const _Float16* const Bfr = Bs + (wn + srow) * LDH + fk;
  const _Float16* Afr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) Afr[mt] = patch + (wm + mt * 32 + frow + W + 1) * LDH + fk;
  const _Float16* const Zfr = zrow + fk;
  const bool odd = wave >= 4;  // waves 4-7 run one phase behind waves 0-3

  // ---- PERSISTENT loop over work items (round 4).  The grid holds at most one workgroup per CU; workgroup b walks the
  // virtual blocks b, b + G, b + 2 G, ... of the former one-item-per-workgroup grid (same XCD for all of them: G is a
  // multiple of 8).  What that buys: the first loads of the NEXT item (its patch chunk and the weights of taps 0 / 1) are
  // issued before the epilogue of the current one and land under it, the epilogue's stores drain under the next K loop,
  // and the per-launch set-up above (LDS carve-up, BN vectors, activation scale) is paid once per CU, not once per tile.
  const int nvb = 8 * (a.sk_regular / 8 + (a.sk_tail_items + 7) / 8);
  // loop state of the item being multiplied (overwritten by the decode of the next one once the K loop is over)
  int64_t gp0 = 0;
  unsigned wvoff[NB];
  unsigned vmask[MT];
  int cc_begin = 0, ncc = 0;
  struct Item { int lin, slice, n0; bool split; int64_t m0; };
  // Everything an item's set-up and epilogue derive from the thread id is RE-derived per item from a laundered copy of it:
  // left to itself the compiler hoists those values out of the item loop, keeps them live across the K loop -- where every
  // register is taken -- and spills them to scratch (which, besides its cost, keeps hipGraph replay off: DESIGN.md 4.5).
  auto opaque_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
  // The same for the kernel arguments: the set-up and the epilogue read them through a laundered pointer to the kernarg
  // segment (ConvArgs is its first member), i.e. with scalar loads where they are used, instead of holding ~60 SGPRs of
  // pointers and sizes live across the K loop (SGPR spills land in VGPR lanes and push the VGPR file over).
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
  auto opaque_args = [&]() { KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p; };
#else  // host pass of the single-source compile: never executed
  typedef const ConvArgs* KArgs;
  auto opaque_args = [&]() { return &a; };
#endif
  auto decode = [&](int vb, Item& it) -> bool {
    if (vb >= nvb) return false;
    const KArgs ka = opaque_args();
    const ConvArgs a = *ka;  // shadows the kernel argument inside this lambda
    const int t_ = opaque_tid();
    const int br0 = t_ >> 3, bk = t_ & 7, frow = t_ & 31, wave_ = t_ >> 6;
    const int wm = WIDE ? (wave_ & 3) * 64 + (wave_ >> 2) * 256 : ((wave_ & 3) >> 1) * 64 + (wave_ >> 2) * 128;
    const int pr0 = t_ / TPR;
    const int rpx = a.sk_regular / 8, tpx = (a.sk_tail_items + 7) / 8;
    const int xcd = vb % 8, li = vb / 8;
    it.slice = 0; it.split = false;
    if (li < rpx) {
      it.lin = xcd * rpx + li;
    } else {
      const int ti = xcd * tpx + (li - rpx);
      if (li - rpx >= tpx || ti >= a.sk_tail_items) return false;
      it.lin = a.sk_regular + ti / a.sk_S;
      it.slice = ti % a.sk_S;
      it.split = a.sk_S > 1;
    }
    const int tile_m = fdiv(it.lin, a.fd_tn), tile_n = it.lin - tile_m * a.tiles_n;
    it.m0 = (int64_t)tile_m * BMT;
    it.n0 = tile_n * BN;
    cc_begin = it.split ? it.slice * ncc_all / a.sk_S : 0;
    ncc = it.split ? (it.slice + 1) * ncc_all / a.sk_S : ncc_all;  // end of this item's chunk range
    gp0 = it.m0 - (W + 1) + pr0;
#pragma unroll
    for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(it.n0 + br0 + BROWS * i) * wrow + 8 * bk) * 2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {  // validity of the 9 taps per fragment row
      const int64_t g = it.m0 + wm + mt * 32 + frow;
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
    return true;
  };
  auto patch_voff = [&](int j) -> unsigned {
    const int64_t gp = gp0 + PROWS * j;
    return (pr0 + PROWS * j < P && gp >= 0 && gp < a.M) ? (unsigned)((gp * Cin + 8 * pk) * ESZ) : kOob;
  };

  pp_floatx16 acc[MT][NT];

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
  // first loads of the item the loop state describes: patch of its first chunk, weights of taps 0 and 1 -> registers
  auto issue_first_loads = [&]() {
#pragma unroll
    for (int j = 0; j < NPC; ++j) load_patch(j, cc_begin);
    load_b(0, cc_begin, 0);
    load_b(1, cc_begin, 1);
  };

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
    if (!WIDE && tap == 8 && next_chunk) store_patch(cc + 1, 1 - PQ);
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
    pp_halfx8 fa[4][MT], fb[4][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      // the select stays HERE (two v_cndmask per tap): hoisted out of the K loop it is 36 addresses per item plus the 36
      // tile-independent ones they are chosen from, all live across the K loop -- the registers the persistent loop needs
      unsigned vm = vmask[i];
      asm volatile("" : "+v"(vm));
      const _Float16* const Ab = ((vm >> tap) & 1u) ? Afr[i] + PQ * P * LDH + d * LDH : Zfr;
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
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y[ni], x[mi], acc[mi][ni], 0, 0, 0);  // D[cout][pixel]
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
    using Q = std::integral_constant<int, WIDE ? 0 : C0>;  // patch buffer
    tap_step(cc, 0, E{}, Q{}); tap_step(cc, 1, O{}, Q{}); tap_step(cc, 2, E{}, Q{});
    tap_step(cc, 3, O{}, Q{}); tap_step(cc, 4, E{}, Q{}); tap_step(cc, 5, O{}, Q{});
    tap_step(cc, 6, E{}, Q{}); tap_step(cc, 7, O{}, Q{}); tap_step(cc, 8, E{}, Q{});
    if (WIDE && cc + 1 < ncc) {
      // single patch buffer: this wave group is past its C of tap 8, i.e. BOTH groups have read their tap-8 fragments (the
      // other group's L ran beside that C).  Group A stores its rows of the next chunk while B multiplies tap 8; B stores
      // while A waits; then A reads tap 0.  The same three statements for both groups, one barrier apart as everything
      // else: two phases without an MFMA per SIMD and chunk swap (~1.4 k of the ~17 k cycles of a chunk).
      store_patch(cc + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  Item cur{};
  int vb = blockIdx.x;
  bool have = decode(vb, cur);
  if (have) issue_first_loads();
  while (have) {
    // ---- prologue of `cur`: its first loads are in flight (issued above, or before the previous item's epilogue)
#ifdef HP_PP_STAMPS
    const unsigned long long st_p0 = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    store_patch(cc_begin, 0);
    store_b(0, 0);
    load_b(0, cc_begin, 2);
    __syncthreads();
#ifdef HP_PP_STAMPS
    const unsigned long long st_t0 = __builtin_readcyclecounter(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (odd) __builtin_amdgcn_s_barrier();
    int cc = cc_begin;
    for (; cc + 1 < ncc; cc += 2) {
      chunk(cc, std::integral_constant<int, 0>{});
      chunk(cc + 1, std::integral_constant<int, 1>{});
    }
    if (cc < ncc) chunk(cc, std::integral_constant<int, 0>{});
    if (!odd) __builtin_amdgcn_s_barrier();  // both wave groups are past their last fragment reads: the LDS buffers are free
#ifdef HP_PP_STAMPS
    if (tid == 0) {
      atomicAdd(&g_pp_stamps[0], __builtin_readcyclecounter() - st_t0);
      atomicAdd(&g_pp_stamps[1], __builtin_amdgcn_s_memrealtime() - st_r0);
      atomicAdd(&g_pp_stamps[2], (unsigned long long)((ncc - cc_begin) * 9));
      atomicAdd(&g_pp_stamps[3], 1ull);
      atomicAdd(&g_pp_stamps[4], st_t0 - st_p0);  // prologue of this item (waiting for its prefetched loads, staging)
      if (vb == (int)blockIdx.x) atomicAdd(&g_pp_stamps[6], st_p0 - st_k0);  // per-workgroup set-up before the first item
    }
    const unsigned long long st_e0 = __builtin_readcyclecounter();
#endif
    // ---- the next item's loop state and first loads, then this item's epilogue (whose stores drain under the next K loop)
    Item nxt{};
    vb += (int)gridDim.x;
    bool finish = true, issued = false;
    const ConvArgs ea = *opaque_args();  // the epilogue's view of the arguments (see opaque_args)
    if (cur.split) finish = splitk_reduce_sc1<BMT, BN, MT, NT, kPPThreads>(ea, acc, cur.lin - ea.sk_regular, cur.slice);
    auto prefetch_next = [&]() {  // overwrites the loop state (the K loop of `cur` is over), keeps cur's epilogue constants
      have = decode(vb, nxt);
      if (have) issue_first_loads();
      issued = true;
    };
    if (finish) {
      const int t_ = opaque_tid(), wave_ = t_ >> 6;
      pp_epilogue_direct<MODE, NT>(ea, acc, cur.m0, cur.n0, WIDE ? (wave_ & 3) * 64 + (wave_ >> 2) * 256 : ((wave_ & 3) >> 1) * 64 + (wave_ >> 2) * 128,
                                   WIDE ? 0 : (wave_ & 1) * (BN / 2), act_inv, t_ & 63, prefetch_next);
    }
    if (!issued) prefetch_next();
#ifdef HP_PP_STAMPS
    if (tid == 0) atomicAdd(&g_pp_stamps[5], __builtin_readcyclecounter() - st_e0);  // slab hand-off + epilogue issue
#endif
    cur = nxt;
  }
}

// (Round 6: `conv3x3_pp1`, the same tile by FOUR waves -- one per SIMD with the whole register file, 128 x 64 wave tiles, 24 fragment
// reads per 48 MFMAs, a half-tap software pipeline -- was built here, parity-green, and deleted: 2300 - 2600 cycles per tap against
// 1650 - 1750.  The non-MFMA instructions of a tap do not hide in the shadow of the same wave's MFMAs on this machine; only a second
// wave of the SIMD issuing them keeps the matrix pipe busy.  profiles/r06_pp1_one_wave_per_simd.txt; the source is in this file's
// history.)

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / STRIDE-2 / pad-1 layers (the first conv of layer2 / 3 / 4) on the same skeleton (round 4).  conv3x3s2_split_f32
// (conv_split.hip) keeps ONE patch buffer: every (phase, chunk) of the space-to-depth walk ends in barrier -> convert + store
// the next patch -> barrier with the matrix pipe idle, a patch feeds 2.25 taps on average, and prologue / epilogue are exposed
// (one item per workgroup): SQ counters put the matrix pipe at 0.15 busy (profiles/r04a_conv_pmc_summary.txt).  Here the walk
// (same entry order e -> (phase, chunk, tap), same pre-split weights: s2_entry / s2_tap of conv_split.hip restated below) runs
// as ping-pong taps on a DOUBLE-buffered patch: the patch of sub-patch u + 1 is converted and stored by the L segment of
// u's last tap while the other wave group multiplies, its loads having been issued when u's own patch was stored; items are
// persistent, the epilogue is pp_epilogue_direct.
__device__ __forceinline__ void pp_s2_entry(int e, int n, int& ph, int& c, int& t, int& T) {
  if (e < n) { ph = 0; c = e; t = 0; T = 1; }
  else if (e < 3 * n) { ph = 1; c = (e - n) >> 1; t = (e - n) & 1; T = 2; }
  else if (e < 5 * n) { ph = 2; c = (e - 3 * n) >> 1; t = (e - 3 * n) & 1; T = 2; }
  else { ph = 3; c = (e - 5 * n) >> 2; t = (e - 5 * n) & 3; T = 4; }
}
__device__ __forceinline__ void pp_s2_tap(int ph, int t, int& di, int& dj, int& kh, int& kw) {
  const int p = ph >> 1, q = ph & 1;
  const int ti = (p && q) ? t >> 1 : t, tj = (p && q) ? t & 1 : t;
  di = p ? ti - 1 : 0; kh = p ? 2 * ti : 1;  // phase row 1 holds the input rows of kh = 0 (one row up) and kh = 2
  dj = q ? tj - 1 : 0; kw = q ? 2 * tj : 1;
}

struct PPS2 {
  static constexpr int BN = 128, NT = 2, CKC = 32, PROWS = kPPThreads / 4;
  static int P(int Wo) { return BM + Wo + 1; }
  static int npc(int Wo) { return (P(Wo) + PROWS - 1) / PROWS; }
  static constexpr int kPreFloats = 2 * 512;
  static size_t lds_bytes(int Wo) { return ((size_t)2 * P(Wo) * LDH + 2 * (size_t)BN * LDH + LDH) * 2 + kPreFloats * 4; }
};

template <bool PRE, int NPC>
__global__ __launch_bounds__(kPPThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3s2_pp(ConvArgs a, int P) {
  constexpr int BN = PPS2::BN, NT = PPS2::NT, CKC = PPS2::CKC, PROWS = PPS2::PROWS;
  constexpr int NB = BN * 8 / kPPThreads, BROWS = kPPThreads / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const patch = reinterpret_cast<_Float16*>(lds_raw);  // [2][P][LDH]
  _Float16* const Bs = patch + 2 * P * LDH;                        // [2][BN][LDH]
  _Float16* const zrow = Bs + 2 * BN * LDH;
  float* const pre_lds = reinterpret_cast<float*>(zrow + LDH);
  float act_sx = 1.f, act_inv = 1.f;
  conv_act_scale(a, act_sx, act_inv);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H, C = a.Cin, Wo = a.Wo, Ho = a.Ho;
  const int n = C / CKC;  // 32-channel chunks of the layer; an item covers chunks [c0, c0 + nl) (all of them unless K-sliced)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((a.M / (Ho * Wo)) * (int64_t)H * W * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((size_t)a.Cout * 9 * C * 4), 0x00020000);
  // shortcut items (ConvArgs::sc_w): the same rows of a second weight array; entries [0, nl) of the walk are its centre taps
  const __amdgpu_buffer_rsrc_t wrsrc_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sc_w ? a.sc_w : a.w), 0, (int)((size_t)a.Cout * 9 * C * 4), 0x00020000);

  const int pk = tid & 3, pr0 = tid >> 2;
  _Float16* const Pst = patch + pr0 * LDH + 8 * pk;
  const int bk = tid & 7, br0 = tid >> 3;
  _Float16* const Bst = Bs + br0 * LDH + 8 * bk;
  if (tid < LDH / 2) reinterpret_cast<unsigned*>(zrow)[tid] = 0u;
  if (PRE) {
    for (int i = tid; i < C; i += kPPThreads) { pre_lds[i] = a.pre_scale[i] * act_sx; pre_lds[C + i] = a.pre_shift[i] * act_sx; }
    __syncthreads();
  }
  const int wm = ((wave & 3) >> 1) * 64 + (wave >> 2) * 128, wn = (wave & 1) * (BN / 2);
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  const int srow = 16 * ((frow >> 2) & 1) + 4 * (frow >> 3) + (frow & 3);  // see conv3x3_pp: 16 consecutive couts per lane
  const _Float16* const Bfr = Bs + (wn + srow) * LDH + fk;
  const _Float16* Afr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) Afr[mt] = patch + (wm + mt * 32 + frow + Wo + 1) * LDH + fk;
  const _Float16* const Zfr = zrow + fk;
  const bool odd = wave >= 4;

  auto opaque_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
  auto opaque_args = [&]() { KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p; };
#else
  auto opaque_args = [&]() { return &a; };
#endif

  // ---- item state.  K slices (tail items of conv_split_plan_tail, as in conv3x3_pp) cut the CHUNK range: a slice walks all
  // four phases of its chunks, so its entries are whole patches; slice bounds are even (pairs of chunks = the planner's unit)
  const int nvb = 8 * (a.sk_regular / 8 + (a.sk_tail_items + 7) / 8 + (a.sc_items + 7) / 8);
  int c0 = 0, nl = n, nent = 9 * n, npatch = 4 * n;
  bool sc_item = false;  // the item being multiplied is a shortcut item: phase 0 only (one entry, one sub-patch per chunk)
  unsigned pbase[NPC], pflag[NPC];  // input pixel (2 oh, 2 ow) of the thread's patch rows; bit 0 inside, 1: row 2 oh + 1, 2: col 2 ow + 1
  unsigned wvoff[NB];
  unsigned vmask[MT];
  struct Item { int lin, slice, n0; bool split, sc; int64_t m0; };
  auto decode = [&](int& vb, Item& it) -> bool {  // advances vb past empty slots of the tail / shortcut regions
    const ConvArgs a = *opaque_args();
    for (;; vb += (int)gridDim.x) {
      if (vb >= nvb) return false;
      const int rpx = a.sk_regular / 8, tpx = (a.sk_tail_items + 7) / 8;
      const int xcd = vb % 8, li = vb / 8;
      it.slice = 0; it.split = false; it.sc = false;
      if (li < rpx) {
        it.lin = xcd * rpx + li;
      } else if (li < rpx + tpx) {
        const int ti = xcd * tpx + (li - rpx);
        if (ti >= a.sk_tail_items) continue;  // (not the end: this workgroup's shortcut items may follow)
        it.lin = a.sk_regular + ti / a.sk_S;
        it.slice = ti % a.sk_S;
        it.split = a.sk_S > 1;
      } else {  // shortcut items follow the tail: they fill the CUs the last (partial) round of 3x3 items leaves idle
        const int spx = (a.sc_items + 7) / 8, si = xcd * spx + (li - rpx - tpx);
        if (si >= a.sc_items) continue;
        it.lin = si;
        it.sc = true;
      }
      break;
    }
    const int lin = it.lin;
    {
      const int U = n / 2;
      const int u0 = it.split ? it.slice * U / a.sk_S : 0, u1 = it.split ? (it.slice + 1) * U / a.sk_S : U;
      sc_item = it.sc;
      c0 = 2 * u0; nl = 2 * (u1 - u0); nent = (it.sc ? 1 : 9) * nl; npatch = (it.sc ? 1 : 4) * nl;
    }
    const int t_ = opaque_tid();
    const int br0 = t_ >> 3, bk = t_ & 7, frow = t_ & 31, wave_ = t_ >> 6, pr0 = t_ >> 2, pk = t_ & 3;
    const int wm = ((wave_ & 3) >> 1) * 64 + (wave_ >> 2) * 128;
    const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
    it.m0 = (int64_t)tile_m * BM;
    it.n0 = tile_n * BN;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int64_t g = it.m0 - (Wo + 1) + pr0 + PROWS * j;
      pbase[j] = 0; pflag[j] = 0;
      if (pr0 + PROWS * j < P && g >= 0 && g < a.M) {
        const int img = fdiv((int)g, a.fd_howo);
        const int rem = (int)g - img * (Ho * Wo);
        const int oh = fdiv(rem, a.fd_wo), ow = rem - oh * Wo;
        pbase[j] = (unsigned)(((((int64_t)img * H + 2 * oh) * W + 2 * ow) * C + 8 * pk) * 4);
        pflag[j] = 1u | (2 * oh + 1 < H ? 2u : 0u) | (2 * ow + 1 < W ? 4u : 0u);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(it.n0 + br0 + BROWS * i) * (18 * C) + 8 * bk) * 2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t g = it.m0 + wm + mt * 32 + frow;
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
    return true;
  };

  pp_floatx16 acc[MT][NT];
  pp_floatx4 pr[NPC][2];
  pp_halfx8 rb[2][NB];
  auto patch_voff = [&](int j, int ph) -> unsigned {
    const unsigned need = 1u | ((ph & 2) ? 2u : 0u) | ((ph & 1) ? 4u : 0u);
    return (pflag[j] & need) == need ? pbase[j] : kOob;
  };
  auto load_patch = [&](int u) {  // sub-patch u = ph * nl + (c - c0) of the item: raw fp32 rows -> registers
    const int ph = u / nl, c = c0 + (u - ph * nl);
    const unsigned soff = (unsigned)((((ph >> 1) * W + (ph & 1)) * C + c * CKC) * 4);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const unsigned vo = patch_voff(j, ph);
      pr[j][0] = ldf4(xrsrc, vo, soff);
      pr[j][1] = ldf4(xrsrc, vo, soff + 16);
    }
  };
  auto store_patch = [&](int u) {  // registers -> patch buffer u & 1 (BN + ReLU prologue, activation scale, hi / lo split)
    const int ph = u / nl, c = c0 + (u - ph * nl);
    _Float16* const dst = Pst + (u & 1) * P * LDH;
    pp_floatx4 ps[2], pb[2];
    if (PRE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ps[h] = *reinterpret_cast<const pp_floatx4*>(pre_lds + c * CKC + 8 * pk + 4 * h);
        pb[h] = *reinterpret_cast<const pp_floatx4*>(pre_lds + C + c * CKC + 8 * pk + 4 * h);
      }
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      if (pr0 + PROWS * j < P) {
        const bool real = patch_voff(j, ph) != kOob;
        pp_halfx4 hi[2], lo[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          pp_floatx4 v = pr[j][h];
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
  };
  auto load_b = [&](int set, int e) {  // e = local entry of the item -> row of the layer's entry-ordered weights
    const int e2 = e < nent ? e : nent - 1;
    int ph, cl, t, T;
    pp_s2_entry(e2, nl, ph, cl, t, T);
    const int eg = (ph == 0 ? 0 : ph == 1 ? n : ph == 2 ? 3 * n : 5 * n) + (c0 + cl) * T + t;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[set][i] = ldh8(sc_item ? wrsrc_sc : wrsrc, wvoff[i], (unsigned)(eg * 128));
  };
  auto store_b = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<pp_halfx8*>(Bst + buf * BN * LDH + BROWS * i * LDH) = rb[set][i];
  };
  auto issue_first_loads = [&]() {
    load_patch(0);
    load_b(0, 0);
    load_b(1, 1);
  };

  // one entry (tap): L segment, barrier, C segment, barrier.  Pb = weight buffer parity (compile time), patch buffer u & 1
  auto entry_step = [&](int e, auto par) {
    constexpr int Pb = decltype(par)::value;
    int ph, c, t, T;
    pp_s2_entry(e, nl, ph, c, t, T);  // c = chunk index inside the item
    int di, dj, kh, kw;
    pp_s2_tap(ph, t, di, dj, kh, kw);
    const int d = di * Wo + dj, bit = kh * 3 + kw;
    const int u = ph * nl + c;
    // ---- L
    store_b(1 - Pb, 1 - Pb);
    load_b(1 - Pb, e + 3);
    if (t == T - 1 && u + 1 < npatch) {  // wave-uniform: the next sub-patch goes into the other buffer now ...
      store_patch(u + 1);
      if (u + 2 < npatch) load_patch(u + 2);  // ... and its successor's loads take the registers
    }
    pp_halfx8 fa[4][MT], fb[4][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      unsigned vm = vmask[i];
      asm volatile("" : "+v"(vm));
      const _Float16* const Ab = ((vm >> bit) & 1u) ? Afr[i] + (u & 1) * P * LDH + d * LDH : Zfr;
#pragma unroll
      for (int q = 0; q < 4; ++q) fa[q][i] = *reinterpret_cast<const pp_halfx8*>(Ab + q * 16);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) fb[q][i] = *reinterpret_cast<const pp_halfx8*>(Bfr + Pb * BN * LDH + i * 32 * LDH + q * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- C
    auto mm = [&](const pp_halfx8 (&x)[MT], const pp_halfx8 (&y)[NT]) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y[ni], x[mi], acc[mi][ni], 0, 0, 0);
    };
    mm(fa[0], fb[0]); mm(fa[0], fb[2]); mm(fa[2], fb[0]);
    mm(fa[1], fb[1]); mm(fa[1], fb[3]); mm(fa[3], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  Item cur{};
  int vb = blockIdx.x;
  bool have = decode(vb, cur);
  if (have) issue_first_loads();
  while (have) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    store_patch(0);
    if (npatch > 1) load_patch(1);
    store_b(0, 0);
    load_b(0, 2);
    __syncthreads();
    if (odd) __builtin_amdgcn_s_barrier();
    for (int e = 0; e < nent; e += 2) {  // 9 n entries, n even
      entry_step(e, std::integral_constant<int, 0>{});
      entry_step(e + 1, std::integral_constant<int, 1>{});
    }
    if (!odd) __builtin_amdgcn_s_barrier();
    Item nxt{};
    vb += (int)gridDim.x;
    bool issued = false, finish = true;
    ConvArgs ea = *opaque_args();
    if (cur.sc) {  // the shortcut's output, bias, activation and scale-back factors (behind its own weights)
      ea.y = ea.sc_y; ea.bias = ea.sc_bias; ea.relu = ea.sc_relu; ea.w = ea.sc_w; ea.residual = nullptr; ea.amax_out = ea.sc_amax_out;
    }
    if (cur.split) finish = splitk_reduce_sc1<BM, BN, MT, NT, kPPThreads>(ea, acc, cur.lin - ea.sk_regular, cur.slice);
    auto prefetch_next = [&]() {
      have = decode(vb, nxt);
      if (have) issue_first_loads();
      issued = true;
    };
    if (finish) {
      const int t_ = opaque_tid(), wave_ = t_ >> 6;
      pp_epilogue_direct<MODE_SPLIT, NT>(ea, acc, cur.m0, cur.n0, ((wave_ & 3) >> 1) * 64 + (wave_ >> 2) * 128, (wave_ & 1) * (BN / 2), act_inv,
                                         t_ & 63, prefetch_next);
    }
    if (!issued) prefetch_next();
    cur = nxt;
  }
}


template <int MODE, bool PRE, int NPC, int NT, bool WIDE = false>
int launch_pp_variant(ConvArgs args, hipStream_t stream) {
  using T = PP<MODE, NT, WIDE>;
  constexpr int BN = T::BN, BM = T::BMT;  // shadows the 256-row constant in this launcher
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& s) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp<MODE, PRE, NPC, NT, WIDE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));  // + the static ticket word
        s.spills = note_kernel(reinterpret_cast<const void*>(&conv3x3_pp<MODE, PRE, NPC, NT, WIDE>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = args.Cout / BN;
  args.fd_howo = make_fastdiv((unsigned)(args.H * args.W));
  args.fd_wo = make_fastdiv((unsigned)args.W);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int rc = conv_split_plan_tail(args, args.tiles_m * args.tiles_n, args.Cin / T::CKC, (size_t)BM * BN, 1, stream);
  if (rc) return rc;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8;
  // persistent grid: one workgroup per CU at most (the LDS admits no second one), a multiple of 8 so that a workgroup's
  // virtual blocks stay on its XCD; HP_PP_GRID (debug.h) overrides the cap: tools/conv_fuzz.py makes every workgroup walk many items
  const int cap = dbg(DBG_PP_GRID) > 0 ? std::max(8, dbg(DBG_PP_GRID) / 8 * 8) : conv_num_cus() / 8 * 8;
  const int grid = std::min(8 * per_xcd, cap);
  hipLaunchKernelGGL((conv3x3_pp<MODE, PRE, NPC, NT, WIDE>), dim3(grid), dim3(kPPThreads), T::lds_bytes(args.W), stream, args,
                     T::P(args.W));
  return check_launch("conv3x3_pp");
}

template <int MODE, bool PRE, int NT>
int launch_pp_nt(const ConvArgs& a, hipStream_t stream) {
  const int npc = PP<MODE, NT>::npc(a.W);
  if (npc <= 3) return launch_pp_variant<MODE, PRE, 3, NT>(a, stream);
  if (npc == 4) return launch_pp_variant<MODE, PRE, 4, NT>(a, stream);
  if (npc <= 6) return launch_pp_variant<MODE, PRE, 6, NT>(a, stream);
  return fail(HP_ERR_ARG, "conv3x3_pp: map too wide for the staged patch");
}

template <int MODE, int NT>
bool pp_shape_ok(int W, int Cin, int Cout, int stride, int pad, int kh, int kw) {
  using T = PP<MODE, NT>;
  return !dbg(DBG_CONV_NO_PP) && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Cout % T::BN == 0 && Cin % T::CKC == 0 && Cin <= 512 &&
         T::npc(W) <= 6 && T::lds_bytes(W) <= 159 * 1024;
}
// 512 x 64 tiles (WIDE) for the 64-channel layers (256 x 64 tiles on this skeleton, NT = 1, lost to them: CHANGELOG round 4)
bool pp_wide64_ok(const ConvArgs& a, int kh, int kw) {
  using T = PP<MODE_SPLIT, 2, true>;
  return !dbg(DBG_CONV_NO_PP) && kh == 3 && kw == 3 && a.stride == 1 && a.pad == 1 && a.Cout % 128 != 0 && a.Cout % 64 == 0 && a.Cin % T::CKC == 0 &&
         a.Cin <= 512 && T::npc(a.W) <= 6 && T::lds_bytes(a.W) <= 159 * 1024;
}

template <bool PRE, int NPC>
int launch_pp_s2_variant(ConvArgs args, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& s) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_pp<PRE, NPC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         159 * 1024));
        s.spills = note_kernel(reinterpret_cast<const void*>(&conv3x3s2_pp<PRE, NPC>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = args.Cout / PPS2::BN;
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  // K slices over PAIRS of 32-channel chunks (a slice's entry count stays even); same policy as conv3x3_pp
  const int rc = conv_split_plan_tail(args, args.tiles_m * args.tiles_n, args.Cin / (2 * PPS2::CKC), (size_t)BM * PPS2::BN, 1, stream);
  if (rc) return rc;
  args.sc_items = args.sc_w ? args.tiles_m * args.tiles_n : 0;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8 + (args.sc_items + 7) / 8;
  const int grid = std::min(8 * per_xcd, conv_num_cus() / 8 * 8);
  hipLaunchKernelGGL((conv3x3s2_pp<PRE, NPC>), dim3(grid), dim3(kPPThreads), PPS2::lds_bytes(args.Wo), stream, args, PPS2::P(args.Wo));
  return check_launch("conv3x3s2_pp");
}

}  // namespace

// stride-2 entry (fp32 / split mode): a.w = weights split by conv_split_transform_weights(..., stride 2)
bool conv_pp_s2_applicable(const ConvArgs& a, int kh, int kw) {
  return !dbg(DBG_CONV_NO_PP) && !dbg(DBG_CONV_NO_PP_S2) && kh == 3 && kw == 3 && a.stride == 2 && a.pad == 1 && a.Cin % 64 == 0 && a.Cin <= 512 && a.Cout % PPS2::BN == 0 &&
         a.Ho == (a.H - 1) / 2 + 1 && a.Wo == (a.W - 1) / 2 + 1 && PPS2::npc(a.Wo) <= 3 && PPS2::lds_bytes(a.Wo) <= 159 * 1024;
}

int launch_conv_pp_s2_split(const ConvArgs& a, hipStream_t stream) {
  return a.pre_scale ? launch_pp_s2_variant<true, 3>(a, stream) : launch_pp_s2_variant<false, 3>(a, stream);
}

// fp32 (split-fp16) entry: a.w = weights split by conv_split_transform_weights
bool conv_pp_split_applicable(const ConvArgs& a, int kh, int kw) {
  if (pp_wide64_ok(a, kh, kw)) return true;
  return pp_shape_ok<MODE_SPLIT, 2>(a.W, a.Cin, a.Cout, a.stride, a.pad, kh, kw);
}

int launch_conv_pp_split(const ConvArgs& a, hipStream_t stream) {
  if (pp_wide64_ok(a, 3, 3))
    return a.pre_scale ? launch_pp_variant<MODE_SPLIT, true, 6, 2, true>(a, stream) : launch_pp_variant<MODE_SPLIT, false, 6, 2, true>(a, stream);
  return a.pre_scale ? launch_pp_nt<MODE_SPLIT, true, 2>(a, stream) : launch_pp_nt<MODE_SPLIT, false, 2>(a, stream);
}

// fp16 entry (the fp16 plan): packed weights [Cout][Kpad] with K = (tap, c)
// fp16 plan, 64-channel layers with ONE 64-channel chunk (Cin == 64: the 60 x 80 layers of ResNet-34): the 512 x 64 tile needs
// no chunk swap at all -- nine taps on one staged patch.
static bool pp_f16_wide64_ok(const ConvArgsH& a) {
  using T = PP<MODE_F16, 2, true>;
  return !dbg(DBG_CONV_NO_PP) && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cout % 128 != 0 && a.Cout % 64 == 0 && a.Cin == 64 &&
         a.Ho == a.H && a.Wo == a.W && T::npc(a.W) <= 11 && T::lds_bytes(a.W) <= 159 * 1024;
}

bool conv_pp_f16_applicable(const ConvArgsH& a) {
  if (pp_f16_wide64_ok(a)) return true;
  return pp_shape_ok<MODE_F16, 2>(a.W, a.Cin, a.Cout, a.stride, a.pad, a.kh, a.kw) && a.Ho == a.H && a.Wo == a.W;
}

int launch_conv_pp_f16(const ConvArgsH& h, hipStream_t stream) {
  ConvArgs a{};
  a.x = reinterpret_cast<const float*>(h.x); a.w = reinterpret_cast<const float*>(h.w); a.bias = h.bias;
  a.residual = reinterpret_cast<const float*>(h.residual);
  a.pre_scale = reinterpret_cast<const float*>(h.pre_scale); a.pre_shift = reinterpret_cast<const float*>(h.pre_shift);
  a.y = reinterpret_cast<float*>(h.y);
  a.M = h.M; a.H = h.H; a.W = h.W; a.Cin = h.Cin; a.Ho = h.Ho; a.Wo = h.Wo; a.Cout = h.Cout; a.stride = 1; a.pad = 1;
  a.Kpad = h.Kpad; a.relu = h.relu; a.no_tail_split = h.no_tail_split;
  if (pp_f16_wide64_ok(h))
    return h.pre_scale ? launch_pp_variant<MODE_F16, true, 11, 2, true>(a, stream) : launch_pp_variant<MODE_F16, false, 11, 2, true>(a, stream);
  return h.pre_scale ? launch_pp_nt<MODE_F16, true, 2>(a, stream) : launch_pp_nt<MODE_F16, false, 2>(a, stream);
}

}  // namespace hp

#ifdef HP_PP_STAMPS
extern "C" __attribute__((visibility("default"))) int hp_debug_pp_stamps(double* out4) {  // cycles, 100-MHz ticks, taps, workgroups, prologue cycles, epilogue cycles
  unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(hp::g_pp_stamps), sizeof(h)) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hp::g_pp_stamps), z, sizeof(z));
  for (int i = 0; i < 7; ++i) out4[i] = (double)h[i];
  return 0;
}
#endif
