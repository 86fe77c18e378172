// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen/cuDNN(oneDNN) conv -> BatchNorm -> ReLU (-> +identity) sequences of the
// reference backbones (MP/models/torchvision_resnet.py:110-126,325-341;
// MP/models/wide_resnet.py:59-65,120-129) with ONE kernel per convolution:
//   y = act( conv(pre(x), W') + bias' [+ residual] )
// where eval-mode BatchNorm that FOLLOWS a conv is folded into (W', bias') on the host
// (net.cpp) and the BatchNorm+ReLU that PRECEDES the convs of a pre-activation block
// (BasicBlockV2.bn1, which cannot be folded -- SURVEY.md Appendix C) is applied as a
// per-channel scale/shift/ReLU prologue while the input tile is staged.
//
// GEMM view: C[M][N] = A[M][K] * B[K][N], M = n*Ho*Wo output pixels, N = Cout,
// K = KH*KW*Cin ordered (kh, kw, c).  Activations are NHWC so that a K-run of one filter
// tap is contiguous in HBM; weights are packed [Cout][Kpad] (K contiguous, zero padded to
// a multiple of 32).  A small host-built look-up table maps every 4-float K-chunk to its
// (tap offset, kh, kw, channel), which makes the 7x7/5x5 stems (Cin padded to a multiple of
// 4) and the 3x3/1x1 body convolutions one code path.
//
// Mapping to the hardware (wave64, 4 waves per workgroup, 2 workgroups per CU):
//   * block tile BM x BN = 128x128 (128x64 for the 64-channel layers), BK = 32; the 4 waves
//     sit 2x2, each owns a 64x64 (64x32) sub-tile = 2x2 (2x1) MFMA tiles of 32x32 -> 64 (32)
//     accumulator VGPRs;
//   * LDS tiles are [rows][32+4] floats, double buffered: the +4 pad makes both the 16-B
//     staging stores and the 16-B fragment reads (ds_read_b128: lane -> row = lane&31,
//     k = 4*(lane>>5)..+3) bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured).  One
//     ds_read_b128 feeds FOUR k-steps: the order of K inside a K-tile is permuted (k-step j
//     of a group multiplies k = j and k = 4+j), legal because A and B use the same order;
//   * the K loop is software pipelined INSIDE each wave so that the only thing a wave waits
//     for is the matrix pipe: after the barrier it reads the first fragments, issues the
//     (branch-free, predicated-by-select) global loads of tile t+1 and the LUT entry of tile
//     t+2, runs half of its MFMAs, then writes tile t+1 to the other LDS buffer (the loads
//     have had >= 2000 matrix-pipe cycles to land), runs the other half, and meets the one
//     barrier of the K-tile.  Fragment reads of group kg+1 are issued under the MFMAs of kg.
//     An f32 MFMA occupies the pipe for 64 cycles, so all ~200 non-MFMA instructions of a
//     K-tile fit in the shadows of its 64 MFMAs;
//   * zero padding and the pre-activation prologue are applied when the staged registers
//     are written to LDS (not when they are loaded), so no load is ever waited for early;
//   * workgroup ids are renumbered so that each XCD walks a contiguous range of tiles
//     (neighbouring tiles share activation rows / weight panels in that XCD's L2).
// The kernel is bound by the fp32 matrix pipe (157.3 TFLOP/s dense); DESIGN.md has the
// roofline and the measured fraction.
// Numerics: exact fp32 FMA chains (the f32 MFMA is bitwise an fmaf chain), K-order differs
// from the reference's oneDNN/cuDNN kernels, so results agree to fp32 round-off, not bitwise.
#include <atomic>
#include <cstdlib>

#include "conv.h"
#include "conv_epilogue.h"
#include "conv_splitk.h"

namespace hp {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row (floats)
constexpr int kThreads = 256;

// 4 waves arranged 2 (M) x 2 (N); wave tile = (BM/2) x (BN/2) = MT x NT MFMA tiles of 32x32.
template <int BM, int BN>
struct Tile {
  static constexpr int WAVES_N = 2;
  static constexpr int WM = BM / 2, WN = BN / 2;    // wave tile
  static constexpr int MT = WM / 32, NT = WN / 32;  // MFMA tiles per wave
  static constexpr int A_CHUNKS = BM * BK / 4 / kThreads;  // float4 per thread per K-tile
  static constexpr int B_CHUNKS = BN * BK / 4 / kThreads;
  static constexpr int LDS_FLOATS = 2 * (BM + BN) * LDK;
};

// registers of one staged K-tile
template <int NA, int NB>
struct Stage {
  floatx4 ra[NA], rb[NB];
  floatx4 ps, pb;     // prologue scale / shift of this chunk's 4 channels (PRE == 1)
  floatx4 se[NA];     // per-sample channel scale of each staged row (PRE == 2: squeeze-excitation)
  unsigned ok;        // bit i: A chunk i is inside the image (else it is zero padding)
};

// PRE: 0 = none, 1 = per-channel scale / shift / ReLU (pre-activation BN), 2 = per-(sample, channel)
// scale (the squeeze-excitation gate of an MBConv block, applied where the projection reads it)

template <int NA, int NB, int PRE>
__device__ __forceinline__ void issue_loads(const ConvArgs& a, int t, const int4 e, const int64_t (&rowoff)[NA],
                                            const int (&ih0)[NA], const int (&iw0)[NA], const int (&imgoff)[NA],
                                            const float* const (&wrow)[NB], Stage<NA, NB>& s) {
  // e = {offset, kh, kw, channel}; kh < 0 marks K padding.  Loads are unconditional: an
  // out-of-image chunk reads a harmless valid address and is zeroed at store time.
  unsigned ok = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int ih = ih0[i] + e.y, iw = iw0[i] + e.z;
    const bool in = (e.y >= 0) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
    const float* p = in ? a.x + rowoff[i] + e.x : a.x;
    s.ra[i] = *reinterpret_cast<const floatx4*>(p);
    ok |= (in ? 1u : 0u) << i;
    if (PRE == 2) s.se[i] = *reinterpret_cast<const floatx4*>(a.pre_scale + (in ? imgoff[i] + e.w : 0));
  }
  s.ok = ok;
#pragma unroll
  for (int i = 0; i < NB; ++i) s.rb[i] = *reinterpret_cast<const floatx4*>(wrow[i] + t * BK);
  if (PRE == 1) {
    const int c = e.y >= 0 ? e.w : 0;
    s.ps = *reinterpret_cast<const floatx4*>(a.pre_scale + c);
    s.pb = *reinterpret_cast<const floatx4*>(a.pre_shift + c);
  }
}

template <int NA, int NB, int PRE>
__device__ __forceinline__ void store_tile(float* Ast, float* Bst, const Stage<NA, NB>& s) {
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    floatx4 v = s.ra[i];
    if (PRE == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(v[q], s.ps[q], s.pb[q]), 0.f);
    }
    if (PRE == 2) v *= s.se[i];
    const bool in = (s.ok >> i) & 1u;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = in ? v[q] : 0.f;
    *reinterpret_cast<floatx4*>(Ast + 32 * i * LDK) = v;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<floatx4*>(Bst + 32 * i * LDK) = s.rb[i];
}

// ---- the same staging work, one chunk at a time (interleaved with the MFMAs) -----------
template <int NA, int NB, int PRE>
__device__ __forceinline__ void load_a_chunk(const ConvArgs& a, const int4 e, int i, int64_t rowoff, int ih0, int iw0,
                                             int imgoff, Stage<NA, NB>& s) {
  const int ih = ih0 + e.y, iw = iw0 + e.z;
  const bool in = (e.y >= 0) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
  const float* p = in ? a.x + rowoff + e.x : a.x;
  s.ra[i] = *reinterpret_cast<const floatx4*>(p);
  s.ok |= (in ? 1u : 0u) << i;
  if (PRE == 2) s.se[i] = *reinterpret_cast<const floatx4*>(a.pre_scale + (in ? imgoff + e.w : 0));
}

template <int NA, int NB, int PRE>
__device__ __forceinline__ void load_b_chunks(const ConvArgs& a, int t, const int4 e, const float* const (&wrow)[NB],
                                              Stage<NA, NB>& s) {
#pragma unroll
  for (int i = 0; i < NB; ++i) s.rb[i] = *reinterpret_cast<const floatx4*>(wrow[i] + t * BK);
  if (PRE == 1) {
    const int c = e.y >= 0 ? e.w : 0;
    s.ps = *reinterpret_cast<const floatx4*>(a.pre_scale + c);
    s.pb = *reinterpret_cast<const floatx4*>(a.pre_shift + c);
  }
}

template <int NA, int NB, int PRE>
__device__ __forceinline__ void store_a_chunk(float* Aw, int i, const Stage<NA, NB>& s) {
  floatx4 v = s.ra[i];
  if (PRE == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(v[q], s.ps[q], s.pb[q]), 0.f);
  }
  if (PRE == 2) v *= s.se[i];
  const bool in = (s.ok >> i) & 1u;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = in ? v[q] : 0.f;
  *reinterpret_cast<floatx4*>(Aw + 32 * i * LDK) = v;
}

template <int NA, int NB>
__device__ __forceinline__ void store_b_chunk(float* Bw, int i, const Stage<NA, NB>& s) {
  *reinterpret_cast<floatx4*>(Bw + 32 * i * LDK) = s.rb[i];
}

template <int MT, int NT>
__device__ __forceinline__ void read_frags(const float* Ab, const float* Bb, int kg, floatx4 (&fa)[MT], floatx4 (&fb)[NT]) {
#pragma unroll
  for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + kg * 8);
#pragma unroll
  for (int i = 0; i < NT; ++i) fb[i] = *reinterpret_cast<const floatx4*>(Bb + i * 32 * LDK + kg * 8);
}

template <int MT, int NT>
__device__ __forceinline__ void mfma_group(const floatx4 (&fa)[MT], const floatx4 (&fb)[NT], floatx16 (&acc)[MT][NT]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][j], fb[ni][j], acc[mi][ni], 0, 0, 0);
}

template <int BM, int BN, int PRE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_igemm_f32(ConvArgs a) {
  using TT = Tile<BM, BN>;
  constexpr int MT = TT::MT, NT = TT::NT, NA = TT::A_CHUNKS, NB = TT::B_CHUNKS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                       // [2][BM][LDK]
  float* Bs = lds + 2 * BM * LDK;        // [2][BN][LDK]

  // XCD-aware work-item order; the tiles of the last partial round are split along K (conv_splitk.h:
  // the stride-2 3x3 layers have 320 / 600 / 1200 tiles for 512 slots)
  int lin, slice;
  bool split;
  if (!splitk_decode(a, lin, slice, split)) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kc = tid & 7;         // which float4 of the 32-float K-tile row
  const int r0 = tid >> 3;        // first staged row (0..31); rows r0 + 32*i

  // ---- per-row (output pixel) constants for the A gather ----
  int64_t rowoff[NA];
  int ih0[NA], iw0[NA], imgoff[NA];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = m0 + r0 + 32 * i;
    if (m < a.M) {
      const int img = fdiv((int)m, a.fd_howo);
      const int rem = (int)m - img * HoWo;
      const int oh = fdiv(rem, a.fd_wo), ow = rem - oh * a.Wo;
      ih0[i] = oh * a.stride - a.pad;
      iw0[i] = ow * a.stride - a.pad;
      rowoff[i] = (((int64_t)img * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
      imgoff[i] = img * a.Cin;
    } else {
      ih0[i] = -(1 << 28); iw0[i] = 0; rowoff[i] = 0; imgoff[i] = 0;
    }
  }
  const float* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wrow[i] = a.w + (int64_t)(n0 + r0 + 32 * i) * a.Kpad + 4 * kc;

  float* const Ast = As + r0 * LDK + 4 * kc;  // this thread's staging slot (row r0, chunk kc)
  float* const Bst = Bs + r0 * LDK + 4 * kc;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = (wave / TT::WAVES_N) * TT::WM;
  const int wn = (wave % TT::WAVES_N) * TT::WN;
  const int frow = lane & 31, fk = 4 * (lane >> 5);
  const float* const Afr = As + (wm + frow) * LDK + fk;  // this lane's fragment base, buffer 0
  const float* const Bfr = Bs + (wn + frow) * LDK + fk;

  Stage<NA, NB> st;
  // K-tile range of this work item
  const int t_begin = split ? slice * a.ktiles / a.sk_S : 0;
  const int t_end = split ? (slice + 1) * a.ktiles / a.sk_S : a.ktiles;
  // the LUT carries one extra K-tile of padding entries, so t+2 below never reads out of range
  issue_loads<NA, NB, PRE>(a, t_begin, a.lut[t_begin * 8 + kc], rowoff, ih0, iw0, imgoff, wrow, st);
  int4 e_next = a.lut[(t_begin + 1) * 8 + kc];
  store_tile<NA, NB, PRE>(Ast, Bst, st);
  __syncthreads();

  // One K-tile = 16 "quads" (k-group kg = q/4, k-step j = q%4; a quad is the MT*NT independent
  // MFMAs of one k-step).  After every quad a small, fixed chunk of the staging work for tile
  // t+1 is issued, so it executes in the shadow of the quad's 64-cycle MFMAs; sched_barrier
  // pins this order (left alone, hipcc clusters loads / MFMAs / stores into three phases and
  // the matrix pipe idles through two of them):
  //   q0-3   address + global load of A chunk 0..3   (q1: LDS fragments of kg1)
  //   q4     global loads of the B chunks, prologue scale/shift, LUT entry of tile t+2
  //   q5     LDS fragments of kg2
  //   q8-11  zero-pad/prologue + LDS store of A chunk 0..3 (q9: LDS fragments of kg3)
  //   q12-13 LDS store of the B chunks
  floatx4 fa[2][MT], fb[2][NT];
  const int last = t_end - 1;
  for (int t = t_begin; t <= last; ++t) {
    const int buf = (t - t_begin) & 1;
    const bool stage = t < last;  // wave-uniform
    const float* Ab = Afr + buf * BM * LDK;
    const float* Bb = Bfr + buf * BN * LDK;
    float* const Aw = Ast + (buf ^ 1) * BM * LDK;
    float* const Bw = Bst + (buf ^ 1) * BN * LDK;
    const int4 e = e_next;
    read_frags<MT, NT>(Ab, Bb, 0, fa[0], fb[0]);
    if (stage) st.ok = 0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int kg = q >> 2, j = q & 3;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kg & 1][mi][j], fb[kg & 1][ni][j], acc[mi][ni], 0, 0, 0);
      // HP_ABL_* are compile-time ablation switches for tools/conv_ablate.sh (which part of the
      // K loop costs matrix-pipe time); never defined in the product build.
#ifndef HP_ABL_NO_DSREAD
      if (q == 1) read_frags<MT, NT>(Ab, Bb, 1, fa[1], fb[1]);
      if (q == 5) read_frags<MT, NT>(Ab, Bb, 2, fa[0], fb[0]);
      if (q == 9) read_frags<MT, NT>(Ab, Bb, 3, fa[1], fb[1]);
#endif
#ifdef HP_ABL_NO_STAGE
      if (false) {
#else
      if (stage) {
#endif
        if (q < 4) {
#pragma unroll
          for (int i = q; i < NA; i += 4) load_a_chunk<NA, NB, PRE>(a, e, i, rowoff[i], ih0[i], iw0[i], imgoff[i], st);
        } else if (q == 4) {
          load_b_chunks<NA, NB, PRE>(a, t + 1, e, wrow, st);
          e_next = a.lut[(t + 2) * 8 + kc];
        } else if (q >= 8 && q < 12) {
#pragma unroll
          for (int i = q - 8; i < NA; i += 4) store_a_chunk<NA, NB, PRE>(Aw, i, st);
        } else if (q == 12) {
#pragma unroll
          for (int i = 0; i < NB / 2; ++i) store_b_chunk<NA, NB>(Bw, i, st);
        } else if (q == 13) {
#pragma unroll
          for (int i = NB / 2; i < NB; ++i) store_b_chunk<NA, NB>(Bw, i, st);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef HP_ABL_NO_BARRIER
    __syncthreads();
#endif
  }

  if (split && !splitk_reduce<BM, BN, MT, NT, kThreads>(a, acc, lin - a.sk_regular, slice)) return;

  // ---- epilogue: bias, residual, ReLU through an LDS transpose (conv_epilogue.h) ----
  conv_epilogue<BM, BN, MT, NT, kThreads>(a, lds, acc, m0, n0, wm, wn);
}

template <int BM, int BN, int PRE>
static int launch_variant(ConvArgs args, hipStream_t stream) {
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = (args.Cout + BN - 1) / BN;  // weights / bias are padded to whole tiles; stores are not
  if (args.M >= (1ll << 31)) return fail(HP_ERR_ARG, "conv: more than 2^31 output pixels");
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int nblk = args.tiles_m * args.tiles_n;
  const size_t lds = Tile<BM, BN>::LDS_FLOATS * sizeof(float);
  int rc = conv_plan_split(args, nblk, lds, args.ktiles, 1, stream);
  if (rc) return rc;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8;
  hipLaunchKernelGGL((conv_igemm_f32<BM, BN, PRE>), dim3(8 * per_xcd), dim3(kThreads), lds, stream, args);
  return check_launch("conv_igemm_f32");
}

int launch_conv(const ConvArgs& a, int variant, hipStream_t stream) {
  const int pre = a.pre_scale == nullptr ? 0 : (a.pre_shift != nullptr ? 1 : 2);
  if (variant == 0) {
    if (pre == 2) return launch_variant<128, 128, 2>(a, stream);
    return pre ? launch_variant<128, 128, 1>(a, stream) : launch_variant<128, 128, 0>(a, stream);
  }
  if (pre == 2) return launch_variant<128, 64, 2>(a, stream);
  return pre ? launch_variant<128, 64, 1>(a, stream) : launch_variant<128, 64, 0>(a, stream);
}

template <int BM, int BN, int PRE>
static int opt_in_lds() {
  // > 64 KB of dynamic LDS needs the opt-in attribute
  HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, PRE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(Tile<BM, BN>::LDS_FLOATS * sizeof(float))));
  return HP_OK;
}

int conv_setup_once() {
  static FirstLaunch fl;
  return fl.once([](FirstLaunch&) {
    int rc;
    if ((rc = opt_in_lds<128, 128, 0>())) return rc;
    if ((rc = opt_in_lds<128, 128, 1>())) return rc;
    if ((rc = opt_in_lds<128, 128, 2>())) return rc;
    if ((rc = opt_in_lds<128, 64, 0>())) return rc;
    if ((rc = opt_in_lds<128, 64, 1>())) return rc;
    if ((rc = opt_in_lds<128, 64, 2>())) return rc;
    return HP_OK;
  });
}

}  // namespace hp

// diagnostics: resident workgroups per CU the runtime grants a conv variant
// ---- kernel family of the single-layer entry points (hp_conv_select_algo; networks: hp_net_set_conv_algo) ----------
namespace hp {
namespace {
std::atomic<int> g_layer_algo{HP_CONV_ALGO_AUTO};
}
int conv_layer_algo() { return g_layer_algo.load(std::memory_order_relaxed); }
__global__ void zero_words_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}
int launch_zero_words(unsigned* p, int n, hipStream_t stream) {
  if (n <= 0) return HP_OK;
  hipLaunchKernelGGL(zero_words_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n);
  return check_launch("zero_words_kernel");
}

}  // namespace hp

namespace hp {
static std::atomic<long long> g_scratch_launches{0};
bool note_kernel(const void* fn) {
  hipFuncAttributes at{};
  return hipFuncGetAttributes(&at, fn) == hipSuccess && at.localSizeBytes > 0;
}
void count_scratch_launch() { g_scratch_launches.fetch_add(1); }
}  // namespace hp

extern "C" long long hp_scratch_launches(void) { return hp::g_scratch_launches.load(); }

extern "C" int hp_conv_select_algo(int algo) {
  HP_REQUIRE(algo >= HP_CONV_ALGO_AUTO && algo <= HP_CONV_ALGO_SPLIT, "hp_conv_select_algo: unknown algorithm");
  hp::g_layer_algo.store(algo, std::memory_order_relaxed);
  return HP_OK;
}

extern "C" int hp_conv_occupancy(int variant) {
  using namespace hp;
  if (conv_setup_once() != HP_OK) return HP_ERR_HIP;
  int nb = 0;
  if (variant == 0) {
    HP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_igemm_f32<128, 128, 0>, kThreads,
                                                              Tile<128, 128>::LDS_FLOATS * sizeof(float)));
  } else {
    HP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_igemm_f32<128, 64, 0>, kThreads,
                                                              Tile<128, 64>::LDS_FLOATS * sizeof(float)));
  }
  return nb;
}
