// 3x3 / stride-1 / pad-1 convolution (85 % of the backbone FLOPs) with the input staged ONCE
// per 32-channel chunk: the "patch" variant of the implicit GEMM in conv.hip.
//
// The generic kernel gathers an im2col A-tile per filter tap, i.e. it pulls every input pixel
// through L2 -> registers -> LDS nine times; its measured limiter is exactly that vector-memory
// path.  For a stride-1 3x3 filter the nine A-tiles of a 128-pixel output tile
// are the same pixels shifted: with pixels numbered linearly g = (img*H + oh)*W + ow, tap
// (kh, kw) of output pixel g reads input pixel g + (kh-1)*W + (kw-1).  So the block stages the
// pixel range [m0 - W - 1, m0 + 127 + W + 1] (P = 128 + 2W + 2 rows x 32 channels, 36-float
// padded rows) once per channel chunk, and the A fragment of tap t is the same ds_read_b128
// at a row offset.  Image borders (where the linear shift would wrap into the neighbouring
// row / image) are handled by zeroing the fragment registers with a per-lane 9-bit validity
// mask (4 v_cndmask per fragment read).  Weights stream as before: one [BN][32] tile per tap,
// double buffered.  Per tap the block now loads ~P*128/9 + BN*128 bytes instead of
// (128 + BN)*128: -36 % at BN = 128, -48 % at BN = 64, and the A-side address math, selects
// and LDS stores shrink by the same factor.
//
// Everything else (MFMA tiling, permuted K order, per-quad software pipelining, epilogue,
// XCD-aware tile order, numerics) is as documented in conv.hip.
#include <cmath>
#include <cstdlib>

#include <map>
#include <mutex>
#include <utility>

#include "conv.h"
#include "conv_epilogue.h"
#include "conv_splitk.h"

namespace hp {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int kThreads = 256;
constexpr int BM = 128;
constexpr int kMaxPatchChunks = 12;  // float4 per thread per channel chunk: P <= 384 rows (W <= 127)

__device__ __forceinline__ floatx4 masked(floatx4 v, bool keep) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = keep ? v[q] : 0.f;
  return v;
}

template <int BN, bool PRE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_patch_f32(ConvArgs a, int P, int npc) {
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int NB = BN * BK / 4 / kThreads;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* patch = lds;                 // [P][LDK]
  float* Bs = lds + P * LDK;          // [2][BN][LDK]

  // work item = a regular tile (whole K) or a (tile, slice) of the tail round (conv_splitk.h):
  // tile quantisation cost 20-40 % on the 15x20 / 8x10 layers
  int tile, slice;
  bool split;
  if (!splitk_decode(a, tile, slice, split)) return;
  const int tile_m = fdiv(tile, a.fd_tn), tile_n = tile - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kc = tid & 7;
  const int r0 = tid >> 3;
  const int W = a.W, H = a.H, Cin = a.Cin;
  const int ncc_all = Cin / BK;
  const int cc_begin = split ? slice * ncc_all / a.sk_S : 0;
  const int ncc = split ? (slice + 1) * ncc_all / a.sk_S : ncc_all;  // end of this item's chunk range
  const int tt0 = cc_begin * 9, ntiles = ncc * 9;

  // ---- patch rows owned by this thread: row = r0 + 32 j, global pixel gp = m0 - (W+1) + row
  const int64_t gp0 = m0 - (W + 1) + r0;
  const float* const xk = a.x + 4 * kc;

  // ---- B staging
  const float* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wrow[i] = a.w + (int64_t)(n0 + r0 + 32 * i) * a.Kpad + 4 * kc;
  float* const Bst = Bs + r0 * LDK + 4 * kc;
  float* const Pst = patch + r0 * LDK + 4 * kc;

  // ---- fragment bases + per-row validity of the 9 taps
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int frow = lane & 31, fk = 4 * (lane >> 5);
  const float* const Afr = patch + (wm + frow + W + 1) * LDK + fk;
  const float* const Bfr = Bs + (wn + frow) * LDK + fk;
  unsigned vmask[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
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

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  floatx4 pr[kMaxPatchChunks];  // staged patch chunk (next channel chunk)
  floatx4 rb[NB];
  floatx4 ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};

  auto patch_row_ok = [&](int j) -> bool {
    const int64_t gp = gp0 + 32 * j;
    return (r0 + 32 * j < P) & (gp >= 0) & (gp < a.M);
  };
  auto load_patch_chunk = [&](int j, int cc) {
    const bool ok = patch_row_ok(j);
    const float* p = ok ? xk + (gp0 + 32 * j) * Cin + cc * BK : a.x;
    pr[j] = *reinterpret_cast<const floatx4*>(p);
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int j = 0; j < kMaxPatchChunks; ++j) {
      if (j < npc && r0 + 32 * j < P) {
        floatx4 v = pr[j];
        if (PRE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(v[q], ps[q], pb[q]), 0.f);
        }
        *reinterpret_cast<floatx4*>(Pst + 32 * j * LDK) = masked(v, patch_row_ok(j));
      }
    }
  };

  // ---- prologue: patch of chunk 0, weights of tile 0
#pragma unroll
  for (int j = 0; j < kMaxPatchChunks; ++j)
    if (j < npc) load_patch_chunk(j, cc_begin);
  if (PRE) {
    ps = *reinterpret_cast<const floatx4*>(a.pre_scale + cc_begin * BK + 4 * kc);
    pb = *reinterpret_cast<const floatx4*>(a.pre_shift + cc_begin * BK + 4 * kc);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const floatx4*>(wrow[i] + cc_begin * BK);
  store_patch();
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<floatx4*>(Bst + 32 * i * LDK) = rb[i];
  __syncthreads();

  floatx4 fa[2][MT], fb[2][NT];
  auto read_frags = [&](int set, const float* Ab, const float* Bb, int kg, int tap) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      fa[set][i] = masked(*reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + kg * 8), (vmask[i] >> tap) & 1u);
#pragma unroll
    for (int i = 0; i < NT; ++i) fb[set][i] = *reinterpret_cast<const floatx4*>(Bb + i * 32 * LDK + kg * 8);
  };

  int cc = cc_begin, tap = 0;
  for (int tt = tt0; tt < ntiles; ++tt) {
    const int buf = (tt - tt0) & 1;
    const bool more = tt + 1 < ntiles;          // wave-uniform
    const int ntap = tap == 8 ? 0 : tap + 1, ncc_ = tap == 8 ? cc + 1 : cc;
    const bool stage_patch = more && (cc + 1 < ncc);  // next chunk's patch is loaded during taps 0..5
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
    const float* Ab = Afr + d * LDK;
    const float* Bb = Bfr + buf * BN * LDK;
    float* const Bw = Bst + (buf ^ 1) * BN * LDK;
    const float* const wnext = nullptr;
    (void)wnext;
    read_frags(0, Ab, Bb, 0, tap);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int kg = q >> 2, j = q & 3;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kg & 1][mi][j], fb[kg & 1][ni][j], acc[mi][ni], 0, 0, 0);
      if (q == 1) read_frags(1, Ab, Bb, 1, tap);
      if (q == 5) read_frags(0, Ab, Bb, 2, tap);
      if (q == 9) read_frags(1, Ab, Bb, 3, tap);
      if (more) {
        if (q < 4) {  // weights of the next tile: one chunk per quad
#pragma unroll
          for (int i = q; i < NB; i += 4)
            rb[i] = *reinterpret_cast<const floatx4*>(wrow[i] + (ntap * Cin + ncc_ * BK));
        } else if (q < 8) {  // patch of the next channel chunk: 2 chunks per tap (taps 0..5)
          if (stage_patch && tap < 6) {
            const int jj = tap * 2 + (q - 4);
            if (q < 6 && jj < npc) {
#pragma unroll
              for (int j2 = 0; j2 < kMaxPatchChunks; ++j2)
                if (j2 == jj) load_patch_chunk(j2, cc + 1);
            }
          }
        } else if (q < 12) {
#pragma unroll
          for (int i = q - 8; i < NB; i += 4) *reinterpret_cast<floatx4*>(Bw + 32 * i * LDK) = rb[i];
        }
      }
      // (a finer MFMA/VALU interleave inside the quad via sched_group_barrier measured 3-5 %
      // slower than leaving the quad's order to hipcc)
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (tap == 8 && more) {  // every wave is done with this chunk's patch: swap in the next one
      if (PRE) {
        ps = *reinterpret_cast<const floatx4*>(a.pre_scale + (cc + 1) * BK + 4 * kc);
        pb = *reinterpret_cast<const floatx4*>(a.pre_shift + (cc + 1) * BK + 4 * kc);
      }
      store_patch();
      __syncthreads();
    }
    tap = ntap; cc = ncc_;
  }

  // ---- split tiles: park the partial sums, the last slice to arrive adds the others
  if (split && !splitk_reduce<BM, BN, MT, NT, kThreads>(a, acc, tile - a.sk_regular, slice)) return;

  // ---- epilogue: bias, residual, ReLU through an LDS transpose (conv_epilogue.h) ----
  conv_epilogue<BM, BN, MT, NT, kThreads>(a, lds, acc, m0, n0, wm, wn);
}

// Decide how the tiles of the last partial round are split (see the kernel), provide the
// slab / counter workspace (grown on demand, reused by every launch; launches that split are
// therefore expected on ONE stream per process, which is how net.cpp issues them).  Counters are
// zeroed when allocated and re-armed by the reducing block.
struct SplitWorkspace { float* slabs = nullptr; size_t slab_bytes = 0; int* counters = nullptr; size_t counter_bytes = 0; int slots = 0; };

double rounds_cost(double r) {  // time of r rounds' worth of equal items; a partial round runs faster
  const double full = std::floor(r), frac = r - full;
  return full + (frac > 0 ? 0.55 + 0.45 * frac : 0.0);
}

}  // namespace

int conv_plan_split(ConvArgs& a, int T, size_t lds_bytes, int k_units, int ktiles_per_unit, hipStream_t stream) {
  // one workspace per (device, stream), guarded: see plan_tail_split in conv_split.hip
  static std::map<std::pair<int, hipStream_t>, SplitWorkspace> wss;
  static std::mutex wss_mutex;
  std::lock_guard<std::mutex> lock(wss_mutex);
  int dev = 0;
  HP_CHECK_HIP(hipGetDevice(&dev));
  SplitWorkspace& ws = wss[std::make_pair(dev, stream)];
  if (ws.slots == 0) {
    int cus = 256;
    HP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    ws.slots = 2 * cus;  // two workgroups per CU
  }
  const int slots = lds_bytes * 2 <= 160 * 1024 ? ws.slots : ws.slots / 2;
  const int ncc = k_units;
  const bool no_split = dbg(DBG_CONV_NO_SPLITK) != 0;
  int regular = (T / slots) * slots, tail = T - regular, S = 1;
  regular -= regular % 8;  // the kernel deals regular tiles to the 8 XCDs evenly
  tail = T - regular;
  // while two lanes share the GPU (no_tail_split) a launch is still sliced when it and its twin on the other lane together
  // cannot fill the GPU: it then plans against half of the slots (the rule of plan_tail_split in conv_split.hip; EfficientNet's
  // projections on the 7 x 10 maps are 70 - 105 workgroups x 44 - 72 K-tiles per lane)
  const bool shared = a.no_tail_split != 0;
  const int fill = shared ? slots / 2 : slots;
  if (tail > 0 && ncc > 1 && !no_split && (!shared || (regular == 0 && tail <= fill))) {
    double best = rounds_cost((double)tail / fill);
    // a slice must stay long (>= 12 K-tiles of 32): parking and re-reading a 64-KB slab costs about
    // as much as 2-3 K-tiles, so splitting short tiles loses (measured on the 64->128 stride-2 layer)
    for (int s = 2; s <= ncc && s <= 16 && (ncc * ktiles_per_unit) / s >= 12; ++s) {
      const double c = rounds_cost((double)tail * s / fill) / s + 0.015 * s;  // + slab traffic / item start-up
      if (c < 0.9 * best) { best = c; S = s; }
    }
  }
  a.sk_regular = regular;
  a.sk_S = S;
  a.sk_tail_items = tail * S;
  a.sk_slabs = nullptr;
  a.sk_counters = nullptr;
  if (S > 1) {
    const size_t need_slab = (size_t)tail * S * BM * 128 * sizeof(float), need_cnt = (size_t)tail * sizeof(int);
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

namespace {

template <int BN, bool PRE>
int launch(ConvArgs args, hipStream_t stream) {
  const int P = BM + 2 * args.W + 2;
  const int npc = (P * 8 + kThreads - 1) / kThreads;
  size_t lds = ((size_t)P * LDK + 2 * BN * LDK) * sizeof(float);
  const size_t lds_epi = (size_t)epilogue_lds_floats<BM, BN>() * sizeof(float);
  if (lds < lds_epi) lds = lds_epi;
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch&) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_f32<BN, PRE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));  // + static ticket word
        return HP_OK;
      }))
    return rc0;
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = args.Cout / BN;
  if (args.M >= (1ll << 31)) return fail(HP_ERR_ARG, "conv: more than 2^31 output pixels");
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int T = args.tiles_m * args.tiles_n;
  int rc = conv_plan_split(args, T, lds, args.Cin / BK, 9, stream);
  if (rc) return rc;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8;
  hipLaunchKernelGGL((conv3x3_patch_f32<BN, PRE>), dim3(8 * per_xcd), dim3(kThreads), lds, stream, args, P, npc);
  return check_launch("conv3x3_patch_f32");
}

}  // namespace

bool conv_patch_applicable(const ConvArgs& a, int kh, int kw) {
  if (kh != 3 || kw != 3 || a.stride != 1 || a.pad != 1 || a.Cin % BK != 0 || a.Cout % 64 != 0) return false;
  const int P = BM + 2 * a.W + 2;
  if ((P * 8 + kThreads - 1) / kThreads > kMaxPatchChunks) return false;
  // patch + double-buffered 128-wide weight tile must fit the 160 KB of LDS
  return ((size_t)P * LDK + 2 * 128 * LDK) * sizeof(float) <= 150 * 1024;
}

int launch_conv_patch(const ConvArgs& a, int variant, hipStream_t stream) {
  const bool pre = a.pre_scale != nullptr;
  if (variant == 0) return pre ? launch<128, true>(a, stream) : launch<128, false>(a, stream);
  return pre ? launch<64, true>(a, stream) : launch<64, false>(a, stream);
}

}  // namespace hp
