// fp16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16, fp32
// accumulate): the conv stack of configuration C5 (MegaPose coarse scoring, 576
// SO(3)-grid views per object, "fp16 MFMA conv").  The reference has no fp16 path; this one keeps
// its arithmetic contract as close as the format allows: weights and activations are rounded to
// fp16 once (BN folded in fp32 first), every dot product is accumulated in fp32, bias / residual /
// ReLU are applied on the fp32 accumulator and the result is rounded to fp16 once per layer.
//
// Same GEMM view and byte geometry as conv.hip (M = n*Ho*Wo pixels, N = Cout, K = (kh, kw, c)
// with NHWC activations), with 16-byte chunks now holding 8 halves:
//   * block tile 128 x 128 (128 x 64 when Cout = 64), BK = 64 halves = one 128-B row per pixel
//     and tap run; 4 waves 2x2, wave tile 64x64 (64x32) = 2x2 (2x1) MFMA tiles of 32x32; 2
//     workgroups per CU;
//   * LDS tiles [rows][64+8] halves: 144-B rows = the 36-dword rows of the fp32 kernel, so both
//     the ds_write_b128 staging and the ds_read_b128 fragment reads (lane -> row = lane & 31,
//     k = 8 (lane >> 5) .. +7, exactly the A/B operand of one 32x32x16 MFMA) stay conflict free;
//   * a K-tile is only 16 MFMAs x 32 cycles per wave (8x shorter than in fp32), far less than
//     the global-load latency, so staging runs two K-tiles ahead in two alternating register
//     sets: tile t+1 is written to the other LDS buffer right after the barrier, the loads of
//     tile t+3 are re-issued into the registers that just became free, one barrier per K-tile;
//   * loads go through buffer descriptors: per-lane byte offsets are 32-bit, out-of-image taps
//     and K padding use an out-of-range offset and read as zero (no select, no 64-bit address
//     arithmetic); weights use a constant per-lane offset plus a wave-uniform tile offset;
//   * the pre-activation BN+ReLU prologue (WideResNet) is applied in packed fp16 when the staged
//     chunk is written to LDS;
//   * epilogue through an LDS transpose: fp32 accumulators -> bias (fp32), residual (fp16), ReLU
//     -> 8 halves per 16-B store;
//   * XCD-aware tile order as in conv.hip.
// The kernel is bound by operand delivery, not by the matrix pipe (a 128x128x64 fp16 tile needs
// 32 KB of operands per 512 matrix-pipe cycles); DESIGN.md has the measured rates.
#include <cstdlib>
#include <type_traits>

#include "conv.h"

namespace hp {

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128;
constexpr int BKH = 64;        // halves per K-tile
constexpr int LDH = BKH + 8;   // padded LDS row (halves)
constexpr unsigned kOob = 0xFFFFFFF0u;  // buffer offset that is always out of range -> zeros

template <int BN>
constexpr size_t lds_bytes_f16() {
  const size_t loop = (size_t)2 * (BM + BN) * LDH * 2;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  return loop > epi ? loop : epi;
}

__device__ __forceinline__ halfx8 load8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(halfx8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int BN, bool PRE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_igemm_f16(ConvArgsH a) {
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int NA = BM * BKH / 8 / kThreads;  // 16-B chunks per thread per K-tile: 4
  constexpr int NB = BN * BKH / 8 / kThreads;  // 4 or 2
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const As = reinterpret_cast<_Float16*>(lds_raw);  // [2][BM][LDH]
  _Float16* const Bs = As + 2 * BM * LDH;                     // [2][BN][LDH]

  const int nblk = a.tiles_m * a.tiles_n;
  const int per_xcd = (nblk + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= nblk) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kc = tid & 7, r0 = tid >> 3;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w), 0, (int)a.w_bytes, 0x00020000);

  // ---- per-row constants of the A gather (byte offsets are 32-bit: x_bytes < 4 GB)
  unsigned rowoff[NA];
  int ih0[NA], iw0[NA];
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
      rowoff[i] = (unsigned)(((((int64_t)img * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin) * 2);
    } else {
      ih0[i] = -(1 << 28); iw0[i] = 0; rowoff[i] = 0;
    }
  }
  unsigned wvoff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(n0 + r0 + 32 * i) * a.Kpad + 8 * kc) * 2);

  _Float16* const Ast = As + r0 * LDH + 8 * kc;
  _Float16* const Bst = Bs + r0 * LDH + 8 * kc;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  const _Float16* const Afr = As + (wm + frow) * LDH + fk;
  const _Float16* const Bfr = Bs + (wn + frow) * LDH + fk;

  // ---- staged K-tiles: two register sets
  struct Stage {
    halfx8 ra[NA], rb[NB];
    halfx8 ps, pb;   // prologue scale / shift of this chunk's 8 channels (PRE)
    unsigned ok;     // PRE only: which A chunks are inside the image
  };
  Stage st[2];
  const int last = a.ktiles - 1;
  auto issue = [&](Stage& s, int t) {
    const int tt = t <= last ? t : last;  // past the end: re-read the last tile (never stored)
    const int4 e = a.lut[tt * 8 + kc];    // {offset (halves), kh, kw, channel}; kh < 0 = K padding
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int ih = ih0[i] + e.y, iw = iw0[i] + e.z;
      const bool in = (e.y >= 0) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
      s.ra[i] = load8(xrsrc, in ? rowoff[i] + (unsigned)(e.x * 2) : kOob, 0);
      ok |= (in ? 1u : 0u) << i;
    }
    s.ok = ok;
#pragma unroll
    for (int i = 0; i < NB; ++i) s.rb[i] = load8(wrsrc, wvoff[i], (unsigned)(tt * BKH * 2));
    if (PRE) {
      const int c = e.y >= 0 ? e.w : 0;
      s.ps = *reinterpret_cast<const halfx8*>(a.pre_scale + c);
      s.pb = *reinterpret_cast<const halfx8*>(a.pre_shift + c);
    }
  };
  auto store = [&](const Stage& s, int buf) {
    _Float16* Aw = Ast + buf * BM * LDH;
    _Float16* Bw = Bst + buf * BN * LDH;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      halfx8 v = s.ra[i];
      if (PRE) {
        v = v * s.ps + s.pb;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = v[q] > (_Float16)0 ? v[q] : (_Float16)0;
        if (!((s.ok >> i) & 1u)) v = halfx8{0, 0, 0, 0, 0, 0, 0, 0};
      }
      *reinterpret_cast<halfx8*>(Aw + 32 * i * LDH) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<halfx8*>(Bw + 32 * i * LDH) = s.rb[i];
  };

  issue(st[0], 0);
  issue(st[1], 1);
  store(st[0], 0);
  issue(st[0], 2);
  __syncthreads();

  // tile t is computed from LDS buffer t & 1 while tile t+1 (register set (t+1) & 1) is written to
  // the other buffer and tile t+3 is requested into the same registers
  auto tile = [&](int t, auto par) {
    constexpr int P = decltype(par)::value;  // t & 1
    const _Float16* Ab = Afr + P * BM * LDH;
    const _Float16* Bb = Bfr + P * BN * LDH;
    halfx8 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[0][i] = *reinterpret_cast<const halfx8*>(Ab + i * 32 * LDH);
#pragma unroll
    for (int i = 0; i < NT; ++i) fb[0][i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[(ks + 1) & 1][i] = *reinterpret_cast<const halfx8*>(Ab + i * 32 * LDH + (ks + 1) * 16);
#pragma unroll
        for (int i = 0; i < NT; ++i) fb[(ks + 1) & 1][i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH + (ks + 1) * 16);
      }
      if (ks == 0 && t < last) store(st[1 - P], 1 - P);
      if (ks == 1 && t < last) issue(st[1 - P], t + 3);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][mi], fb[ks & 1][ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int t = 0; t <= last; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    if (t + 1 <= last) tile(t + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue: accumulators -> LDS [row][BN+4] floats -> bias, residual, ReLU -> 8 halves / store
  float* const cl = reinterpret_cast<float*>(lds_raw);
  constexpr int LDC = BN + 4;
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
  constexpr int ITERS = BM * C8 / kThreads;
  const int c8 = tid % C8;
  const int n = n0 + 8 * c8;
  floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (a.bias) {
    b0 = *reinterpret_cast<const floatx4*>(a.bias + n);
    b1 = *reinterpret_cast<const floatx4*>(a.bias + n + 4);
  }
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const int row = tid / C8 + k * (kThreads / C8);
    const int64_t m = m0 + row;
    if (m < a.M) {
      floatx4 v0 = *reinterpret_cast<const floatx4*>(cl + row * LDC + 8 * c8) + b0;
      floatx4 v1 = *reinterpret_cast<const floatx4*>(cl + row * LDC + 8 * c8 + 4) + b1;
      if (a.residual) {
        const halfx8 rr = *reinterpret_cast<const halfx8*>(a.residual + m * a.Cout + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] += (float)rr[q]; v1[q] += (float)rr[4 + q]; }
      }
      if (a.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
      }
      halfx8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q] = (_Float16)v0[q]; o[4 + q] = (_Float16)v1[q]; }
      *reinterpret_cast<halfx8*>(a.y + m * a.Cout + n) = o;
    }
  }
}

// ---- 3x3 / stride-1 / pad-1 layers: input staged ONCE per 64-channel chunk ("patch" variant, as
// conv_patch.hip does for fp32).  The generic kernel above is bound by operand delivery (32 KB per
// 512 matrix-pipe cycles at 128x128); for a stride-1 3x3 filter the nine A tiles of a 128-pixel
// output tile are the same pixels shifted, so the block stages pixels [m0 - W - 1, m0 + 127 + W + 1]
// once per channel chunk and reads the A fragment of tap (kh, kw) at a row offset, zeroing rows whose
// shifted pixel wraps around an image border with a per-lane 9-bit mask.  Operand traffic per tap
// falls from (128 + BN) x 128 B to ~P/9 x 128 B + BN x 128 B: -37 % at BN = 128, -55 % at BN = 64.

// NPC = 16-B chunks per thread per staged patch (compile time: they live in registers for a whole
// channel chunk): ceil(P * 8 / 256) -> 5 / 6 / 7 / 10 for W = 10 / 20 / 40 / 80
template <int BN, bool PRE, int NPC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_patch_f16(ConvArgsH a, int P) {
  constexpr int npc = NPC;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int NB = BN * BKH / 8 / kThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const patch = reinterpret_cast<_Float16*>(lds_raw);  // [P][LDH]
  _Float16* const Bs = patch + P * LDH;                            // [2][BN][LDH]

  const int nblk = a.tiles_m * a.tiles_n;
  const int per_xcd = (nblk + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= nblk) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kc = tid & 7, r0 = tid >> 3;
  const int W = a.W, H = a.H, Cin = a.Cin;
  const int ncc = Cin / BKH, ntaps = ncc * 9;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w), 0, (int)a.w_bytes, 0x00020000);

  // patch rows of this thread: row = r0 + 32 j, pixel gp = m0 - (W + 1) + row (out of tensor -> zeros)
  const int64_t gp0 = m0 - (W + 1) + r0;
  auto patch_voff = [&](int j) -> unsigned {
    const int64_t gp = gp0 + 32 * j;
    return (r0 + 32 * j < P && gp >= 0 && gp < a.M) ? (unsigned)((gp * Cin + 8 * kc) * 2) : kOob;
  };
  unsigned wvoff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wvoff[i] = (unsigned)(((int64_t)(n0 + r0 + 32 * i) * a.Kpad + 8 * kc) * 2);
  _Float16* const Pst = patch + r0 * LDH + 8 * kc;
  _Float16* const Bst = Bs + r0 * LDH + 8 * kc;

  // fragment bases + validity of the 9 taps per fragment row
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  const _Float16* const Afr = patch + (wm + frow + W + 1) * LDH + fk;
  const _Float16* const Bfr = Bs + (wn + frow) * LDH + fk;
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

  halfx8 pr[NPC];  // next channel chunk of the patch
  halfx8 rb[2][NB];             // weights of taps t+1 / t+2 (alternating sets)
  auto load_patch = [&](int j, int cc) { pr[j] = load8(xrsrc, patch_voff(j), (unsigned)(cc * BKH * 2)); };
  auto store_patch = [&](int cc) {
    halfx8 ps, pb;
    if (PRE) {
      ps = *reinterpret_cast<const halfx8*>(a.pre_scale + cc * BKH + 8 * kc);
      pb = *reinterpret_cast<const halfx8*>(a.pre_shift + cc * BKH + 8 * kc);
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      if (j < npc && r0 + 32 * j < P) {
        halfx8 v = pr[j];
        if (PRE) {
          const bool real = patch_voff(j) != kOob;  // pixels outside the tensor stay zero
          v = v * ps + pb;
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = (real && v[q] > (_Float16)0) ? v[q] : (_Float16)0;
        }
        *reinterpret_cast<halfx8*>(Pst + 32 * j * LDH) = v;
      }
    }
  };
  auto load_b = [&](int set, int tt) {  // tt = cc * 9 + tap; K order is (tap, c)
    const int t2 = tt < ntaps ? tt : ntaps - 1;
    const int cc = t2 / 9, tap = t2 - cc * 9;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[set][i] = load8(wrsrc, wvoff[i], (unsigned)((tap * Cin + cc * BKH) * 2));
  };
  auto store_b = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<halfx8*>(Bst + buf * BN * LDH + 32 * i * LDH) = rb[set][i];
  };

  // prologue: patch of chunk 0, weights of taps 0 (-> LDS), 1 and 2 (-> registers)
#pragma unroll
  for (int j = 0; j < NPC; ++j)
    if (j < npc) load_patch(j, 0);
  load_b(0, 0);
  load_b(1, 1);
  store_patch(0);
  store_b(0, 0);
  load_b(0, 2);
  __syncthreads();

  auto tap_step = [&](int tt, int cc, int tap, auto par) {
    constexpr int Pb = decltype(par)::value;  // tt & 1: LDS weight buffer of this tap
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
    const _Float16* Ab = Afr + d * LDH;
    const _Float16* Bb = Bfr + Pb * BN * LDH;
    const bool more = tt + 1 < ntaps;
    const bool next_chunk = cc + 1 < ncc;
    halfx8 fa[2][MT], fb[2][NT];
    auto read_frags = [&](int set, int ks) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        halfx8 v = *reinterpret_cast<const halfx8*>(Ab + i * 32 * LDH + ks * 16);
        if (!((vmask[i] >> tap) & 1u)) v = halfx8{0, 0, 0, 0, 0, 0, 0, 0};
        fa[set][i] = v;
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) fb[set][i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH + ks * 16);
    };
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) read_frags((ks + 1) & 1, ks + 1);
      if (ks == 0 && more) store_b(1 - Pb, 1 - Pb);   // weights of tap tt+1 (register set (tt+1) & 1)
      if (ks == 1 && more) load_b(1 - Pb, tt + 3);    // ... and that set takes tap tt+3
      if (ks == 2 && next_chunk) {                    // next chunk's patch: a slice per tap (taps 0..7)
#pragma unroll
        for (int j = 0; j < NPC; ++j)
          if (j < npc && (j == 2 * tap || j == 2 * tap + 1)) load_patch(j, cc + 1);
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][mi], fb[ks & 1][ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    if (tap == 8 && next_chunk) {  // every wave is done with this chunk's patch: swap in the next one
      store_patch(cc + 1);
      __syncthreads();
    }
  };
  int tt = 0;
  for (int cc = 0; cc < ncc; ++cc) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++tt) {
      if ((cc * 9 + tap) & 1) tap_step(tt, cc, tap, std::integral_constant<int, 1>{});
      else tap_step(tt, cc, tap, std::integral_constant<int, 0>{});
    }
  }

  // ---- epilogue (as the generic kernel)
  float* const cl = reinterpret_cast<float*>(lds_raw);
  constexpr int LDC = BN + 4;
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
  constexpr int ITERS = BM * C8 / kThreads;
  const int c8 = tid % C8;
  const int n = n0 + 8 * c8;
  floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (a.bias) {
    b0 = *reinterpret_cast<const floatx4*>(a.bias + n);
    b1 = *reinterpret_cast<const floatx4*>(a.bias + n + 4);
  }
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const int row = tid / C8 + k * (kThreads / C8);
    const int64_t m = m0 + row;
    if (m < a.M) {
      floatx4 v0 = *reinterpret_cast<const floatx4*>(cl + row * LDC + 8 * c8) + b0;
      floatx4 v1 = *reinterpret_cast<const floatx4*>(cl + row * LDC + 8 * c8 + 4) + b1;
      if (a.residual) {
        const halfx8 rr = *reinterpret_cast<const halfx8*>(a.residual + m * a.Cout + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] += (float)rr[q]; v1[q] += (float)rr[4 + q]; }
      }
      if (a.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
      }
      halfx8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q] = (_Float16)v0[q]; o[4 + q] = (_Float16)v1[q]; }
      *reinterpret_cast<halfx8*>(a.y + m * a.Cout + n) = o;
    }
  }
}

template <int BN, bool PRE, int NPC>
int launch_patch_variant(ConvArgsH args, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& fl_) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_f16<BN, PRE, NPC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        fl_.spills = note_kernel(reinterpret_cast<const void*>(&conv3x3_patch_f16<BN, PRE, NPC>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  const int P = BM + 2 * args.W + 2;
  size_t lds = ((size_t)P * LDH + 2 * (size_t)BN * LDH) * 2;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  if (lds < epi) lds = epi;
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = args.Cout / BN;
  if (args.M >= (1ll << 31)) return fail(HP_ERR_ARG, "conv: more than 2^31 output pixels");
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int nblk = args.tiles_m * args.tiles_n;
  hipLaunchKernelGGL((conv3x3_patch_f16<BN, PRE, NPC>), dim3(8 * ((nblk + 7) / 8)), dim3(kThreads), lds, stream, args, P);
  return check_launch("conv3x3_patch_f16");
}

// instantiated patch sizes: BN = 64 (the 64-channel layers, 60x80 maps): NPC 10; BN = 128: NPC 5 / 6 / 7
int patch_f16_npc(const ConvArgsH& a, int kh, int kw) {
  if (kh != 3 || kw != 3 || a.stride != 1 || a.pad != 1 || a.Cin % BKH != 0 || a.Cout % 64 != 0) return 0;
  const int P = BM + 2 * a.W + 2;
  const int npc = (P * 8 + kThreads - 1) / kThreads;
  if (((size_t)P * LDH + 2 * 128 * LDH) * 2 > 80 * 1024) return 0;  // two workgroups per CU
  if (a.Cout % 128 != 0) return npc <= 10 ? 10 : 0;
  return npc <= 5 ? 5 : npc <= 6 ? 6 : npc <= 7 ? 7 : 0;
}

template <bool PRE>
int launch_patch(const ConvArgsH& a, int npc, hipStream_t stream) {
  if (a.Cout % 128 != 0) return launch_patch_variant<64, PRE, 10>(a, stream);
  if (npc == 5) return launch_patch_variant<128, PRE, 5>(a, stream);
  if (npc == 6) return launch_patch_variant<128, PRE, 6>(a, stream);
  return launch_patch_variant<128, PRE, 7>(a, stream);
}

template <int BN, bool PRE>
int launch_variant(ConvArgsH args, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& fl_) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16<BN, PRE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_f16<BN>()));
        fl_.spills = note_kernel(reinterpret_cast<const void*>(&conv_igemm_f16<BN, PRE>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = args.Cout / BN;
  if (args.M >= (1ll << 31)) return fail(HP_ERR_ARG, "conv: more than 2^31 output pixels");
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int nblk = args.tiles_m * args.tiles_n;
  hipLaunchKernelGGL((conv_igemm_f16<BN, PRE>), dim3(8 * ((nblk + 7) / 8)), dim3(kThreads), lds_bytes_f16<BN>(), stream, args);
  return check_launch("conv_igemm_f16");
}

// ---- network input: fp32 NHWC [.., c_in] -> fp16 NHWC [.., c_out] (c_out >= c_in, zero padded) ----
__global__ __launch_bounds__(256) void cast_pad_f32_f16(const float* x, _Float16* y, int64_t pixels, int c_in, int c_out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-half chunk of y each
  const int c8 = c_out / 8;
  if (idx >= pixels * c8) return;
  const int64_t p = idx / c8;
  const int c = (int)(idx % c8) * 8;
  halfx8 o;
#pragma unroll
  for (int q = 0; q < 8; ++q) o[q] = c + q < c_in ? (_Float16)x[p * c_in + c + q] : (_Float16)0;
  *reinterpret_cast<halfx8*>(y + p * c_out + c) = o;
}

// ---- 3x3 stride-2 pad-1 max pooling, NHWC fp16, 8 channels per lane ----
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_f16(const _Float16* x, _Float16* y, int n, int H, int W, int C,
                                                             int Ho, int Wo) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C8 = C / 8;
  if (idx >= (int64_t)n * Ho * Wo * C8) return;
  const int c8 = (int)(idx % C8);
  int64_t p = idx / C8;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int img = (int)(p / Ho);
  halfx8 m;
#pragma unroll
  for (int q = 0; q < 8; ++q) m[q] = (_Float16)(-65504.f);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if ((unsigned)iw >= (unsigned)W) continue;
      const halfx8 v = *reinterpret_cast<const halfx8*>(x + (((int64_t)img * H + ih) * W + iw) * C + 8 * c8);
#pragma unroll
      for (int q = 0; q < 8; ++q) m[q] = v[q] > m[q] ? v[q] : m[q];
    }
  }
  *reinterpret_cast<halfx8*>(y + (((int64_t)img * Ho + oh) * Wo + ow) * C + 8 * c8) = m;
}

}  // namespace

int launch_conv_f16(const ConvArgsH& a, hipStream_t stream) {
  if (a.x_bytes >= (1ll << 32) - 256 || a.w_bytes >= (1ll << 31))
    return fail(HP_ERR_ARG, "conv_igemm_f16: tensor too large for 32-bit buffer offsets (lower max_batch)");
  const bool pre = a.pre_scale != nullptr;
  if (conv_pp_f16_applicable(a)) return launch_conv_pp_f16(a, stream);  // 3x3 s1, Cout % 128 == 0, Cin % 64 == 0: conv_pp.hip
  if (const int npc = patch_f16_npc(a, a.kh, a.kw)) return pre ? launch_patch<true>(a, npc, stream) : launch_patch<false>(a, npc, stream);
  if (a.Cout % 128 == 0) return pre ? launch_variant<128, true>(a, stream) : launch_variant<128, false>(a, stream);
  return pre ? launch_variant<64, true>(a, stream) : launch_variant<64, false>(a, stream);
}

int launch_cast_pad_f16(const float* x, void* y, int64_t pixels, int c_in, int c_out, hipStream_t stream) {
  const int64_t total = pixels * (c_out / 8);
  hipLaunchKernelGGL(cast_pad_f32_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x,
                     reinterpret_cast<_Float16*>(y), pixels, c_in, c_out);
  return check_launch("cast_pad_f32_f16");
}

int launch_maxpool_f16(const void* x, void* y, int n, int H, int W, int C, int Ho, int Wo, hipStream_t stream) {
  const int64_t total = (int64_t)n * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const _Float16*>(x), reinterpret_cast<_Float16*>(y), n, H, W, C, Ho, Wo);
  return check_launch("maxpool3x3s2_nhwc_f16");
}

}  // namespace hp
