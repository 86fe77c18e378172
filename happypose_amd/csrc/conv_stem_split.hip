// The CosyPose stem in one launch: 5x5 / stride-2 / pad-2 convolution on the unpadded 6-channel network input +
// folded BN + ReLU + 3x3 / stride-2 / pad-1 max-pool (CP/models/wide_resnet.py:98-107: conv1, bn1, relu, maxpool),
// on the split-fp16 scheme of conv_split.hip (fp32 operands as fp16 hi / lo halves, three fp16 MFMAs per product,
// fp32 accumulation, per-cout power-of-two weight scaling).
//
// The gather version of this layer (conv_igemm_split.hip, POOL) is bound by operand delivery: every output pixel
// fetches its 5 x 30 floats from L2 again (each input pixel is used by 6.25 outputs) and every 128-row tile its 40 KB
// of weights.  Here a workgroup owns 3 x 16 POOLED pixels of one image = the 7 x 33 conv pixels under them (231 of
// its 256 GEMM rows) and stages what they need exactly once:
//   * the input region, 17 rows x 69 pixels x 6 channels = 414 contiguous floats per row (16-B aligned: tiles start at
//     even pixels), split into an fp16 hi plane and an fp16 lo plane in LDS (2 x 14 KB); zero outside the image;
//   * all 5 K-tiles (filter rows) of the split weights, 5 x 64 rows x [32 hi | 32 lo] halves (45 KB).
// The K order is the planner's filter-row packing (K-tile kh = the 5 taps x 6 channels of filter row kh = 30 contiguous
// floats of an input row, + 2 floats whose weights are zero), so the A fragment of conv pixel (dr, dc), K-tile kh,
// k-step kk is 8 contiguous halves at row 2 dr + kh, half offset 12 dc + 16 kk + 8 (lane >> 5): two ds_read_b64.
// 4 waves x (64 rows x 64 couts); two workgroups per CU (74 KB of LDS each) overlap one another's staging / epilogue.
// Epilogue: scale back, conv tile -> LDS [row][68], max over the 3 x 3 windows (conv pixels outside the map do not
// take part, as with PyTorch's -inf padding), bias, ReLU, 16-B stores of the pooled map.
#include <algorithm>
#include <cstdlib>

#include "conv.h"
#include "conv_epilogue.h"

namespace hp {

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr int PR = 3, PC = 16;                 // pooled tile
constexpr int CH = 2 * PR + 1, CW = 2 * PC + 1;  // conv tile 7 x 33
constexpr int IR = 2 * CH + 3;                 // 17 input rows
constexpr int NCH = 104;                       // 16-B chunks per input row (69 px x 6 ch = 414 floats -> 416)
constexpr int ROWP = 4 * NCH;                  // halves per staged input row
constexpr int PLANE = IR * ROWP;               // halves per plane (hi / lo)
constexpr int KT = 5;                          // K-tiles = filter rows
constexpr int LDB = 72;                        // weight row pitch (halves)
constexpr int BN = 64, BMR = 256;              // couts, GEMM rows
constexpr int LDC = BN + 4;
constexpr size_t kLdsLoop = ((size_t)2 * PLANE + (size_t)KT * BN * LDB) * 2;
constexpr size_t kLdsEpi = (size_t)BMR * LDC * 4;
constexpr size_t kLds = kLdsLoop > kLdsEpi ? kLdsLoop : kLdsEpi;

__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem5x5s2_pool_split(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const Ahi = reinterpret_cast<_Float16*>(lds_raw);  // [IR][ROWP]
  _Float16* const Alo = Ahi + PLANE;
  _Float16* const Bs = Alo + PLANE;                              // [KT][BN][LDB]

  const int nblk = a.tiles_m;
  const int per_xcd = (nblk + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= nblk) return;
  const int img = fdiv(lin, a.fd_howo);
  const int rem = lin - img * a.sk_S2;
  const int ty = fdiv(rem, a.fd_wo), tx = rem - ty * a.sk_S3;
  const int oh0 = 2 * PR * ty - 1, ow0 = 2 * PC * tx - 1;  // first conv pixel of the tile (may be -1)
  const int ih_base = 2 * oh0 - 2, fl_base = (2 * ow0 - 2) * 6;  // first input row / float offset inside an input row

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W;

  // ---- stage the input region (fp32 -> hi / lo planes) and the weights
  {
    constexpr int NIT = (IR * NCH + kThreads - 1) / kThreads;  // 7
    floatx4 v[NIT];
    const float* const ximg = a.x + (int64_t)img * H * W * 6;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * kThreads, rr = idx / NCH, j = idx - rr * NCH;
      const int ih = ih_base + rr, fl = fl_base + 4 * j;
      const bool ok = idx < IR * NCH && (unsigned)ih < (unsigned)H && fl >= 0 && fl + 4 <= W * 6;
      v[k] = ok ? *reinterpret_cast<const floatx4*>(ximg + ((int64_t)ih * W) * 6 + fl) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const _Float16* const wsplit = reinterpret_cast<const _Float16*>(a.w);
    constexpr int NWB = KT * BN * 8 / kThreads;  // 10 16-B chunks of weights per thread
    halfx8 wv[NWB];
#pragma unroll
    for (int k = 0; k < NWB; ++k) {
      const int idx = tid + k * kThreads, c8 = idx & 7, row = (idx >> 3) % BN, kt = idx / (8 * BN);
      wv[k] = *reinterpret_cast<const halfx8*>(wsplit + (size_t)row * (KT * 64) + kt * 64 + c8 * 8);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * kThreads;
      if (idx < IR * NCH) {
        const halfx4 hi = __builtin_convertvector(v[k], halfx4);
        const halfx4 lo = __builtin_convertvector(v[k] - __builtin_convertvector(hi, floatx4), halfx4);
        *reinterpret_cast<halfx4*>(Ahi + idx * 4) = hi;  // idx * 4 = rr * ROWP + 4 j
        *reinterpret_cast<halfx4*>(Alo + idx * 4) = lo;
      }
    }
#pragma unroll
    for (int k = 0; k < NWB; ++k) {
      const int idx = tid + k * kThreads, c8 = idx & 7, row = (idx >> 3) % BN, kt = idx / (8 * BN);
      *reinterpret_cast<halfx8*>(Bs + (kt * BN + row) * LDB + c8 * 8) = wv[k];
    }
  }
  __syncthreads();

  // ---- K loop: 5 filter rows x 2 k-steps x 3 MFMAs per 32 x 32 tile
  const int frow = lane & 31, hsel = lane >> 5;
  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int r = wave * 64 + mt * 32 + frow;
    r = r < CH * CW ? r : CH * CW - 1;  // rows past the tile repeat its last pixel; they are never read back
    const int dr = r / CW, dc = r - dr * CW;
    abase[mt] = 2 * dr * ROWP + 12 * dc + 8 * hsel;
  }
  const _Float16* const Bfr = Bs + frow * LDB + 8 * hsel;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto read_a = [&](const _Float16* plane, int mt, int kt, int kk) -> halfx8 {
    const _Float16* p = plane + abase[mt] + kt * ROWP + 16 * kk;
    const halfx4 lo4 = *reinterpret_cast<const halfx4*>(p);
    const halfx4 hi4 = *reinterpret_cast<const halfx4*>(p + 4);
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  };
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      halfx8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        ah[mt] = read_a(Ahi, mt, kt, kk);
        al[mt] = read_a(Alo, mt, kt, kk);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bh[nt] = *reinterpret_cast<const halfx8*>(Bfr + (kt * BN + nt * 32) * LDB + 16 * kk);
        bl[nt] = *reinterpret_cast<const halfx8*>(Bfr + (kt * BN + nt * 32) * LDB + 32 + 16 * kk);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }

  // ---- pooled epilogue
  const float* const unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(a.w) + (size_t)BN * (KT * 64));
  float* const cl = reinterpret_cast<float*>(lds_raw);
  __syncthreads();  // every wave is done with the staged operands
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float s = unscale[nt * 32 + frow];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
        cl[row * LDC + nt * 32 + frow] = acc[mt][nt][r] * s;
      }
  }
  __syncthreads();
  constexpr int C4 = BN / 4;
  const int Hp = (a.Ho - 1) / 2 + 1, Wp = (a.Wo - 1) / 2 + 1;
  float pool_chk = 0.f;
#pragma unroll
  for (int it0 = 0; it0 < PR * PC * C4; it0 += kThreads) {
    const int it = it0 + tid;
    const int c4 = it % C4, pp = it / C4, py = pp / PC, px = pp - py * PC;
    const int ph = PR * ty + py, pw = PC * tx + px;
    if (ph >= Hp || pw >= Wp) continue;
    floatx4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    floatx4 seen = {0.f, 0.f, 0.f, 0.f};  // v_max drops a NaN operand: the non-finite guard sums what the window reads
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int dr = 2 * py + dy, dc = 2 * px + dx;
        if ((unsigned)(oh0 + dr) < (unsigned)a.Ho && (unsigned)(ow0 + dc) < (unsigned)a.Wo)
        {
          const floatx4 cv = *reinterpret_cast<const floatx4*>(cl + (dr * CW + dc) * LDC + 4 * c4);
          best = __builtin_elementwise_max(best, cv);
          seen += cv;
        }
      }
    if (a.bias) best += *reinterpret_cast<const floatx4*>(a.bias + 4 * c4);
    best = __builtin_elementwise_max(best, floatx4{0.f, 0.f, 0.f, 0.f});
    *reinterpret_cast<floatx4*>(a.y + (((int64_t)img * Hp + ph) * Wp + pw) * BN + 4 * c4) = best;
    pool_chk += (seen[0] + seen[1]) + (seen[2] + seen[3]);
  }
  conv_report_nonfinite(a, pool_chk);
}

// ---- round 4: the same layer as a PERSISTENT two-group kernel (the scheme of conv_stem7x7s2_pool_f16_pp, conv_stem7.hip) ----
// The tile kernel above re-reads its 45 KB of weights from L2 for every 48-pooled-pixel tile, waits out an HBM round trip before
// its first MFMA, and the two workgroups of a CU run in lockstep (both staging, both multiplying, both pooling): 145 us per 64
// images where its MFMAs take ~50.  Here one 512-thread workgroup per CU walks tiles b, b + G, ...; the split weights, the
// scale-back factors and the bias stay in LDS; the two wave groups alternate roles per phase (two workgroup barriers each):
// M = the 10 k-steps x 12 MFMAs of a tile, fragments one step ahead, with the global loads of the group's next tile issued
// at its head; O = the pooled epilogue of the tile just multiplied, then those loads are split into hi / lo halves and
// stored.  Wave-local pooling as in the fp16 kernel: a wave's 64 GEMM rows are the 7 x 9 conv pixels under its 3 x 4 pooled
// pixels, the MFMAs run transposed (weights as A: a lane holds one pixel and 16 consecutive couts), scale-back + bias + ReLU
// happen in registers (all monotone, so they commute with the max: same values as the tile kernel, same MFMA order), the
// tile goes to LDS 32 couts at a time and the wave pools what it wrote itself.  The epilogue tile aliases the group's input.
#ifndef HP_S5_MID_KT
#define HP_S5_MID_KT 3
#endif
namespace s5p {
constexpr int kT = 512;
constexpr int ROWP2 = 424;                    // halves per staged input row: 416 + 8 (424 mod 64 = 40: see kRowPix)
constexpr int PLANE2 = IR * ROWP2;            // halves per plane
constexpr int IN_BYTES = 2 * PLANE2 * 2;      // hi + lo planes
constexpr int W_BYTES = KT * BN * LDB * 2;    // 46080
constexpr int C_BYTES = 2 * BN * 4;           // scale-back factors, bias
constexpr int EPITCH = (32 + 4) * 4;          // bytes per pixel of the epilogue tile: 32 couts + 16 B
constexpr int EPI_BYTES = 4 * 64 * EPITCH;    // per group: four waves x 64 pixels
constexpr int GRP_BYTES = IN_BYTES > EPI_BYTES ? IN_BYTES : EPI_BYTES;
constexpr size_t kLds2 = (size_t)W_BYTES + C_BYTES + 2 * GRP_BYTES;
constexpr int NIT = (IR * NCH + 255) / 256;   // 16-B chunks of the input region per thread of a group
static_assert(GRP_BYTES % 16 == 0 && (W_BYTES + C_BYTES) % 16 == 0 && ROWP2 % 4 == 0, "aligned carve-up");
// GEMM row 32 mt + (lane & 31) of a wave -> conv pixel 9 dr + dcl of its 7 x 9 block.  A ds_read_b64 is served 32 lanes at a
// time, each lane two banks: the fragments of a half-wave start at dword dr ROWP2 + 6 dcl (mod 64) and should all differ.
// The first 32 rows do; the other 32 carry five 2- / 3-way repeats (63 pixels do not spread evenly over 32 even residues).
__device__ constexpr unsigned char kRowPix[64] = {
    0, 9, 18, 27, 36, 1, 2, 3, 4, 8, 10, 11, 12, 19, 20, 21, 28, 29, 30, 37, 38, 39, 45, 46, 47, 48, 5, 6, 7, 55, 56, 57,
    13, 22, 31, 40, 49, 14, 15, 16, 17, 54, 23, 24, 25, 32, 33, 34, 41, 42, 43, 50, 51, 52, 58, 59, 60, 61, 26, 35, 44, 53, 62, 62};
}  // namespace s5p

__global__ __launch_bounds__(s5p::kT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem5x5s2_pool_split_pp(ConvArgs a) {
  using namespace s5p;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, wl = wave & 3, gt = tid & 255;
  _Float16* const Bs = reinterpret_cast<_Float16*>(lds_raw);                       // [KT][BN][LDB]
  float* const consts = reinterpret_cast<float*>(lds_raw + W_BYTES);                // [64] scale-back, [64] bias
  unsigned char* const Grp = lds_raw + W_BYTES + C_BYTES + grp * GRP_BYTES;
  _Float16* const Ahi = reinterpret_cast<_Float16*>(Grp);
  _Float16* const Alo = Ahi + PLANE2;
  const int H = a.H, W = a.W;
  const int Hp = (a.Ho - 1) / 2 + 1, Wp = (a.Wo - 1) / 2 + 1;

  {  // weights, scale-back factors, bias -> LDS, once
    const _Float16* const wsplit = reinterpret_cast<const _Float16*>(a.w);
    for (int idx = tid; idx < KT * BN * 8; idx += kT) {
      const int c8 = idx & 7, row = (idx >> 3) % BN, kt = idx / (8 * BN);
      *reinterpret_cast<halfx8*>(Bs + (kt * BN + row) * LDB + c8 * 8) =
          *reinterpret_cast<const halfx8*>(wsplit + (size_t)row * (KT * 64) + kt * 64 + c8 * 8);
    }
    if (tid < BN) {
      consts[tid] = reinterpret_cast<const float*>(wsplit + (size_t)BN * (KT * 64))[tid];
      consts[BN + tid] = a.bias ? a.bias[tid] : 0.f;
    }
  }
  // workgroup barrier that orders LDS traffic only (__syncthreads() also waits for the staging loads in flight)
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nblk = a.tiles_m, per_xcd = (nblk + 7) / 8, G = (int)gridDim.x;
  const int T = (8 * per_xcd - (int)blockIdx.x + G - 1) / G;
  struct Tile { int img, ty, tx; bool ok; };
  auto tile_of = [&](int j) -> Tile {
    Tile t{0, 0, 0, false};
    if (j < 0 || j >= T) return t;
    const int v = (int)blockIdx.x + G * j, lin = (v & 7) * per_xcd + (v >> 3);
    if (lin >= nblk) return t;
    t.img = fdiv(lin, a.fd_howo);
    const int rem = lin - t.img * a.sk_S2;
    t.ty = fdiv(rem, a.fd_wo); t.tx = rem - t.ty * a.sk_S3;
    t.ok = true;
    return t;
  };

  const int frow = lane & 31, hsel = lane >> 5;
  int boff[2], pix[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    pix[mt] = kRowPix[32 * mt + frow];
    const int dr = pix[mt] / 9, dcl = pix[mt] - 9 * dr;
    boff[mt] = 2 * dr * ROWP2 + 12 * (8 * wl + dcl) + 8 * hsel;  // halves
  }
  // MFMA row i of a 32-cout block multiplies weight row sigma(i) (conv_pp.hip): 16 consecutive couts per lane
  const int srow = 16 * ((frow >> 2) & 1) + 4 * (frow >> 3) + (frow & 3);
  const _Float16* const Wfr = Bs + srow * LDB + 8 * hsel;

  floatx16 acc[2][2];
  float chk = 0.f;
  floatx4 sv[NIT];  // the group's next tile: 16-B chunks of its input region (fp32), issued at the head of the M role
  auto issue_loads = [&](const Tile& ts) {
    if (!ts.ok) return;
#ifdef HP_S5_ABL_NOLOAD
    if (a.M > 0) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) sv[k] = floatx4{0.f, 0.f, 0.f, 0.f};
      return;
    }
#endif
    const int ih_base = 2 * (2 * PR * ts.ty - 1) - 2, fl_base = (2 * (2 * PC * ts.tx - 1) - 2) * 6;
    const float* const ximg = a.x + (int64_t)ts.img * H * W * 6;
    int g_ = gt;
    asm volatile("" : "+v"(g_));  // re-derive the chunk coordinates here: hoisted out of the tile loop they stay live through the M role
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = g_ + 256 * k, rr = idx / NCH, j = idx - rr * NCH;
      const int ih = ih_base + rr, fl = fl_base + 4 * j;
      const bool ok = idx < IR * NCH && (unsigned)ih < (unsigned)H && fl >= 0 && fl + 4 <= W * 6;
      sv[k] = ok ? *reinterpret_cast<const floatx4*>(ximg + ((int64_t)ih * W) * 6 + fl) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
  };

  auto role_m = [&](const Tile& t, const Tile& tnext) {
    issue_loads(tnext);
    if (t.ok) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    halfx8 xh[2][2], xl[2][2], wh[2][2], wlo[2][2];  // [slot][mt / nt]
    auto read_x = [&](const _Float16* plane, int mt, int kt, int kk) -> halfx8 {
      const _Float16* const p = plane + boff[mt] + kt * ROWP2 + 16 * kk;
      const halfx4 lo4 = *reinterpret_cast<const halfx4*>(p), hi4 = *reinterpret_cast<const halfx4*>(p + 4);
      return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto fetch = [&](int st, int slot) {
      const int kt = st >> 1, kk = st & 1;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { xh[slot][mt] = read_x(Ahi, mt, kt, kk); xl[slot][mt] = read_x(Alo, mt, kt, kk); }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        wh[slot][nt] = *reinterpret_cast<const halfx8*>(Wfr + (kt * BN + nt * 32) * LDB + 16 * kk);
        wlo[slot][nt] = *reinterpret_cast<const halfx8*>(Wfr + (kt * BN + nt * 32) * LDB + 32 + 16 * kk);
      }
    };
    __builtin_amdgcn_s_setprio(3);  // the multiplying wave goes first on its SIMD (its partner is in the O role)
    auto mm = [&](int sl) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {  // the tile kernel's order: x_hi w_hi, x_hi w_lo, x_lo w_hi
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sl][nt], xh[sl][mt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[sl][nt], xh[sl][mt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sl][nt], xl[sl][mt], acc[mt][nt], 0, 0, 0);
        }
    };
#ifdef HP_S5_ABL_NOMFMA  // diagnostics builds (tools/stem_ablate.sh): phases compiled out
    const bool mm_on = t.ok && a.M < 0;
#else
    const bool mm_on = t.ok;
#endif
    if (mm_on) fetch(0, 0);
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {  // two k-steps per filter row; rolled: unrolled ten times the addresses of all steps stay live
      if (mm_on) {
        fetch(2 * kt + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mm_on) {
        if (kt + 1 < KT) fetch(2 * kt + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the O group's mid-phase barrier, LATE in the M role: its first half (the fp32 epilogue through LDS) is the long one,
      // its second (hi / lo split + store of the staged chunks) short
      if (kt == HP_S5_MID_KT) lds_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    lds_barrier();
  };

  auto role_o = [&](const Tile& te, const Tile& ts, bool load_now) {
    if (load_now) issue_loads(ts);
#ifdef HP_S5_ABL_NOEPI
    if (te.ok && a.M < 0) {
#else
    if (te.ok) {
#endif
      unsigned char* const Ew = Grp + wl * 64 * EPITCH;
      const int oh0 = 2 * PR * te.ty - 1, ow0 = 2 * PC * te.tx - 1 + 8 * wl;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        floatx4 sc[4], bi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sc[q] = *reinterpret_cast<const floatx4*>(consts + 32 * nt + 16 * hsel + 4 * q);
          bi[q] = *reinterpret_cast<const floatx4*>(consts + BN + 32 * nt + 16 * hsel + 4 * q);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          floatx4 o[4];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[mt][nt][r] * sc[r >> 2][r & 3] + bi[r >> 2][r & 3];
            chk += v;  // before the ReLU (fmaxf drops a NaN): the non-finite guard
            o[r >> 2][r & 3] = fmaxf(v, 0.f);
          }
          floatx4* const dst = reinterpret_cast<floatx4*>(Ew + pix[mt] * EPITCH + 64 * hsel);
          dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
        }
        // the wave reads back what it wrote itself (the LDS serves a wave's accesses in order): 12 pooled pixels x 8 pieces of 4
        // couts = 96 items, lanes 0-31 take two.  (One 8-cout item per lane -- 18 reads in one round trip -- needs 72 registers
        // for them and spills: 276 instead of 254 us per 128 images.)
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int item = lane + 64 * rep;
          if (item < 96) {
            const int c4 = item & 7, pp = item >> 3, py = pp >> 2, pxl = pp & 3;
            const int ph = PR * te.ty + py, pw = PC * te.tx + 4 * wl + pxl;
            if (ph < Hp && pw < Wp) {
              // all nine reads issued back to back: a conv pixel outside the map is replaced by the window's centre, which is
              // always inside (guarded reads compile to nine dependent LDS round trips per item: 313 -> 254 us per 128 images)
              const unsigned char* const ctr = Ew + (9 * (2 * py + 1) + 2 * pxl + 1) * EPITCH + 16 * c4;
              floatx4 v[9];
#pragma unroll
              for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                  const int dr = 2 * py + dy, dcl = 2 * pxl + dx;
                  const bool in = (unsigned)(oh0 + dr) < (unsigned)a.Ho && (unsigned)(ow0 + dcl) < (unsigned)a.Wo;
                  v[3 * dy + dx] = *reinterpret_cast<const floatx4*>(in ? Ew + (9 * dr + dcl) * EPITCH + 16 * c4 : ctr);
                }
              floatx4 best = v[0];
#pragma unroll
              for (int k = 1; k < 9; ++k) best = __builtin_elementwise_max(best, v[k]);
              *reinterpret_cast<floatx4*>(a.y + (((int64_t)te.img * Hp + ph) * Wp + pw) * BN + 32 * nt + 4 * c4) = best;
            }
          }
        }
      }
    }
    lds_barrier();  // every wave of the group is done with the epilogue tile: the input region may be overwritten
    if (ts.ok) {
      int g_ = gt;
      asm volatile("" : "+v"(g_));
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = g_ + 256 * k, rr = idx / NCH, j = idx - rr * NCH;
        if (idx < IR * NCH) {
          const halfx4 hi = __builtin_convertvector(sv[k], halfx4);
          const halfx4 lo = __builtin_convertvector(sv[k] - __builtin_convertvector(hi, floatx4), halfx4);
          *reinterpret_cast<halfx4*>(Ahi + rr * ROWP2 + 4 * j) = hi;
          *reinterpret_cast<halfx4*>(Alo + rr * ROWP2 + 4 * j) = lo;
        }
      }
    }
    lds_barrier();
  };

  // phases p = -1 .. T: group g multiplies tile p when p - g is even, otherwise it finishes tile p - 1 and stages tile p + 1
  int p = -1;
  if (grp == 1) { lds_barrier(); lds_barrier(); p = 0; }
  bool first = true;
  while (true) {
    role_o(tile_of(p - 1), tile_of(p + 1), first);
    first = false;
    if (++p > T) break;
    role_m(tile_of(p), tile_of(p + 2));
    if (++p > T) break;
  }
  conv_report_nonfinite(a, chk);
}

}  // namespace

// the layer this kernel is written for: 5x5 / stride 2 / pad 2 on the unpadded 6-channel input in the planner's
// filter-row packing (Kpad = 5 x 32), 64 output channels, ReLU, even input width, followed by the 3x3 / s2 / p1 max-pool
bool conv_stem_split_applicable(const ConvArgs& a, int kh, int kw, int run_mode) {
  return run_mode && kh == 5 && kw == 5 && a.stride == 2 && a.pad == 2 && a.Cin == 6 && a.Cout == BN && a.Kpad == KT * 32 &&
         a.W % 2 == 0 && a.Wo == a.W / 2 && a.Ho == (a.H - 1) / 2 + 1 && a.relu == HP_ACT_RELU && !a.residual && !a.pre_scale &&
         (int64_t)a.H * a.W * 6 < (1ll << 31);
}

// a.w = weights split by conv_igemm_split_transform_weights (64 rows), a.y = the POOLED map [n][Hp][Wp][64]
int launch_conv_stem_split_pool(ConvArgs args, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch&) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem5x5s2_pool_split),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        return HP_OK;
      }))
    return rc0;
  const int Hp = (args.Ho - 1) / 2 + 1, Wp = (args.Wo - 1) / 2 + 1;
  const int tiles_y = (Hp + PR - 1) / PR, tiles_x = (Wp + PC - 1) / PC;
  const int n_img = (int)(args.M / ((int64_t)args.Ho * args.Wo));
  args.tiles_m = n_img * tiles_y * tiles_x;
  args.fd_howo = make_fastdiv((unsigned)(tiles_y * tiles_x));
  args.fd_wo = make_fastdiv((unsigned)tiles_x);
  args.sk_S2 = tiles_y * tiles_x;
  args.sk_S3 = tiles_x;
  if (!dbg(DBG_STEM5_OLD)) {  // (set: the tile kernel, for the A/B test)
    static FirstLaunch fl2;
    if (const int rc0 = fl2.once([](FirstLaunch&) {
          HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem5x5s2_pool_split_pp),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)s5p::kLds2));
          return HP_OK;
        }))
      return rc0;
    const int grid = std::min(8 * ((args.tiles_m + 7) / 8), conv_num_cus() / 8 * 8);
    hipLaunchKernelGGL(conv_stem5x5s2_pool_split_pp, dim3(grid), dim3(s5p::kT), s5p::kLds2, stream, args);
    return check_launch("conv_stem5x5s2_pool_split_pp");
  }
  hipLaunchKernelGGL(conv_stem5x5s2_pool_split, dim3((args.tiles_m + 7) / 8 * 8), dim3(kThreads), kLds, stream, args);
  return check_launch("conv_stem5x5s2_pool_split");
}

}  // namespace hp
