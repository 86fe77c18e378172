// The MegaPose stems in one launch: 7x7 / stride-2 / pad-3 convolution + folded BN + ReLU + 3x3 / stride-2 / pad-1
// max-pool (MP/models/torchvision_resnet.py:216-219,325-330: conv1, bn1, relu, maxpool) for any input channel count
// (9 coarse, 27 RGB refiner, 32 RGB-D refiner), in two arithmetic modes:
//   MODE_SPLIT  fp32 input [n][H][W][cp] (cp = channels rounded to 4), fp32 operands as fp16 hi / lo halves, three MFMAs
//               per product (conv_split.hip's scheme), fp32 pooled output;
//   MODE_F16    the fp16 plan of configuration C5: fp16 input [n][H][W][16], one MFMA per product, fp16 pooled output.
//
// The gather kernels these layers ran on (conv_igemm_split.hip / conv_f16.hip) fetch every input pixel once per tap and
// output pixel from L2 (each pixel is used by 12.25 outputs) and write the full-resolution conv map for the max-pool to
// read again (C3: 629 MB + 629 MB; C5: 1.4 GB + 1.4 GB per 576 views): 1.19 ms of a 4.3 ms C3 forward, 2.41 ms of a
// 12.8 ms C5 chunk.  Here -- the scheme of conv_stem_split.hip (the CosyPose 5x5 stem), generalised -- a workgroup owns
// 3 x 16 POOLED pixels of one image = the 7 x 33 conv pixels under them (231 of its 256 GEMM rows) and stages the input
// region they need, 19 rows x 71 pixels, ONCE per channel slab of SC channels (SC = 16 in fp16: one slab; 8 or 4 in
// fp32: cp / SC slabs), as [row][pixel][SC] halves (an fp16 hi and an fp16 lo plane in MODE_SPLIT).  For filter row kh
// the 7 taps x SC channels of a conv pixel are then 7 SC CONTIGUOUS halves of input row 2 dr + kh starting at pixel 2 dc:
// the A fragment of k-step kk is one ds_read_b128 at half offset 2 dc SC + 16 kk + 8 (lane >> 5).  The K loop walks
// (slab, kh) steps of ceil(7 SC / 16) k-steps (the tail of the last k-step multiplies real pixels with zero weights);
// the weights of a step, packed at plan time as [step][64 couts][hi | lo], are double buffered in LDS, their loads one
// step ahead in registers.  4 waves x (64 rows x 64 couts); two workgroups per CU overlap one another's staging and
// pooled epilogue (conv tile -> LDS -> 3x3 / s2 maxima, conv pixels outside the map excluded as with PyTorch's -inf
// padding -> bias -> ReLU -> 16-B stores).  The conv map is never written.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "conv.h"
#include "conv_epilogue.h"

namespace hp {

typedef _Float16 s7_halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 s7_halfx4 __attribute__((ext_vector_type(4)));
typedef float s7_floatx16 __attribute__((ext_vector_type(16)));
typedef float s7_floatx4 __attribute__((ext_vector_type(4)));

namespace {

enum { MODE_SPLIT = 0, MODE_F16 = 1 };
constexpr int kThreads = 256;
constexpr int PR = 3, PC = 16;                   // pooled tile
constexpr int CH = 2 * PR + 1, CW = 2 * PC + 1;  // conv tile 7 x 33
constexpr int IR = 2 * CH + 5, IPX = 2 * CW + 5; // input region 19 rows x 71 pixels
constexpr int BN = 64, BMR = 256, LDC = BN + 4;

template <int MODE, int SC>
struct Stem7 {
  static constexpr int KS = (7 * SC + 15) / 16;          // k-steps of 16 per (slab, kh) step
  // Staged row layout.  SC = 4: [pixel][4] -- consecutive conv pixels (consecutive lanes of an A fragment, 2 input pixels
  // apart) are 16 B apart: conflict-free 16-B reads.  With [pixel][SC] the lane stride is 32 B (SC = 8) / 64 B (SC = 16):
  // 2- / 4-way bank conflicts on every A fragment (the fp16 stem ran at a twelfth of its MFMA time).  SC >= 8 therefore
  // stores a row as two planes of 8-half chunks, [plane][position][8]: SC = 8: plane = pixel parity, position = pixel / 2
  // (lanes < 32 of k-step kk read pixel 2 dc + 2 kk, lanes >= 32 the odd pixel after it); SC = 16: plane = channel half
  // (the lane's k half), position = parity * HP + pixel / 2 (k-step kk = pixel 2 dc + kk).  Either way a fragment is
  // row + hsel * plane pitch + (dc + f(kk)) * 8 halves: 16 B from lane to lane.
  static constexpr bool DEINT = SC >= 8;
  static constexpr int HP = (IPX + 3) / 2;               // positions per pixel parity (2 pixels of slack for the last k-step)
  static constexpr int NPOS = SC == 16 ? 2 * HP : HP;    // positions per plane
  static constexpr int ROWP = DEINT ? 2 * NPOS * 8 + 8   // halves per staged row (16-B multiple; + 16 B between rows)
                                    : ((IPX + 2) * SC + 7) / 8 * 8 + 8;
  // half offset of the 8-half (SC >= 8) or SC-half chunk `q` of staged pixel cc inside its row
  static constexpr __host__ __device__ int pix_off(int cc, int q) {
    if (SC == 16) return q * NPOS * 8 + ((cc & 1) * HP + (cc >> 1)) * 8;
    if (SC == 8) return (cc & 1) * NPOS * 8 + (cc >> 1) * 8 + 4 * q;  // q: 4-half piece of the pixel's 8 channels
    return cc * SC + 4 * q;
  }
  // half offset of k-step kk relative to the fragment base of a conv pixel
  static constexpr __host__ __device__ int kstep_off(int kk) {
    if (SC == 16) return ((kk & 1) * HP + (kk >> 1)) * 8;
    if (SC == 8) return kk * 8;
    return 16 * kk;
  }
  static constexpr int PLANE = IR * ROWP + 16;           // halves per plane
  static constexpr int NPLANES = MODE == MODE_SPLIT ? 2 : 1;
  static constexpr int WROW = (NPLANES * KS * 16 + 31) / 32 * 32;  // halves per cout and step: [hi k-steps | lo k-steps], padded
  static constexpr int LDB = WROW + 8;                   // LDS pitch of a weight row
  static constexpr size_t kLdsLoop = ((size_t)NPLANES * PLANE + 2 * (size_t)BN * LDB) * 2;
  static constexpr size_t kLdsEpi = (size_t)BMR * LDC * 4;
  static constexpr size_t kLds = kLdsLoop > kLdsEpi ? kLdsLoop : kLdsEpi;
};

// a.x: MODE_SPLIT fp32 [n][H][W][Cin], MODE_F16 fp16 [n][H][W][16]; a.w: packed steps [NS * 7][64][WROW] halves
// (+ [64] fp32 scale-back factors behind them in MODE_SPLIT); a.y: POOLED map [n][Hp][Wp][64] (fp32 / fp16)
template <int MODE, int SC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem7x7s2_pool(ConvArgs a) {
  using S = Stem7<MODE, SC>;
  constexpr int KS = S::KS, ROWP = S::ROWP, PLANE = S::PLANE, WROW = S::WROW, LDB = S::LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const Ahi = reinterpret_cast<_Float16*>(lds_raw);  // [IR][ROWP]
  _Float16* const Alo = Ahi + PLANE;                             // MODE_SPLIT only
  _Float16* const Bs = Ahi + S::NPLANES * PLANE;                 // [2][BN][LDB]

  const int nblk = a.tiles_m;
  const int per_xcd = (nblk + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= nblk) return;
  const int img = fdiv(lin, a.fd_howo);
  const int rem = lin - img * a.sk_S2;
  const int ty = fdiv(rem, a.fd_wo), tx = rem - ty * a.sk_S3;
  const int oh0 = 2 * PR * ty - 1, ow0 = 2 * PC * tx - 1;  // first conv pixel of the tile (may be -1)
  const int ih_base = 2 * oh0 - 3, iw_base = 2 * ow0 - 3;  // first input row / column of the staged region

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const int NS = Cin / SC, nsteps = NS * 7;

  // zero the staged planes once: the slack positions are read by the last k-step against zero weights (must be finite)
  {
    const s7_halfx8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < S::NPLANES * PLANE / 8; i += kThreads) *reinterpret_cast<s7_halfx8*>(Ahi + 8 * i) = z8;
    __syncthreads();
  }

  // ---- staging helpers
  constexpr int PPT = MODE == MODE_SPLIT ? SC / 4 : SC / 8;  // 16-B pieces per pixel and slab
  constexpr int NPIECE = IR * IPX * PPT;
  constexpr int NIT = (NPIECE + kThreads - 1) / kThreads;
  auto stage_slab = [&](int slab) {
    if constexpr (MODE == MODE_SPLIT) {
      const float* const ximg = a.x + (int64_t)img * H * W * Cin + slab * SC;
      s7_floatx4 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * kThreads, px = idx / PPT, q = idx - px * PPT, rr = px / IPX, cc = px - rr * IPX;
        const int ih = ih_base + rr, iw = iw_base + cc;
        const bool ok = idx < NPIECE && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        v[k] = ok ? *reinterpret_cast<const s7_floatx4*>(ximg + ((int64_t)ih * W + iw) * Cin + 4 * q) : s7_floatx4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * kThreads;
        if (idx < NPIECE) {
          const int px = idx / PPT, q = idx - px * PPT, rr = px / IPX, cc = px - rr * IPX;
          const s7_halfx4 hi = __builtin_convertvector(v[k], s7_halfx4);
          const s7_halfx4 lo = __builtin_convertvector(v[k] - __builtin_convertvector(hi, s7_floatx4), s7_halfx4);
          *reinterpret_cast<s7_halfx4*>(Ahi + rr * ROWP + S::pix_off(cc, q)) = hi;
          *reinterpret_cast<s7_halfx4*>(Alo + rr * ROWP + S::pix_off(cc, q)) = lo;
        }
      }
    } else {
      const _Float16* const ximg = reinterpret_cast<const _Float16*>(a.x) + (int64_t)img * H * W * Cin + slab * SC;
      s7_halfx8 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * kThreads, px = idx / PPT, q = idx - px * PPT, rr = px / IPX, cc = px - rr * IPX;
        const int ih = ih_base + rr, iw = iw_base + cc;
        const bool ok = idx < NPIECE && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const s7_halfx8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        v[k] = ok ? *reinterpret_cast<const s7_halfx8*>(ximg + ((int64_t)ih * W + iw) * Cin + 8 * q) : zero;
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * kThreads;
        if (idx < NPIECE) {
          const int px = idx / PPT, q = idx - px * PPT, rr = px / IPX, cc = px - rr * IPX;
          *reinterpret_cast<s7_halfx8*>(Ahi + rr * ROWP + (SC == 16 ? S::pix_off(cc, q) : cc * SC + 8 * q)) = v[k];
        }
      }
    }
  };
  constexpr int NWB = BN * WROW / 8 / kThreads;  // 16-B weight pieces per thread and step
  static_assert(BN * WROW / 8 % kThreads == 0, "weight pieces must divide evenly");
  const _Float16* const wpk = reinterpret_cast<const _Float16*>(a.w);
  s7_halfx8 wv[NWB];
  auto load_w = [&](int step) {
    const int s2 = step < nsteps ? step : nsteps - 1;
#pragma unroll
    for (int k = 0; k < NWB; ++k) {
      const int idx = tid + k * kThreads;
      wv[k] = *reinterpret_cast<const s7_halfx8*>(wpk + (size_t)s2 * BN * WROW + (size_t)idx * 8);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NWB; ++k) {
      const int idx = tid + k * kThreads, row = idx / (WROW / 8), c8 = idx - row * (WROW / 8);
      *reinterpret_cast<s7_halfx8*>(Bs + (buf * BN + row) * LDB + c8 * 8) = wv[k];
    }
  };

  // ---- fragment bases
  const int frow = lane & 31, hsel = lane >> 5;
  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int r = wave * 64 + mt * 32 + frow;
    r = r < CH * CW ? r : CH * CW - 1;  // rows past the tile repeat its last pixel; they are never read back
    const int dr = r / CW, dc = r - dr * CW;
    abase[mt] = S::DEINT ? 2 * dr * ROWP + hsel * S::NPOS * 8 + dc * 8 : 2 * dr * ROWP + 2 * dc * SC + 8 * hsel;
  }
  const _Float16* const Bfr = Bs + frow * LDB + 8 * hsel;

  s7_floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- K loop over (slab, kh) steps
  load_w(0);
  stage_slab(0);
  store_w(0);
  load_w(1);
  __syncthreads();
  for (int slab = 0; slab < NS; ++slab) {
    if (slab > 0) {  // every wave is done with the previous slab (the barrier at the end of its last step)
      stage_slab(slab);
      __syncthreads();
    }
#pragma unroll 1
    for (int kh = 0; kh < 7; ++kh) {
      const int step = slab * 7 + kh, buf = step & 1;
      const _Float16* const Bb = Bfr + buf * BN * LDB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        s7_halfx8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          ah[mt] = *reinterpret_cast<const s7_halfx8*>(Ahi + abase[mt] + kh * ROWP + S::kstep_off(kk));
          if (MODE == MODE_SPLIT) al[mt] = *reinterpret_cast<const s7_halfx8*>(Alo + abase[mt] + kh * ROWP + S::kstep_off(kk));
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          bh[nt] = *reinterpret_cast<const s7_halfx8*>(Bb + nt * 32 * LDB + 16 * kk);
          if (MODE == MODE_SPLIT) bl[nt] = *reinterpret_cast<const s7_halfx8*>(Bb + nt * 32 * LDB + KS * 16 + 16 * kk);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            if (MODE == MODE_SPLIT) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            }
          }
      }
      store_w(1 - buf);   // weights of step + 1 (loaded during the previous step) ...
      load_w(step + 2);   // ... and the registers take step + 2
      __syncthreads();
    }
  }

  // ---- pooled epilogue
  float* const cl = reinterpret_cast<float*>(lds_raw);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float s = 1.f;
    if (MODE == MODE_SPLIT) s = reinterpret_cast<const float*>(wpk + (size_t)nsteps * BN * WROW)[nt * 32 + frow];
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
    s7_floatx4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    s7_floatx4 seen = {0.f, 0.f, 0.f, 0.f};  // v_max drops a NaN operand: the non-finite guard sums what the window reads
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int dr = 2 * py + dy, dc = 2 * px + dx;
        if ((unsigned)(oh0 + dr) < (unsigned)a.Ho && (unsigned)(ow0 + dc) < (unsigned)a.Wo) {
          const s7_floatx4 cv = *reinterpret_cast<const s7_floatx4*>(cl + (dr * CW + dc) * LDC + 4 * c4);
          best = __builtin_elementwise_max(best, cv);
          seen += cv;
        }
      }
    if (a.bias) best += *reinterpret_cast<const s7_floatx4*>(a.bias + 4 * c4);
    best = __builtin_elementwise_max(best, s7_floatx4{0.f, 0.f, 0.f, 0.f});
    const int64_t o = (((int64_t)img * Hp + ph) * Wp + pw) * BN + 4 * c4;
    if constexpr (MODE == MODE_SPLIT) {
      *reinterpret_cast<s7_floatx4*>(a.y + o) = best;
    } else {
      *reinterpret_cast<s7_halfx4*>(reinterpret_cast<_Float16*>(a.y) + o) = __builtin_convertvector(best, s7_halfx4);
    }
    pool_chk += (seen[0] + seen[1]) + (seen[2] + seen[3]);
  }
  if (MODE == MODE_SPLIT) conv_report_nonfinite(a, pool_chk);
}

template <int MODE, int SC>
int launch_stem7(ConvArgs args, hipStream_t stream) {
  using S = Stem7<MODE, SC>;
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch&) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7x7s2_pool<MODE, SC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds));
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
  hipLaunchKernelGGL((conv_stem7x7s2_pool<MODE, SC>), dim3((args.tiles_m + 7) / 8 * 8), dim3(kThreads), S::kLds, stream, args);
  return check_launch("conv_stem7x7s2_pool");
}


// ---- fp16 plan with <= 9 real input channels (the MegaPose coarse model: 9), round 4: a PERSISTENT two-group kernel ------
// What the tile kernel above costs on that input (C5, 576 views per launch, profiles/r04d_*: 1.55 ms, matrix pipe 0.42 busy on
// K padded 9 -> 16 channels): every 48-pooled-pixel tile re-reads all 7 x 64 x 128 weight halves from L2 (6.6 GB per launch),
// zero-fills and stages its input behind an exposed HBM round trip, and the two workgroups of a CU run in lockstep -- both
// staging, both multiplying, both pooling at the same time.  Here:
//   * one 512-thread workgroup per CU walks tiles b, b + G, ...; the weights of all seven filter rows stay in LDS (77 KB);
//   * K order (kw, c) over a TWELVE-slot pixel: the staged row is [pixel][12] halves, so the fragment of conv pixel dc starts
//     48 dc bytes into the row: one aligned, conflict-free ds_read_b128 (kRowPix).  35 k-steps per tile instead of 49
//     (s7p::KSTEPS).  (A ten-slot pixel was built first: its fragments are only 8-byte aligned -- two ds_read_b64 at half
//     the rate with 2-way bank conflicts whatever the row pitch, and the four multiplying waves ask the LDS for more
//     cycles than their MFMAs take.)
//   * the two wave groups (waves 0-3 / 4-7: one wave per SIMD each) alternate ROLES per phase, two workgroup barriers per
//     phase: M = the 35 k-steps of a tile back to back, fragments fetched two k-steps ahead, and the global loads of the
//     group's next tile issued at its head; O = the pooled epilogue of the tile just multiplied, then those loads go to LDS.
//     A SIMD always has one wave in its MFMA stream and one doing everything else (the ping-pong of conv_pp.hip, at tile
//     granularity);
//   * wave-local pooling: a wave's 64 GEMM rows are the 7 x 9 conv pixels under its 3 x 4 pooled pixels (the column
//     shared with the neighbour is computed twice), the MFMAs run transposed (weights as A: a lane holds one pixel and
//     16 consecutive couts), so bias + ReLU + fp16 rounding happen in registers (max commutes with all three), the tile
//     goes to LDS as 32-B pieces, 32 couts at a time, and the wave pools what it wrote itself -- no barrier inside the
//     epilogue; the epilogue tile aliases the group's input region.
namespace s7p {
constexpr int kT = 512;
constexpr int CPP = 12;                  // staged slots per pixel: channels 0-8, the shifted copy of channel 8, two zeros
constexpr int CMAX = 9;                  // real input channels this layout holds
constexpr int PXB = 2 * CPP;             // bytes per staged pixel
constexpr int NPX = 72;                  // staged pixels per row: 71 + the one the last k-step's zero-weight slots read
constexpr int PITCH = 1744;              // bytes per staged row (72 x 24 = 1728): with kRowPix below the fragment reads are conflict-free
constexpr int IN_BYTES = IR * PITCH;     // 19 rows
// K slots of a filter row: pixel kw, channel c -> 12 kw + c.  Six pixels and channels 0-7 of the seventh fill 80 slots = 5
// k-steps; the one product left over, (kw 6, channel 8), rides in the first pixel's spare slot 9: the staging writes channel 8
// of pixel q + 6 there (a shifted copy), and slot 9 carries the weight of (kw 6, channel 8).  35 k-steps per tile, not 42.
constexpr int KSTEPS = 5;
constexpr int WROWB = (KSTEPS * 16 + 8) * 2;   // bytes per (kh, cout) weight row: 80 halves + 16 B (bank spread)
constexpr int W_BYTES = 7 * BN * WROWB;  // 78848
constexpr int EPITCH = (BN + 8) * 2;     // bytes per pixel of the epilogue tile: 64 halves + 16 B
constexpr int EPI_BYTES = 4 * 64 * EPITCH;     // per group: four waves x 64 pixels
constexpr int GRP_BYTES = (IN_BYTES > EPI_BYTES ? IN_BYTES : EPI_BYTES);  // the epilogue tile aliases the group's input
constexpr size_t kLds = (size_t)W_BYTES + 2 * GRP_BYTES;
constexpr int NIT = (IR * NPX + 255) / 256;    // staged pixels per thread of a group
static_assert(GRP_BYTES % 16 == 0 && W_BYTES % 16 == 0 && PITCH % 16 == 0, "16-B aligned carve-up");
static_assert(kLds <= 160 * 1024, "one workgroup per CU");
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
// GEMM row r = 32 mt + (lane & 31) of a wave -> conv pixel 9 dr + dcl of its 7 x 9 block (63 pixels, one twice).  A
// ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md); the 16
// fragments of a group start at (2 dr PITCH + 48 dcl) and must fall into 16 different 16-B bank granules mod 256 B.  The
// table gives every group 16 different granules except the last one (three 2-way pairs); generated by spreading the
// occurrences of each granule over the four groups.
__device__ constexpr unsigned char kRowPix[64] = {
    0, 1, 2, 3, 11, 12, 13, 14, 15, 16, 17, 46, 4, 5, 6, 7, 36, 20, 21, 29, 8, 9, 10, 18, 19, 27, 28, 37, 30, 38, 39, 48,
    22, 23, 24, 25, 33, 34, 35, 54, 56, 58, 42, 43, 26, 55, 45, 57, 51, 52, 60, 61, 47, 31, 32, 40, 41, 49, 50, 59, 44, 53, 62, 62};
}  // namespace s7p

// a.x: fp16 [n][H][W][16] (channels >= 9 are never used); a.w: [7][64][88] halves (conv_stem7_pack_weights, layout 2);
// a.y: pooled fp16 map [n][Hp][Wp][64]
__global__ __launch_bounds__(s7p::kT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem7x7s2_pool_f16_pp(ConvArgs a) {
  using namespace s7p;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, wl = wave & 3, gt = tid & 255;
  unsigned char* const Wl = lds_raw;
  unsigned char* const Grp = lds_raw + W_BYTES + grp * GRP_BYTES;
  const int H = a.H, W = a.W;
  const _Float16* const xg = reinterpret_cast<const _Float16*>(a.x);
  _Float16* const yg = reinterpret_cast<_Float16*>(a.y);
  const int Hp = (a.Ho - 1) / 2 + 1, Wp = (a.Wo - 1) / 2 + 1;

  for (int i = tid; i < W_BYTES / 16; i += kT) reinterpret_cast<uintx4*>(Wl)[i] = reinterpret_cast<const uintx4*>(a.w)[i];

  // tiles of this workgroup: virtual indices v = b + G j (same XCD for all j: G % 8 == 0), XCD-contiguous renumbering
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

  // ---- fragment bases (bytes); the wave's 7 x 9 block starts at conv column 8 wl of the tile
  const int frow = lane & 31, hsel = lane >> 5;
  int boff[2], pix[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    pix[mt] = kRowPix[32 * mt + frow];
    const int dr = pix[mt] / 9, dcl = pix[mt] - 9 * dr;
    boff[mt] = 2 * dr * PITCH + 2 * (8 * wl + dcl) * PXB + 16 * hsel;
  }
  // MFMA row i of a 32-cout block multiplies weight row sigma(i) (conv_pp.hip): the lane's 16 accumulator rows are couts
  // 16 hsel .. 16 hsel + 15 of the block
  const int srow = 16 * ((frow >> 2) & 1) + 4 * (frow >> 3) + (frow & 3);
  const int aoff = srow * WROWB + 16 * hsel;

  // workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the staging loads a group
  // has in flight across its M role and for the pooled stores of the O role (an HBM round trip per phase)
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  s7_floatx16 acc[2][2];
  s7_floatx4 bias_r[2][4];  // the lane's 2 x 16 couts (fetched once: inside the epilogue the loads would queue behind the staging loads)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias_r[nt][q] = a.bias ? *reinterpret_cast<const s7_floatx4*>(a.bias + 32 * nt + 16 * hsel + 4 * q) : s7_floatx4{0.f, 0.f, 0.f, 0.f};

  // staging registers: the group's NEXT tile, pixel = 16 B (channels 0-7) + 8 B (8-11) of its 32-B record.  The loads are issued
  // at the head of the M role (two phases before the tile is multiplied) and land under its MFMA stream; the O role that
  // follows only moves them to LDS.  (Issued at the head of the O role they cost the phase an exposed HBM round trip.)
  uintx4 sv[NIT];
  unsigned sw[NIT];  // the dword holding channel 8 (a third load per pixel -- channel 8 of the pixel six to the right, so that the
                     // record could be composed in registers and stored as three ds_write_b64 -- cost 450 us per launch: TA-bound)
  auto issue_loads = [&](const Tile& ts) {
    if (!ts.ok) return;
#ifdef HP_S7_ABL_NOLOAD
    if (a.M > 0) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) { sv[k] = uintx4{0u, 0u, 0u, 0u}; sw[k] = 0u; }
      return;
    }
#endif
    const int ih_base = 2 * (2 * PR * ts.ty - 1) - 3, iw_base = 2 * (2 * PC * ts.tx - 1) - 3;
    const _Float16* const ximg = xg + (int64_t)ts.img * H * W * 16;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = gt + 256 * k, rr = idx / NPX, cc = idx - rr * NPX;
      const int ih = ih_base + rr, iw = iw_base + cc;
      const bool ok = idx < IR * NPX && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      const _Float16* const p = ximg + ((int64_t)ih * W + iw) * 16;
      sv[k] = ok ? *reinterpret_cast<const uintx4*>(p) : uintx4{0u, 0u, 0u, 0u};
      sw[k] = ok ? *reinterpret_cast<const unsigned*>(p + 8) : 0u;
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
    // 35 k-steps, fragments fetched TWO steps ahead of their MFMAs (left to itself the compiler issues each ds_read right
    // before its use and waits for it: the LDS round trip, ~100+ cycles under the other group's traffic, every step)
    constexpr int NSTEP = 7 * KSTEPS;
    s7_halfx8 px[3][2], wt[3][2];
    auto fetch = [&](int st, int slot) {
      const int kh = st / KSTEPS, kk = st - kh * KSTEPS;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) px[slot][mt] = *reinterpret_cast<const s7_halfx8*>(Grp + boff[mt] + kh * PITCH + 32 * kk);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        wt[slot][nt] = *reinterpret_cast<const s7_halfx8*>(Wl + (kh * BN + 32 * nt) * WROWB + aoff + 32 * kk);
    };
#ifdef HP_S7_ABL_NOMFMA  // diagnostics builds (tools/stem_ablate.sh): phases compiled out
    const bool mm_on = t.ok && a.M < 0;
#else
    const bool mm_on = t.ok;
#endif
#ifndef HP_S7_ABL_NOPRIO
    __builtin_amdgcn_s_setprio(3);  // the multiplying wave goes first on its SIMD (its partner is in the O role)
#endif
    if (mm_on) { fetch(0, 0); fetch(1, 1); }
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      if (mm_on) {
        if (st + 2 < NSTEP) fetch(st + 2, (st + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt[st % 3][nt], px[st % 3][mt], acc[mt][nt], 0, 0, 0);  // D[cout][pixel]
        __builtin_amdgcn_sched_barrier(0);
      }
      if (st == 4 * KSTEPS - 1) lds_barrier();  // the O group's mid-phase barrier
    }
#ifndef HP_S7_ABL_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    lds_barrier();
  };

  auto role_o = [&](const Tile& te, const Tile& ts, bool load_now) {
    if (load_now) issue_loads(ts);
    // ---- pooled epilogue of the tile this group multiplied in the previous phase
#ifdef HP_S7_ABL_NOEPI
    if (te.ok && a.M < 0) {
#else
    if (te.ok) {
#endif
      unsigned char* const Ew = Grp + wl * 64 * EPITCH;
      const int oh0 = 2 * PR * te.ty - 1, ow0 = 2 * PC * te.tx - 1 + 8 * wl;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          s7_halfx8 o[2];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(acc[mt][nt][r] + bias_r[nt][r >> 2][r & 3], 0.f);
            o[r >> 3][r & 7] = (_Float16)v;
          }
          unsigned char* const dst = Ew + pix[mt] * EPITCH + 64 * nt + 32 * hsel;
          *reinterpret_cast<s7_halfx8*>(dst) = o[0];
          *reinterpret_cast<s7_halfx8*>(dst + 16) = o[1];
        }
      // the wave reads back what it wrote itself (the LDS serves a wave's accesses in order): 12 pooled pixels x 8 pieces of
      // 8 couts = 96 items, lanes 0-31 take two
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int item = lane + 64 * rep;
        if (item < 96) {
          const int c8 = item & 7, pp = item >> 3, py = pp >> 2, pxl = pp & 3;
          const int ph = PR * te.ty + py, pw = PC * te.tx + 4 * wl + pxl;
          if (ph < Hp && pw < Wp) {
            // all nine reads issued back to back: a conv pixel outside the map is replaced by the window's centre, which is
            // always inside (guarded reads compile to nine dependent LDS round trips per item)
            const unsigned char* const ctr = Ew + (9 * (2 * py + 1) + 2 * pxl + 1) * EPITCH + 16 * c8;
            s7_halfx8 v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) {
                const int dr = 2 * py + dy, dcl = 2 * pxl + dx;
                const bool in = (unsigned)(oh0 + dr) < (unsigned)a.Ho && (unsigned)(ow0 + dcl) < (unsigned)a.Wo;
                v[3 * dy + dx] = *reinterpret_cast<const s7_halfx8*>(in ? Ew + (9 * dr + dcl) * EPITCH + 16 * c8 : ctr);
              }
            s7_halfx8 best = v[0];
#pragma unroll
            for (int k = 1; k < 9; ++k) best = __builtin_elementwise_max(best, v[k]);
            *reinterpret_cast<s7_halfx8*>(yg + (((int64_t)te.img * Hp + ph) * Wp + pw) * BN + 8 * c8) = best;
          }
        }
      }
    }
    lds_barrier();  // every wave of the group is done with the epilogue tile: the input region may be overwritten
    if (ts.ok) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = gt + 256 * k, rr = idx / NPX, cc = idx - rr * NPX;
        if (idx < IR * NPX) {
          unsigned char* const d = Grp + rr * PITCH + cc * PXB;
          reinterpret_cast<uintx2*>(d)[0] = uintx2{sv[k][0], sv[k][1]};
          reinterpret_cast<uintx2*>(d)[1] = uintx2{sv[k][2], sv[k][3]};
          const unsigned short c8 = (unsigned short)(sw[k] & 0xFFFFu);
          reinterpret_cast<unsigned short*>(d)[8] = c8;                                  // channel 8 ...
          if (cc >= 6) reinterpret_cast<unsigned short*>(d - 6 * PXB)[9] = c8;           // ... and its copy six pixels to the left
          if (cc >= NPX - 6) reinterpret_cast<unsigned short*>(d)[9] = 0;                // (nobody writes the last six)
          reinterpret_cast<unsigned*>(d)[5] = 0u;                                        // slots 10, 11: whatever the record's pad channels hold stays out
        }
      }
    }
    lds_barrier();
  };

  // phases p = -1 .. T: group g multiplies tile p when p - g is even, otherwise it finishes tile p - 1 and stages tile p + 1
  // (tiles of parity g).  Group 1 idles through phase -1.
  int p = -1;
  if (grp == 1) { lds_barrier(); lds_barrier(); p = 0; }
  bool first = true;  // the group's first tile has no M role before it to carry its loads
  while (true) {
    role_o(tile_of(p - 1), tile_of(p + 1), first);
    first = false;
    if (++p > T) break;
    role_m(tile_of(p), tile_of(p + 2));
    if (++p > T) break;
  }
}

template <int DUMMY = 0>
int launch_stem7_f16_pp(ConvArgs args, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch&) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7x7s2_pool_f16_pp),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)s7p::kLds));
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
  const int per_xcd = (args.tiles_m + 7) / 8;
  const int grid = std::min(8 * per_xcd, conv_num_cus() / 8 * 8);
  hipLaunchKernelGGL(conv_stem7x7s2_pool_f16_pp, dim3(grid), dim3(s7p::kT), s7p::kLds, stream, args);
  return check_launch("conv_stem7x7s2_pool_f16_pp");
}

// the persistent fp16 kernel takes the layers whose real channels fit its pixel layout (<= 9) (HP_STEM7_F16_OLD=1: the tile kernel)
bool stem7_f16_pp(int cin_real) {
  return !dbg(DBG_STEM7_F16_OLD) && cin_real > 0 && cin_real <= s7p::CMAX;
}

int slab_of(int cin, int f16) { return f16 ? 16 : (cin % 8 == 0 ? 8 : 4); }

}  // namespace

// the layers this kernel is written for: 7x7 / stride 2 / pad 3, 64 output channels, ReLU, followed by the 3x3 / s2 / p1
// max-pool; fp32 input with Cin % 4 == 0 channels in memory, or the fp16 plan's 16-channel input
bool conv_stem7_applicable(int kh, int kw, int stride, int pad, int cin_mem, int cout, int relu, int f16) {
  return kh == 7 && kw == 7 && stride == 2 && pad == 3 && cout == BN && relu == HP_ACT_RELU &&
         (f16 ? cin_mem == 16 : (cin_mem % 4 == 0 && cin_mem <= 64));
}

// Plan time (host): PyTorch-layout weights [64][cin_real][7][7] with the folded-BN scale already applied ->
// [step = slab * 7 + kh][cout][hi k-steps | lo k-steps] halves, k = kw * SC + c inside a step, + (fp32 mode) the [64]
// power-of-two scale-back factors behind them.  Returns the byte size; h_out may be null to query it.
size_t conv_stem7_pack_weights(const float* h_w /* [64][cin_real][7][7], BN folded */, int cin_real, int cin_mem, int f16,
                               void* h_out) {
  if (f16 && stem7_f16_pp(cin_real)) {  // layout 2 (conv_stem7x7s2_pool_f16_pp): [kh][cout][88 halves], k = 12 kw + c; (kw 6, c 8) -> slot 9
    const size_t bytes = (size_t)s7p::W_BYTES;
    if (!h_out) return bytes;
    std::vector<_Float16> out(bytes / 2, (_Float16)0.f);
    for (int o = 0; o < BN; ++o)
      for (int c = 0; c < cin_real; ++c)
        for (int kh = 0; kh < 7; ++kh)
          for (int kw = 0; kw < 7; ++kw)
            out[((size_t)kh * BN + o) * (s7p::WROWB / 2) + ((kw == 6 && c == 8) ? 9 : kw * s7p::CPP + c)] = (_Float16)h_w[(((size_t)o * cin_real + c) * 7 + kh) * 7 + kw];
    std::memcpy(h_out, out.data(), bytes);
    return bytes;
  }
  const int SC = slab_of(cin_mem, f16), KS = (7 * SC + 15) / 16, NS = cin_mem / SC, nsteps = NS * 7;
  const int wrow = ((f16 ? 1 : 2) * KS * 16 + 31) / 32 * 32;
  const size_t halves = (size_t)nsteps * BN * wrow, bytes = halves * 2 + (f16 ? 0 : BN * 4);
  if (!h_out) return bytes;
  std::vector<_Float16> out(halves, (_Float16)0.f);
  std::vector<float> unscale(BN, 1.f);
  for (int o = 0; o < BN; ++o) {
    int sh = 0;
    if (!f16) {  // per-cout power-of-two scaling so that max |w| lands in [2^13, 2^14) (conv_split.hip)
      float mx = 0.f;
      for (int i = 0; i < cin_real * 49; ++i) mx = std::fmax(mx, std::fabs(h_w[(size_t)o * cin_real * 49 + i]));
      int e = 0;
      if (mx > 0.f && mx < 3.0e38f) (void)std::frexp(mx, &e);
      sh = mx > 0.f ? 14 - e : 0;
      unscale[o] = std::ldexp(1.f, -sh);
    }
    for (int c = 0; c < cin_real; ++c) {
      const int slab = c / SC, cs = c - slab * SC;
      for (int kh = 0; kh < 7; ++kh)
        for (int kw = 0; kw < 7; ++kw) {
          const float v = std::ldexp(h_w[(((size_t)o * cin_real + c) * 7 + kh) * 7 + kw], sh);
          const int k = kw * SC + cs;
          _Float16* row = out.data() + ((size_t)(slab * 7 + kh) * BN + o) * wrow;
          const _Float16 hi = (_Float16)v;
          row[k] = hi;
          if (!f16) row[KS * 16 + k] = (_Float16)(v - (float)hi);
        }
    }
  }
  std::memcpy(h_out, out.data(), halves * 2);
  if (!f16) std::memcpy(reinterpret_cast<char*>(h_out) + halves * 2, unscale.data(), BN * 4);
  return bytes;
}

// K elements per filter row the fp16 launch multiplies (7 x 16 channels in 7 k-steps, or 80 slots in 5): bench.py's executed FLOPs
int conv_stem7_f16_krow(int cin_real) { return stem7_f16_pp(cin_real) ? s7p::KSTEPS * 16 : 112; }

// a.x / a.Cin = the input as it lies in memory, a.w = conv_stem7_pack_weights, a.y = the POOLED map; a.Kpad = the REAL input
// channel count the weights were packed for (it selects the same layout as conv_stem7_pack_weights did)
int launch_conv_stem7_pool(const ConvArgs& a, int f16, hipStream_t stream) {
  if (f16 && stem7_f16_pp(a.Kpad)) return launch_stem7_f16_pp<>(a, stream);
  if (f16) return launch_stem7<MODE_F16, 16>(a, stream);
  return a.Cin % 8 == 0 ? launch_stem7<MODE_SPLIT, 8>(a, stream) : launch_stem7<MODE_SPLIT, 4>(a, stream);
}

}  // namespace hp
