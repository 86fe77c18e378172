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
#include <cmath>
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
  static bool opted = false;
  if (!opted) {
    HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7x7s2_pool<MODE, SC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds));
    opted = true;
  }
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

int slab_of(int cin, int f16) { return f16 ? 16 : (cin % 8 == 0 ? 8 : 4); }

}  // namespace

// the layers this kernel is written for: 7x7 / stride 2 / pad 3, 64 output channels, ReLU, followed by the 3x3 / s2 / p1
// max-pool; fp32 input with Cin % 4 == 0 channels in memory, or the fp16 plan's 16-channel input
bool conv_stem7_applicable(int kh, int kw, int stride, int pad, int cin_mem, int cout, int relu, int f16) {
  static const bool off = std::getenv("HP_NO_STEM7_KERNEL") != nullptr;
  return !off && kh == 7 && kw == 7 && stride == 2 && pad == 3 && cout == BN && relu == HP_ACT_RELU &&
         (f16 ? cin_mem == 16 : (cin_mem % 4 == 0 && cin_mem <= 64));
}

// Plan time (host): PyTorch-layout weights [64][cin_real][7][7] with the folded-BN scale already applied ->
// [step = slab * 7 + kh][cout][hi k-steps | lo k-steps] halves, k = kw * SC + c inside a step, + (fp32 mode) the [64]
// power-of-two scale-back factors behind them.  Returns the byte size; h_out may be null to query it.
size_t conv_stem7_pack_weights(const float* h_w /* [64][cin_real][7][7], BN folded */, int cin_real, int cin_mem, int f16,
                               void* h_out) {
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

// a.x / a.Cin = the input as it lies in memory, a.w = conv_stem7_pack_weights, a.y = the POOLED map
int launch_conv_stem7_pool(const ConvArgs& a, int f16, hipStream_t stream) {
  if (f16) return launch_stem7<MODE_F16, 16>(a, stream);
  return a.Cin % 8 == 0 ? launch_stem7<MODE_SPLIT, 8>(a, stream) : launch_stem7<MODE_SPLIT, 4>(a, stream);
}

}  // namespace hp
