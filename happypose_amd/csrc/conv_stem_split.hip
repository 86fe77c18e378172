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
  static bool opted = false;
  if (!opted) {
    HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem5x5s2_pool_split),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
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
  hipLaunchKernelGGL(conv_stem5x5s2_pool_split, dim3((args.tiles_m + 7) / 8 * 8), dim3(kThreads), kLds, stream, args);
  return check_launch("conv_stem5x5s2_pool_split");
}

}  // namespace hp
