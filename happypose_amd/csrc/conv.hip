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
// Mapping to the hardware (wave64, 4 waves per workgroup):
//   * block tile BM x BN = 128x128 (128x64 for the 64-channel layers), BK = 32; the 4 waves
//     sit 2x2, each owns a 64x64 (64x32) sub-tile = 2x2 (2x1) MFMA tiles of 32x32 -> 64 (32)
//     accumulator VGPRs;
//   * global -> registers -> LDS staging with 16-B loads (one filter tap row of a pixel is a
//     128-B line shared by 8 adjacent lanes), zero padding by predication, double-buffered
//     LDS with one barrier per K-tile; the loads of tile t+1 are issued before the MFMAs of
//     tile t;
//   * LDS tiles are [rows][32+4] floats: the +4 pad makes both the 16-B staging stores and
//     the 16-B fragment reads (ds_read_b128: lane -> row = lane&31, k = 4*(lane>>5)..+3)
//     bank-conflict free.  One ds_read_b128 feeds FOUR k-steps: the order of K inside a
//     K-tile is permuted (k-step j of a group multiplies k = j and k = 4+j), which is legal
//     because A and B use the same permutation;
//   * per K-tile a wave issues 64 MFMAs (4096 matrix-pipe cycles) against 16 ds_read_b128
//     and ~8 global loads: the kernel is bound by the fp32 matrix pipe (157 TFLOP/s), see
//     DESIGN.md for the roofline;
//   * workgroup ids are renumbered so that each XCD walks a contiguous range of tiles
//     (neighbouring tiles share activation rows / weight panels in that XCD's L2).
// Numerics: exact fp32 FMA chains (the f32 MFMA is bitwise an fmaf chain), K-order differs
// from the reference's oneDNN/cuDNN kernels, so results agree to fp32 round-off, not bitwise.
#include "conv.h"

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

template <int NA, int NB>
__device__ __forceinline__ void load_tile(const ConvArgs& a, int t, int kc, const int64_t (&rowoff)[NA],
                                          const int (&ih0)[NA], const int (&iw0)[NA],
                                          const float* const (&wrow)[NB], floatx4 (&ra)[NA], floatx4 (&rb)[NB]) {
  const int4 e = a.lut[t * 8 + kc];  // {offset, kh, kw, channel}; kh < 0 -> K padding
  floatx4 ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};
  const bool pre = a.pre_scale != nullptr;
  if (pre && e.y >= 0) {
    ps = *reinterpret_cast<const floatx4*>(a.pre_scale + e.w);
    pb = *reinterpret_cast<const floatx4*>(a.pre_shift + e.w);
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int ih = ih0[i] + e.y, iw = iw0[i] + e.z;
    const bool ok = (e.y >= 0) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      v = *reinterpret_cast<const floatx4*>(a.x + rowoff[i] + e.x);
      if (pre) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(v[q], ps[q], pb[q]), 0.f);
      }
    }
    ra[i] = v;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const floatx4*>(wrow[i] + t * BK);
}

template <int NA, int NB>
__device__ __forceinline__ void store_tile(float* Ast, float* Bst, const floatx4 (&ra)[NA], const floatx4 (&rb)[NB]) {
#pragma unroll
  for (int i = 0; i < NA; ++i) *reinterpret_cast<floatx4*>(Ast + 32 * i * LDK) = ra[i];
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<floatx4*>(Bst + 32 * i * LDK) = rb[i];
}

template <int BM, int BN>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_igemm_f32(ConvArgs a) {
  using TT = Tile<BM, BN>;
  constexpr int MT = TT::MT, NT = TT::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                       // [2][BM][LDK]
  float* Bs = lds + 2 * BM * LDK;        // [2][BN][LDK]

  // XCD-aware tile order (dispatch puts block b on XCD b % 8)
  const int nblk = a.tiles_m * a.tiles_n;
  const int per_xcd = (nblk + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= nblk) return;
  const int tile_m = lin / a.tiles_n, tile_n = lin % a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kc = tid & 7;         // which float4 of the 32-float K-tile row
  const int r0 = tid >> 3;        // first staged row (0..31); rows r0 + 32*i

  // ---- per-row (output pixel) constants for the A gather ----
  int64_t rowoff[TT::A_CHUNKS];
  int ih0[TT::A_CHUNKS], iw0[TT::A_CHUNKS];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < TT::A_CHUNKS; ++i) {
    const int64_t m = m0 + r0 + 32 * i;
    if (m < a.M) {
      const int img = (int)(m / HoWo);
      const int rem = (int)(m - (int64_t)img * HoWo);
      const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
      ih0[i] = oh * a.stride - a.pad;
      iw0[i] = ow * a.stride - a.pad;
      rowoff[i] = (((int64_t)img * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
    } else {
      ih0[i] = -(1 << 28); iw0[i] = 0; rowoff[i] = 0;
    }
  }
  const float* wrow[TT::B_CHUNKS];
#pragma unroll
  for (int i = 0; i < TT::B_CHUNKS; ++i) wrow[i] = a.w + (int64_t)(n0 + r0 + 32 * i) * a.Kpad + 4 * kc;

  floatx4 ra[TT::A_CHUNKS], rb[TT::B_CHUNKS];
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

  load_tile(a, 0, kc, rowoff, ih0, iw0, wrow, ra, rb);
  store_tile(Ast, Bst, ra, rb);
  __syncthreads();

  for (int t = 0; t < a.ktiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < a.ktiles) load_tile(a, t + 1, kc, rowoff, ih0, iw0, wrow, ra, rb);
    const float* Ab = As + buf * BM * LDK + (wm + frow) * LDK + fk;
    const float* Bb = Bs + buf * BN * LDK + (wn + frow) * LDK + fk;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      floatx4 av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + kg * 8);
#pragma unroll
      for (int i = 0; i < NT; ++i) bv[i] = *reinterpret_cast<const floatx4*>(Bb + i * 32 * LDK + kg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
    }
    if (t + 1 < a.ktiles) store_tile(Ast + (buf ^ 1) * BM * LDK, Bst + (buf ^ 1) * BN * LDK, ra, rb);
    __syncthreads();
  }

  // ---- epilogue: bias, residual, ReLU; NHWC store (32 consecutive channels per half-wave) ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn + nt * 32 + (lane & 31);
      const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M) {
          float v = acc[mt][nt][r] + bias;
          if (a.residual) v += a.residual[m * a.Cout + n];
          if (a.relu) v = fmaxf(v, 0.f);
          a.y[m * a.Cout + n] = v;
        }
      }
    }
  }
}

// ---- 3x3 stride-2 pad-1 max pooling, NHWC, 4 channels per lane ------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const float* x, float* y, int n, int H, int W,
                                                         int C, int Ho, int Wo) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C / 4;
  const int64_t total = (int64_t)n * Ho * Wo * C4;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  int64_t p = idx / C4;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int img = (int)(p / Ho);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if ((unsigned)iw >= (unsigned)W) continue;
      const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)img * H + ih) * W + iw) * C + 4 * c4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  *reinterpret_cast<float4*>(y + (((int64_t)img * Ho + oh) * Wo + ow) * C + 4 * c4) = m;
}

// ---- head: spatial mean -> [fc 512x512 + bias] -> pose / logits linear heads -----------
// one workgroup per sample; features [HW][C] NHWC.
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
  __shared__ float feat[512];
  __shared__ float feat2[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = a.x + (int64_t)b * a.HW * a.C;
  for (int c = tid; c < a.C; c += 256) {
    float s = 0.f;
    for (int p = 0; p < a.HW; ++p) s += x[(int64_t)p * a.C + c];
    feat[c] = s / (float)a.HW;
  }
  __syncthreads();
  const float* f = feat;
  if (a.fc_w) {  // torchvision ResNet: avgpool -> fc (MP/models/torchvision_resnet.py:337-341)
    for (int o = tid; o < a.C; o += 256) {
      const float* w = a.fc_w + (int64_t)o * a.C;
      float s = 0.f;
      for (int c = 0; c < a.C; ++c) s = fmaf(w[c], feat[c], s);
      feat2[o] = s + a.fc_b[o];
    }
    __syncthreads();
    f = feat2;
  }
  if (a.features) for (int c = tid; c < a.C; c += 256) a.features[(int64_t)b * a.C + c] = f[c];
  // linear heads: one wave per output row, lanes stride the 512 features
  const int lane = tid & 63, wave = tid >> 6;
  for (int o = wave; o < a.pose_dim + a.n_logits; o += 4) {
    const bool is_pose = o < a.pose_dim;
    const int oo = is_pose ? o : o - a.pose_dim;
    const float* w = (is_pose ? a.pose_w : a.logit_w) + (int64_t)oo * a.C;
    float s = 0.f;
    for (int c = lane; c < a.C; c += 64) s = fmaf(w[c], f[c], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
      if (is_pose) { if (a.pose_out) a.pose_out[(int64_t)b * a.pose_dim + oo] = s + a.pose_b[oo]; }
      else { if (a.logit_out) a.logit_out[(int64_t)b * a.n_logits + oo] = s + a.logit_b[oo]; }
    }
  }
}

int launch_conv(const ConvArgs& a, int variant, hipStream_t stream) {
  ConvArgs args = a;
  if (variant == 0) {
    constexpr int BM = 128, BN = 128;
    args.tiles_m = (int)((a.M + BM - 1) / BM);
    args.tiles_n = a.Cout / BN;
    const int nblk = args.tiles_m * args.tiles_n;
    const size_t lds = Tile<BM, BN>::LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN>), dim3(8 * ((nblk + 7) / 8)), dim3(kThreads), lds, stream, args);
  } else {
    constexpr int BM = 128, BN = 64;
    args.tiles_m = (int)((a.M + BM - 1) / BM);
    args.tiles_n = a.Cout / BN;
    const int nblk = args.tiles_m * args.tiles_n;
    const size_t lds = Tile<BM, BN>::LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN>), dim3(8 * ((nblk + 7) / 8)), dim3(kThreads), lds, stream, args);
  }
  return check_launch("conv_igemm_f32");
}

int launch_maxpool(const float* x, float* y, int n, int H, int W, int C, int Ho, int Wo, hipStream_t stream) {
  const int64_t total = (int64_t)n * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, n, H, W, C, Ho, Wo);
  return check_launch("maxpool3x3s2_nhwc");
}

int launch_head(const HeadArgs& a, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(head_kernel, dim3(batch), dim3(256), 0, stream, a);
  return check_launch("head_kernel");
}

int conv_setup_once() {
  static bool done = false;
  if (done) return HP_OK;
  // > 64 KB of dynamic LDS needs the opt-in attribute
  HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<128, 128>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(Tile<128, 128>::LDS_FLOATS * sizeof(float))));
  HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<128, 64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(Tile<128, 64>::LDS_FLOATS * sizeof(float))));
  done = true;
  return HP_OK;
}

}  // namespace hp
