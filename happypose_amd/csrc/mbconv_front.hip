// Front half of an EfficientNet MBConv block in ONE launch (CP/models/efficientnet.py MBConvBlock.forward:
// _expand_conv -> _bn0 -> swish -> _depthwise_conv -> _bn1 -> swish, and the spatial sums squeeze-excitation needs):
//
//     x [n][H][W][Cin]  --1x1, BN, swish-->  E (never leaves the CU)  --k x k depthwise / stride s, BN, swish-->  D
//
// The expanded tensor E is 6x the block input and, unfused, is written once and read once through HBM (1.4 GB for the
// first block of stage 2 at batch 128); those two passes were ~30 % of a forward.  Here a workgroup owns a TH x TW tile
// of D of one image.  It stages the input pixels under the tile (with the depthwise halo) ONCE as fp16 hi / lo halves
// (the split-fp16 scheme of conv_igemm_split.hip: three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate, the
// layer's own split weights and scale-back factors), then walks the expanded channels 32 at a time:
//   expansion GEMM  [pixels x Cin] x [Cin x 32]  -> BN shift + swish -> E tile in LDS (zero outside the image: the
//                   depthwise convolution pads E, not x)
//   depthwise       from LDS, one lane = 4 channels x a run of output rows (sliding window, weights in registers)
//   store D, add the stored values into per-(image, tile, channel) sums (SE pooling partials, fixed order).
// Restricted to blocks with Cin <= 64 (the staged input tile must fit LDS next to E): stages 2-4 of EfficientNet-b3,
// where the large expanded maps are.  The FMA order of the depthwise part equals dwconv_strip_kernel's; the expansion
// equals conv_igemm_split_f32's products with another summation tree (within the split scheme's 2e-5 of max|ref|).
#include "conv.h"

#include <cstdlib>

namespace hp {

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCC = 32;        // expanded channels per pass
constexpr int kLDE = kCC + 4;  // E row pitch (floats)
constexpr int kMaxRows = 256;  // GEMM rows (input pixels of a tile, padded to 32): at most 8 MFMA row tiles, two per wave

template <int K, int S> struct FrontTile {
  static constexpr int TH = S == 1 ? 8 : 4, TW = S == 1 ? 16 : 8;
  static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K, P = IH * IW;
  static constexpr int MP = (P + 31) / 32 * 32;  // staged GEMM rows
  static_assert(MP <= kMaxRows, "tile does not fit the GEMM rows");
};

template <int KP> constexpr int lda() { return 2 * KP + 8; }  // halves per staged pixel: [KP/32][32 hi | 32 lo] + pad

template <int K, int S, int KP>
constexpr size_t front_lds_bytes() {
  return (size_t)FrontTile<K, S>::MP * lda<KP>() * 2 + (size_t)FrontTile<K, S>::P * kLDE * 4 + kMaxRows /* pixel-in-image flags */ +
         256 * 16 /* pool */ +
         (size_t)(K * K + 1) * kCC * 4 /* depthwise weights + shift of the pass */;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division: the expanded tile is activated with
// its halo (1.4-1.9x the elements the unfused launches touch) and the division's ten instructions per element made the
// launch VALU-bound
__device__ __forceinline__ float swish1(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ floatx4 swish4(floatx4 v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = swish1(v[q]);
  return v;
}

template <int K, int S, int KP>
__global__ __launch_bounds__(256, 2) void mbconv_front_kernel(FrontArgs a) {
  using T = FrontTile<K, S>;
  constexpr int LDA = lda<KP>();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const As = reinterpret_cast<_Float16*>(lds_raw);                              // [MP][LDA]
  float* const Es = reinterpret_cast<float*>(lds_raw + (size_t)T::MP * LDA * 2);            // [P][kLDE]
  unsigned char* const inimg = reinterpret_cast<unsigned char*>(Es + T::P * kLDE);        // [kMaxRows]
  floatx4* const red = reinterpret_cast<floatx4*>(inimg + kMaxRows);                           // [256]
  float* const Wd = reinterpret_cast<float*>(red + 256);                                  // [K*K + 1][kCC]: taps, then the BN shift

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x;
  const int oh0 = ty * T::TH, ow0 = tx * T::TW;
  const int ih0 = oh0 * S - a.pad_t, iw0 = ow0 * S - a.pad_l;

  // ---- stage the input pixels of the tile: fp32 -> fp16 hi / lo, zero outside the image and beyond Cin
  const float* const ximg = a.x + (int64_t)img * a.H * a.W * a.Cin;
  constexpr int Q = KP / 4;
  for (int idx = tid; idx < T::MP * Q; idx += 256) {
    const int p = idx / Q, c = 4 * (idx - p * Q);
    const int r = p / T::IW, cc = p - r * T::IW;
    const int ih = ih0 + r, iw = iw0 + cc;
    const bool in = p < T::P && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (in && c < a.Cin) v = *reinterpret_cast<const floatx4*>(ximg + ((int64_t)ih * a.W + iw) * a.Cin + c);
    const halfx4 hi = __builtin_convertvector(v, halfx4);
    const halfx4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), halfx4);
    _Float16* dst = As + p * LDA + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<halfx4*>(dst) = hi;
    *reinterpret_cast<halfx4*>(dst + 32) = lo;
    if (c == 0) inimg[p] = in ? 1 : 0;
  }
  __syncthreads();

  const _Float16* const wsplit = reinterpret_cast<const _Float16*>(a.w_split);
  const float* const unscale = reinterpret_cast<const float*>(wsplit + (size_t)a.rows_pad * KP * 2);
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  constexpr int n_rt = (T::P + 31) / 32;  // live MFMA row tiles
  // depthwise lane mapping: 8 channel quads x output positions
  const int c4 = tid & 7;
  const int col = (tid >> 3) % T::TW;
  const int rgrp = (tid >> 3) / T::TW;  // S == 1: 2 groups of 4 rows; S == 2: 4 single rows
  constexpr int RT = S == 2 ? 1 : (K == 5 ? 2 : 4);  // output rows per lane and pass (5x5: 25 taps leave room for two accumulators)
  constexpr int NPASS = S == 2 ? 1 : 4 / RT;         // S == 1: a lane owns 4 of the 8 rows of its column
  constexpr int NR = (RT - 1) * S + K;  // E rows a lane walks
  float chk = 0.f;
  // The expansion's MFMAs run TRANSPOSED (weights as the A operand, rows in the order sigma(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) +
  // (i & 3): conv_pp.hip): a lane holds ONE staged pixel per row tile and 16 CONSECUTIVE expanded channels, so the activated
  // tile goes to LDS as four 16-B stores per row tile (round 5; one channel of 16 pixels per lane meant 16 ds_write_b32 with
  // an address each).  In-image flag of this lane's pixel in row tile wave + 4 i: bit i.
  const int srow = 16 * ((frow >> 2) & 1) + 4 * (frow >> 3) + (frow & 3);
  const int h16 = 16 * (lane >> 5);
  unsigned live = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (wave + 4 * i) * 32 + frow;
    if (p < T::P && inimg[p]) live |= 1u << i;
  }

  const int n_chunks = (a.Cexp + kCC - 1) / kCC;
  // the pass's weights are fetched one pass ahead (L2 latency is ~1 us: unhidden it was a third of a pass): B fragments of
  // the expansion in registers, depthwise taps + shift through one float4 per lane
  halfx8 bfrag[2 * (KP / 16)];
  floatx4 wd_next = {0.f, 0.f, 0.f, 0.f};
  constexpr int kWdQuads = (K * K + 1) * (kCC / 4);
  static_assert(kWdQuads <= 256, "one depthwise weight quad per lane");
  auto fetch = [&](int n0) {
    const _Float16* const wrow = wsplit + (size_t)(n0 + srow) * KP * 2 + fk;
#pragma unroll
    for (int kk = 0; kk < KP / 16; ++kk) {
      const int ko = (kk >> 1) * 64 + (kk & 1) * 16;
      bfrag[2 * kk] = *reinterpret_cast<const halfx8*>(wrow + ko);
      bfrag[2 * kk + 1] = *reinterpret_cast<const halfx8*>(wrow + ko + 32);
    }
    if (tid < kWdQuads) {
      const int row = tid / (kCC / 4), cg = n0 + 4 * (tid - row * (kCC / 4));
      wd_next = floatx4{0.f, 0.f, 0.f, 0.f};
      if (cg < a.Cexp) wd_next = *reinterpret_cast<const floatx4*>((row < K * K ? a.w_dw + (size_t)row * a.Cexp : a.bias_d) + cg);
    }
  };
  fetch(0);
#pragma unroll 1
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int n0 = ch * kCC;
    // ---- expansion: this wave's row tiles x 32 channels, B fragments straight from global (L2-resident weights)
    floatx16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP / 16; ++kk) {
      const int ko = (kk >> 1) * 64 + (kk & 1) * 16;
      const halfx8 bh = bfrag[2 * kk], bl = bfrag[2 * kk + 1];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rt = wave + 4 * i;
        if (rt < n_rt) {
          const _Float16* ap = As + (rt * 32 + frow) * LDA + ko + fk;
          const halfx8 ah = *reinterpret_cast<const halfx8*>(ap);
          const halfx8 al = *reinterpret_cast<const halfx8*>(ap + 32);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[i], 0, 0, 0);  // D[channel][pixel]
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[i], 0, 0, 0);
        }
      }
    }
    // ---- BN shift + swish -> E (a lane holds channels n0 + h16 .. + 15 of pixel frow of each of its row tiles)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + h16 + 4 * g;
      const floatx4 us = *reinterpret_cast<const floatx4*>(unscale + n);  // rows padded to whole passes by the planner
      floatx4 be = {0.f, 0.f, 0.f, 0.f};
      if (n < a.Cexp) be = *reinterpret_cast<const floatx4*>(a.bias_e + n);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pp = (wave + 4 * i) * 32 + frow;
        if (wave + 4 * i < n_rt && pp < T::P) {
          floatx4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float pre = fmaf(acc[i][4 * g + q], us[q], be[q]);
            chk += pre;
            v[q] = (live >> i) & 1u ? swish1(pre) : 0.f;
          }
          *reinterpret_cast<floatx4*>(Es + pp * kLDE + h16 + 4 * g) = v;
        }
      }
    }
    // the pass's depthwise taps go through LDS (per-lane global reads cost a 64-bit address each: the 5x5 walk spilled)
    if (tid < kWdQuads) *reinterpret_cast<floatx4*>(Wd + 4 * tid) = wd_next;
    if (ch + 1 < n_chunks) fetch(n0 + kCC);
    __syncthreads();
    // ---- depthwise from LDS
    const int c = n0 + 4 * c4;
    floatx4 psum = {0.f, 0.f, 0.f, 0.f};
    if (c < a.Cexp) {
      floatx4 w[K * K];
#pragma unroll
      for (int t = 0; t < K * K; ++t) w[t] = *reinterpret_cast<const floatx4*>(Wd + t * kCC + 4 * c4);
      const floatx4 bias = *reinterpret_cast<const floatx4*>(Wd + K * K * kCC + 4 * c4);
#pragma unroll 1
      for (int pass = 0; pass < NPASS; ++pass) {
        const int orow = S == 2 ? rgrp : rgrp * 4 + pass * RT;  // first output row (in the tile) of this pass
        floatx4 o[RT];
#pragma unroll
        for (int q = 0; q < RT; ++q) o[q] = bias;
        const int er0 = orow * S, ec0 = col * S;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          floatx4 xv[K];
#pragma unroll
          for (int dx = 0; dx < K; ++dx) xv[dx] = *reinterpret_cast<const floatx4*>(Es + ((er0 + r) * T::IW + ec0 + dx) * kLDE + 4 * c4);
#pragma unroll
          for (int q = 0; q < RT; ++q) {
            const int dy = r - q * S;
            if (dy >= 0 && dy < K) {
#pragma unroll
              for (int dx = 0; dx < K; ++dx)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[q][e] = fmaf(xv[dx][e], w[dy * K + dx][e], o[q][e]);
            }
          }
        }
        const int ow = ow0 + col;
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int oh = oh0 + orow + q;
          if (oh < a.Ho && ow < a.Wo) {
            const floatx4 v = swish4(o[q]);
            *reinterpret_cast<floatx4*>(a.y + (((int64_t)img * a.Ho + oh) * a.Wo + ow) * a.Cexp + c) = v;
            psum += v;
          }
        }
      }
    }
    red[tid] = psum;
    __syncthreads();  // E is free again; the pool partials are complete
    if (tid < 8 && n0 + 4 * tid < a.Cexp) {
      floatx4 s = red[tid];
      for (int j = 1; j < 32; ++j) s += red[tid + 8 * j];
      const int tile = ty * gridDim.x + tx;
      *reinterpret_cast<floatx4*>(a.pool_partial + ((int64_t)img * gridDim.x * gridDim.y + tile) * a.Cexp + n0 + 4 * tid) = s;
    }
  }
  if (a.status && !(fabsf(chk) <= 3.0e38f)) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int K, int S, int KP>
int launch_front_t(const FrontArgs& a, hipStream_t stream) {
  using T = FrontTile<K, S>;
  static FirstLaunch fl;
  constexpr size_t lds = front_lds_bytes<K, S, KP>();
  static_assert(lds <= 160 * 1024, "LDS of a CU");
  if (const int rc0 = fl.once([](FirstLaunch& fl_) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_front_kernel<K, S, KP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fl_.spills = note_kernel(reinterpret_cast<const void*>(&mbconv_front_kernel<K, S, KP>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  const dim3 grid((unsigned)((a.Wo + T::TW - 1) / T::TW), (unsigned)((a.Ho + T::TH - 1) / T::TH), (unsigned)a.n);
  hipLaunchKernelGGL((mbconv_front_kernel<K, S, KP>), grid, dim3(256), lds, stream, a);
  return check_launch("mbconv_front_kernel");
}

}  // namespace

bool mbconv_front_applicable(int cin, int kpad, int cexp, int k, int stride) {
  // not the 5x5 / stride-1 blocks: 25 taps x 4 channels in registers beside a sliding window of E rows spills (60 - 67 VGPRs, 100 -
  // 160 B of scratch per lane), and the launch (642 us) loses to expansion + strip kernel (~230 us): not instantiated
  if (k == 5 && stride == 1) return false;
  return !dbg(DBG_NO_MBCONV_FRONT) && (kpad == 32 || kpad == 64) && cin % 4 == 0 && cin <= kpad && cexp % 4 == 0 && (k == 3 || k == 5) &&
         (stride == 1 || stride == 2);
}

// pooling partials per image the launch writes (tiles of D)
int mbconv_front_tiles(int Ho, int Wo, int stride) {
  const int th = stride == 1 ? 8 : 4, tw = stride == 1 ? 16 : 8;
  return ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
}

int launch_mbconv_front(const FrontArgs& a, hipStream_t stream) {
  if (a.n >= 65536) return fail(HP_ERR_ARG, "mbconv_front: batch too large");
#define HP_FRONT(K_, S_)                                                                    \
  if (a.k == K_ && a.stride == S_)                                                          \
    return a.Kpad == 32 ? launch_front_t<K_, S_, 32>(a, stream) : launch_front_t<K_, S_, 64>(a, stream);
  HP_FRONT(3, 1) HP_FRONT(3, 2) HP_FRONT(5, 2)
#undef HP_FRONT
  return fail(HP_ERR_ARG, "mbconv_front: unsupported depthwise geometry");
}

}  // namespace hp
