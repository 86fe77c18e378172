// Argument blocks and launchers of the conv-stack kernels (conv.hip), used by net.cpp.
#pragma once

#include <cstdlib>

#include "common.h"

namespace hp {

// activation applied by the conv epilogues (ConvArgs::relu)
enum { HP_ACT_NONE = 0, HP_ACT_RELU = 1, HP_ACT_SWISH = 2 };

// Division by a launch-time constant as multiply-high + shift: n / d = (umulhi(n, m) + n) >> l for
// 0 <= n < 2^31, m = floor(2^32 (2^l - d) / d) + 1, l = ceil(log2 d).  A runtime integer division costs
// ~40 vector instructions (a 64-bit one ~200), and the tile decodes at the head of every workgroup need
// several per thread: a quarter of the 5x5 stem's time before they were replaced.
struct FastDiv { unsigned m, l; };
inline FastDiv make_fastdiv(unsigned d) {
  if (d <= 1) return FastDiv{0u, 0u};
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  return FastDiv{(unsigned)((((1ull << l) - d) << 32) / d + 1), l};
}
#ifdef __HIPCC__
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)((__umulhi((unsigned)n, f.m) + (unsigned)n) >> f.l); }
#endif

struct ConvArgs {
  const float* x;          // NHWC [n][H][W][Cin]
  const float* w;          // packed [Cout][Kpad]
  const float* bias;       // [Cout] or null
  const float* residual;   // NHWC [n][Ho][Wo][Cout] or null
  const float* pre_scale;  // [Cin] or null  (x_act = relu(x * scale + shift)); with pre_shift == null:
                           // [n_img][Cin] squeeze-excitation gate, x_act = x * scale[img] (generic kernel only)
  const float* pre_shift;
  const int4* lut;         // [Kpad/4] {offset, kh, kw, channel}; kh < 0 marks K padding
  float* y;                // NHWC [n][Ho][Wo][Cout]
  int64_t M;               // n * Ho * Wo
  int H, W, Cin, Ho, Wo, Cout, stride, pad, Kpad, ktiles, relu;
  int tiles_m, tiles_n;    // filled by launch_conv
  FastDiv fd_howo, fd_wo, fd_tn;  // by Ho*Wo, Wo, tiles_n (filled by the launchers; M < 2^31)
  // tail split-K of the patch kernel (filled by launch_conv_patch)
  int sk_regular, sk_S, sk_tail_items;
  int sk_S2, sk_S3;        // pooled-stem launch (conv_igemm_split.hip): tiles per image / per tile row
  float* sk_slabs;
  int* sk_counters;
  // per-launch switches (zero-initialised = defaults), carried here instead of process-wide state so that
  // networks on different host threads / streams cannot race:
  int algo;                // HP_CONV_ALGO_* of the launching network (only the Winograd schedule reads it here)
  // conv3x3s2_pp only: the block's 1x1 / stride-2 shortcut as EXTRA WORK ITEMS of the 3x3 / stride-2 launch (same input, same
  // output shape; its input pixel is the 3x3's centre tap).  sc_w = the shortcut's weights in the centre of a zero 3x3 filter,
  // split by conv_split_transform_weights(..., stride 2); null = no shortcut items.
  const float* sc_w;
  const float* sc_bias;
  float* sc_y;
  unsigned* sc_amax_out;
  int sc_relu, sc_items;   // sc_items: filled by the launcher (tiles_m x tiles_n)
  int no_tail_split;       // 1: leave the tiles of a partially filled last round whole (a second lane fills the CUs)
  unsigned* status;        // host-visible word or null: the epilogue stores 1 when it produced a non-finite value
                           // (split-fp16 launches: an activation beyond the fp16 range turned into inf / NaN)
  // Dynamic range of the split-fp16 scheme (x = x_hi + x_lo in fp16: |x| must stay below 65504, and x_lo falls into the
  // fp16 subnormals -- an ABSOLUTE floor of 2^-25 -- once |x| < ~0.1).  Every launch that goes through the shared epilogue
  // leaves max|y| of what it stored in *amax_out (bits of a non-negative float, atomicMax; the network zeroes its words at
  // the start of a forward); a split-fp16 consumer of that tensor reads the producer's word, bounds its staged input by
  // amax_a * max|x| + amax_b (the BN + ReLU prologue: max|scale|, max|shift|; 1, 0 without one) and multiplies what it
  // splits by the power of two that puts that bound at 2^13 -- exact, undone with the weights' scale in the epilogue.
  // amax_in == nullptr: no scaling (the first layer, producers that do not track their range).
  const unsigned* amax_in;
  float amax_a, amax_b;
  unsigned* amax_out;
};

constexpr int kAmaxSlots = 8, kAmaxStride = 64;  // words per op: 8 slots, 64 words (256 B) apart
#ifdef __HIPCC__
// the activation scale of a split-fp16 launch (see ConvArgs::amax_in): sx = 2^s with bound * 2^s in [2^12, 2^13), and 1 / sx
__device__ __forceinline__ void conv_act_scale(const ConvArgs& a, float& sx, float& inv_sx) {
  sx = 1.f; inv_sx = 1.f;
  if (a.amax_in) {
    unsigned m = 0u;  // the producer spreads its atomics over kAmaxSlots words 256 B apart (one L2 channel each)
#pragma unroll
    for (int k = 0; k < kAmaxSlots; ++k) m = max(m, a.amax_in[k * kAmaxStride]);
    const float bound = fmaf(a.amax_a, __uint_as_float(m), a.amax_b);
    if (bound > 1e-30f && bound < 1e30f) {
      const int e = (int)((__float_as_uint(bound) >> 23) & 255u) - 126;  // bound = m 2^e, m in [0.5, 1)
      const int sh = 13 - e;
      sx = __uint_as_float((unsigned)(127 + sh) << 23);
      inv_sx = __uint_as_float((unsigned)(127 - sh) << 23);
    }
  }
}
#endif

// fp16 path (conv_f16.hip): activations / weights / prologue vectors are halves, bias is fp32
struct ConvArgsH {
  const _Float16* x;          // NHWC [n][H][W][Cin], Cin % 8 == 0
  const _Float16* w;          // packed [Cout][Kpad], Kpad % 64 == 0
  const float* bias;          // [Cout] or null
  const _Float16* residual;   // NHWC [n][Ho][Wo][Cout] or null
  const _Float16* pre_scale;  // [Cin] or null  (x_act = relu(x * scale + shift), packed fp16)
  const _Float16* pre_shift;
  const int4* lut;            // [Kpad/8] {offset (halves), kh, kw, channel}; kh < 0 marks K padding
  _Float16* y;
  int64_t M, x_bytes, w_bytes;
  int H, W, Cin, Ho, Wo, Cout, stride, pad, Kpad, ktiles, relu;
  int kh, kw;                 // filter size (selects the patch-staged 3x3 kernel)
  int tiles_m, tiles_n;       // filled by the launcher
  FastDiv fd_howo, fd_wo, fd_tn;  // by Ho*Wo, Wo, tiles_n (filled by the launcher; M < 2^31)
  int no_tail_split;          // as ConvArgs::no_tail_split
};

// depthwise k x k conv + folded BN + swish (mbconv.hip)
struct DwArgs {
  const float* x; const float* w /* [k*k][C], BN scale folded */; const float* bias /* [C] BN shift */; float* y;
  int n, H, W, C, Ho, Wo, k, stride, pad_t, pad_l;
  float* pool_partial;  // or null: [n][dwconv_pool_strips(Ho)][C] sums of the outputs (squeeze-excitation pooling)
};

// expansion 1x1 + BN + swish + depthwise k x k + BN + swish in one launch (mbconv_front.hip)
struct FrontArgs {
  const float* x;          // block input NHWC [n][H][W][Cin]
  const void* w_split;     // the expansion's weights as conv_igemm_split_transform_weights leaves them (rows_pad x Kpad)
  const float* bias_e;     // [Cexp] folded BN shift of the expansion
  const float* w_dw;       // [k*k][Cexp] depthwise weights, BN scale folded
  const float* bias_d;     // [Cexp]
  float* y;                // depthwise output NHWC [n][Ho][Wo][Cexp]
  float* pool_partial;     // [n][mbconv_front_tiles][Cexp] sums of y (squeeze-excitation pooling partials)
  unsigned* status;        // non-finite guard word or null
  int n, H, W, Cin, Cexp, Ho, Wo, k, stride, pad_t, pad_l, Kpad, rows_pad;
};

struct HeadArgs {
  int x_is_half;   // features are fp16 (fp16 plan) instead of fp32
  const void* x;   // NHWC [b][HW][C]
  int HW, C;
  const float* fc_w; const float* fc_b;        // optional 512x512 fc (torchvision ResNet)
  const float* pose_w; const float* pose_b; int pose_dim;
  const float* logit_w; const float* logit_b; int n_logits;
  float* pose_out; float* logit_out; float* features;
  float* ws_pool; float* ws_fc;   // [batch][C] workspaces of the fc path (three launches)
};

// variant 0: 128x128 block tile, variant 1: 128x64 (Cout == 64)
int launch_conv(const ConvArgs& a, int variant, hipStream_t stream);
// 3x3 / stride 1 / pad 1 / Cin % 32 == 0: input staged once per channel chunk (conv_patch.hip)
bool conv_patch_applicable(const ConvArgs& a, int kh, int kw);
int launch_conv_patch(const ConvArgs& a, int variant, hipStream_t stream);
// 3x3 / stride 1 / pad 1 / Cin % 16 == 0 / Cout % 32 == 0: Winograd F(2x2,3x3) (conv_wino.hip);
// a.w must point at weights transformed by conv_wino_transform_weights
bool conv_wino_applicable(const ConvArgs& a, int kh, int kw);
bool conv_wino_launchable(const ConvArgs& a);  // per launch: batch-dependent limits
size_t conv_wino_weight_floats(int cout, int cin);
int conv_wino_transform_weights(const float* d_w, float* d_U, int cout, int cin, int Kpad, hipStream_t stream);
int launch_conv_wino(const ConvArgs& a, hipStream_t stream);
// 3x3 / pad 1, stride 1 (Cin % 32 == 0, Cout % 64 == 0) or stride 2 (Cin % 64 == 0, Cout % 128 == 0): fp32 operands split into fp16 halves, three fp16
// MFMAs per product (conv_split.hip); a.w must point at weights split by conv_split_transform_weights
bool conv_split_applicable(const ConvArgs& a, int kh, int kw);
bool conv_split_launchable(const ConvArgs& a);  // per launch: batch-dependent limits
size_t conv_split_weight_bytes(int cout, int cin);
int conv_split_transform_weights(const float* d_w, void* d_ws, int cout, int cin, int Kpad, int stride, hipStream_t stream);
int launch_conv_split(const ConvArgs& a, hipStream_t stream);
// tail split-K planning of the one-workgroup-per-CU 3x3 kernels (conv_split.hip): T tiles, ncc splittable K units
int conv_split_plan_tail(ConvArgs& a, int T, int ncc, size_t tile_floats, int wg_per_cu, hipStream_t stream);
// ping-pong 3x3 / stride-1 kernel for Cout % 128 == 0 (conv_pp.hip): the two waves of a SIMD alternate between an
// LDS / staging segment and an MFMA burst; fp32 (split-fp16 weights of conv_split_transform_weights) and fp16 modes
bool conv_pp_split_applicable(const ConvArgs& a, int kh, int kw);
// the stride-2 3x3 layers on the same skeleton (conv3x3s2_pp): double-buffered space-to-depth patches, persistent items
bool conv_pp_s2_applicable(const ConvArgs& a, int kh, int kw);
int launch_conv_pp_s2_split(const ConvArgs& a, hipStream_t stream);
int launch_conv_pp_split(const ConvArgs& a, hipStream_t stream);
struct ConvArgsH;
bool conv_pp_f16_applicable(const ConvArgsH& a);
int launch_conv_pp_f16(const ConvArgsH& a, hipStream_t stream);
// the same scheme for every other layer the generic kernel runs (stems, 1x1, odd 3x3): conv_igemm_split.hip
size_t conv_igemm_split_weight_bytes(int rows_pad, int Kpad);
int conv_igemm_split_transform_weights(const float* d_w, void* d_ws, int rows_pad, int Kpad, hipStream_t stream);
bool conv_igemm_split_launchable(const ConvArgs& a);
int launch_conv_igemm_split(const ConvArgs& a, int variant, hipStream_t stream);
// the CosyPose stem (5x5 s2 on 6 channels + ReLU + max-pool) with the input region of a tile staged once (conv_stem_split.hip)
bool conv_stem_split_applicable(const ConvArgs& a, int kh, int kw, int run_mode);
int launch_conv_stem_split_pool(ConvArgs args, hipStream_t stream);
// the MegaPose stems (7x7 s2 p3 -> 64, + ReLU + 3x3 s2 max-pool) with the input region of a pooled tile staged once per
// channel slab (conv_stem7.hip); fp32 (split-fp16 arithmetic) or the fp16 plan; weights packed on the host at plan time
int conv_num_cus();  // compute units of the current device (conv_pp.hip)
bool conv_stem7_applicable(int kh, int kw, int stride, int pad, int cin_mem, int cout, int relu, int f16);
size_t conv_stem7_pack_weights(const float* h_w, int cin_real, int cin_mem, int f16, void* h_out);
int launch_conv_stem7_pool(const ConvArgs& a, int f16, hipStream_t stream);
int conv_stem7_f16_krow(int cin_real);
bool conv_igemm_split_pool_launchable(const ConvArgs& a, int cout_pad);
int launch_conv_igemm_split_pool(const ConvArgs& a, hipStream_t stream);
// plan-time choice between the split-fp16 kernel and the exact-fp32 ones for a 3x3 stride-1 layer
inline bool conv_use_split(int algo, int H, int W, int cin, int cout) {
  (void)H; (void)W; (void)cin; (void)cout;  // measured faster than Winograd on every WideResNet / ResNet-34 layer shape
  return algo == HP_CONV_ALGO_SPLIT || algo == HP_CONV_ALGO_AUTO;
}
// which generic-kernel layers move to conv_igemm_split.hip: all of them (1x1 included)
inline bool conv_use_igemm_split(int kh, int Kpad) {
  (void)kh; (void)Kpad;
  return true;
}
// hipGraph safety (hp_scratch_launches): does kernel `fn` use scratch?  (asked once per instantiation, where it opts in to its
// LDS size); every launch of such a kernel is counted
bool note_kernel(const void* fn);
void count_scratch_launch();
int launch_maxpool(const float* x, float* y, int n, int H, int W, int C, int Ho, int Wo, hipStream_t stream);
int launch_zero_words(unsigned* p, int n, hipStream_t stream);  // a kernel, not a memset: memset nodes misbehave under hipGraph replay
// nearest resize to (Ho, Wo) (stride_mode 0) or stride-2 subsampling (stride_mode 1), NHWC, C % 4 == 0
int launch_resize_nearest(const float* x, float* y, int n, int H, int W, int C, int Ho, int Wo, int stride_mode, hipStream_t stream);
// (x - mean) / std per channel, NCHW [n,3,h,w] -> NHWC [n,h,w,4]
int launch_normalize_nhwc4(const float* x, float* y, int n, int h, int w, const float* mean3, const float* std3, hipStream_t stream);
// bilinear resize [n,3,hi,wi] -> [ho,wo] (align_corners = False), (v - mean) / std, into the top-left of a zeroed NHWC4 canvas [n,hp,wp,4]
int launch_normalize_resize_nhwc4(const float* x, float* y, int n, int hi, int wi, int ho, int wo, int hp, int wp, const float* mean3,
                                  const float* std3, hipStream_t stream);
int launch_head(const HeadArgs& a, int batch, hipStream_t stream);
int launch_dwconv(const DwArgs& a, hipStream_t stream);
bool mbconv_front_applicable(int cin, int kpad, int cexp, int k, int stride);
int mbconv_front_tiles(int Ho, int Wo, int stride);
int launch_mbconv_front(const FrontArgs& a, hipStream_t stream);
bool dwconv_pools(const DwArgs& a);   // the launch can also write a.pool_partial
int dwconv_pool_strips(int Ho, int k, int stride);  // partial sums per image it writes
// squeeze-excitation: pooled [n][C] = mean over HW of y; gate [n][C] = sigmoid(W2 swish(W1 pooled + b1) + b2)
int se_partial_floats(int n, int C);  // workspace of launch_se
int launch_se(const float* y, float* partial, float* pooled, float* sq, float* gate, const float* w1, const float* b1,
              const float* w2t /* expand weights transposed to [Cse][C] */, const float* b2, int n, int HW, int C, int Cse,
              int n_partials /* > 0: partial already holds [n][n_partials][C] sums (launch_dwconv), 0: pool y here */,
              hipStream_t stream);
int launch_conv_f16(const ConvArgsH& a, hipStream_t stream);
int launch_cast_pad_f16(const float* x, void* y, int64_t pixels, int c_in, int c_out, hipStream_t stream);
int launch_maxpool_f16(const void* x, void* y, int n, int H, int W, int C, int Ho, int Wo, hipStream_t stream);
int conv_setup_once();
// kernel family of the SINGLE-LAYER entry points hp_conv2d_nhwc / hp_conv2d_nhwc_f16 (hp_conv_select_algo: parity tests and
// tools/conv_fuzz.py walk the families with it).  Networks never read it: a network's choice is hp_net_set_conv_algo, default
// AUTO, handed to the launchers in ConvArgs::algo.
int conv_layer_algo();

}  // namespace hp
