// Generic implicit-GEMM convolution on the split-fp16 scheme of conv_split.hip (fp32 operands as fp16 hi / lo
// halves, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation, per-cout power-of-two weight scaling):
// the layers the patch-staged 3x3 kernels do not cover -- the 5x5 / 7x7 stride-2 stems (CP/models/wide_resnet.py
// conv1, MP/models/torchvision_resnet.py conv1) and 1x1 / odd 3x3 layers.  Same GEMM view, K order, look-up table
// and packed-weight geometry as conv.hip (K-tile = 32 floats = the 8 LUT chunks of 4 floats; the stem's "filter
// row" packing included), so the planner reuses its LUT; only the staged tiles differ: LDS rows are
// [32 hi | 32 lo] halves (144-B pitch), a K-tile is 2 k-steps x 3 MFMAs per 32x32 tile instead of 16 fp32 MFMAs.
// These layers are short in K and long in M (the CosyPose stem: K = 160, 2.4 M output pixels, 629 MB written), so
// the kernel is a plain double-buffered loop at 2-3 workgroups per CU and ends up bound by its output stream.
#include <cstdlib>

#include "conv.h"
#include "conv_epilogue.h"
#include "conv_splitk.h"

namespace hp {

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128;
constexpr int LDH = 64 + 8;  // LDS row: 32 hi + 32 lo halves, padded to 36 dwords

// NBUF = 2: double-buffered K loop, two workgroups per CU; NBUF = 1 (short K: the CosyPose stem has 5 K-tiles):
// one LDS buffer and two barriers per K-tile, which lets FOUR workgroups share a CU -- these launches are gather /
// store streams with a dozen MFMAs per K-tile, and occupancy is what hides their latency
template <int BN, int NBUF>
constexpr size_t igs_lds_bytes() {
  const size_t loop = (size_t)NBUF * (BM + BN) * LDH * 2;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  return loop > epi ? loop : epi;
}

// fp32 packed weights [rows][Kpad] (BN folded, K padded to 32) -> [rows][Kpad / 32][32 hi | 32 lo] halves + per-row
// scale-back factors [rows] (fp32) behind them
__global__ __launch_bounds__(256) void split_weights_generic_kernel(const float* w, _Float16* ws, float* unscale, int Kpad) {
  const int o = blockIdx.x;
  const float* row = w + (size_t)o * Kpad;
  __shared__ float red[256];
  float mx = 0.f;
  for (int k = threadIdx.x; k < Kpad; k += 256) mx = fmaxf(mx, fabsf(row[k]));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);
  const int s = mx > 0.f ? 14 - e : 0;  // max |w| 2^s in [2^13, 2^14)
  if (threadIdx.x == 0) unscale[o] = ldexpf(1.f, -s);
  _Float16* out = ws + (size_t)o * Kpad * 2;
  for (int k = threadIdx.x; k < Kpad; k += 256) {
    const float v = ldexpf(row[k], s);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    out[(size_t)(k >> 5) * 64 + (k & 31)] = hi;
    out[(size_t)(k >> 5) * 64 + 32 + (k & 31)] = lo;
  }
}

// POOL: the conv is followed by ReLU and a 3x3 / stride-2 / pad-1 max-pool (the stems: CP/models/wide_resnet.py:104-107,
// MP/models/torchvision_resnet.py:216-219) and the kernel writes the POOLED map.  A workgroup owns 3 x 8 pooled pixels
// of one image and computes the 7 x 17 conv pixels under them (119 of its 128 GEMM rows; 24 % of the conv pixels are
// computed twice, which costs little: the launch is a store stream) -- the 629 MB conv map of a CosyPose stem is never
// written or re-read.  Conv pixels outside the map do not take part in the max (PyTorch pads with -inf).
constexpr int kPoolRows = 3, kPoolCols = 8, kPoolCW = 2 * kPoolCols + 1, kPoolCH = 2 * kPoolRows + 1;

// Direct epilogue (round 5; conv_pp.hip's, for this kernel's tiles).  The MFMAs run TRANSPOSED -- weights as the A operand
// (rows read in the order sigma(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3)), pixels as B -- so a lane holds ONE pixel (l & 31)
// and 16 CONSECUTIVE output channels per 32 x 32 tile: acc[mt][nt][r] = (pixel m0 + wm + 32 mt + (l & 31), channel n0 + wn + 32 nt +
// 16 (l >> 5) + r).  A 4 x 4 transpose inside each lane quad gives lane j piece j (4 channels) of the quad's four pixels: store k
// writes pixel k of every quad, 4 lanes x 16 B = 64 contiguous bytes, the two half-waves complete the 128-B line.  No LDS
// transpose, no barrier: the shared epilogue (conv_epilogue.h: 32 ds_write_b32 + 8 ds_read_b128 per lane between two barriers,
// then 8 stores) was HALF of these launches -- compiled without loads and MFMAs the 120 x 160 / 24-channel projection still took
// 69 of its 141 us, the 7 x 10 / 1392-channel expansion 13.6 of 28.  Same dot products in the same order: bit-identical outputs.
__device__ __forceinline__ void igs_quad_transpose4(int qj, float& r0, float& r1, float& r2, float& r3) {
  {
    const bool b = qj & 1;
    const float s01 = b ? r0 : r1, s23 = b ? r2 : r3;
    const float g01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
    const float g23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
    if (b) { r0 = g01; r2 = g23; } else { r1 = g01; r3 = g23; }
  }
  {
    const bool b = qj & 2;
    const float s02 = b ? r0 : r2, s13 = b ? r1 : r3;
    const float g02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
    const float g13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
    if (b) { r0 = g02; r1 = g13; } else { r2 = g02; r3 = g13; }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void igs_epilogue_direct(const ConvArgs& a, floatx16 (&acc)[MT][NT], const float* unscale, float act_inv, int64_t m0,
                                                    int n0, int wm, int wn, int lane) {
  const int px = lane & 31, h16 = 16 * (lane >> 5), qj = lane & 3, qp = px & ~3;
  float chk = 0.f, amax = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + wn + nt * 32 + h16 + 4 * qj;  // this lane's four channels after the transpose
    const bool n_ok = n < a.Cout;                    // the last tile of a layer whose Cout is not a multiple of the tile width
    const floatx4 sc = *reinterpret_cast<const floatx4*>(unscale + n) * act_inv;  // rows / bias padded to whole tiles by the planner
    floatx4 bias = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias = *reinterpret_cast<const floatx4*>(a.bias + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      floatx4 res[4];
      if (a.residual) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + wm + mt * 32 + qp + k;
          res[k] = *reinterpret_cast<const floatx4*>(a.residual + (m < a.M && n_ok ? m * a.Cout + n : 0));
        }
      }
      floatx4 t[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) t[g] = floatx4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float r0 = t[0][c], r1 = t[1][c], r2 = t[2][c], r3 = t[3][c];
        igs_quad_transpose4(qj, r0, r1, r2, r3);
        t[0][c] = r0; t[1][c] = r1; t[2][c] = r2; t[3][c] = r3;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // t[k] = channels [n, n + 4) of pixel row m0 + wm + 32 mt + qp + k
        const int64_t m = m0 + wm + mt * 32 + qp + k;
        floatx4 v = t[k] * sc + bias;
        if (a.residual) v += res[k];
        if (m < a.M && n_ok) {
          chk += (v[0] + v[1]) + (v[2] + v[3]);
          if (a.relu == HP_ACT_RELU) v = __builtin_elementwise_max(v, floatx4{0.f, 0.f, 0.f, 0.f});
          else if (a.relu == HP_ACT_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.f + __expf(-v[q]));
          }
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          *reinterpret_cast<floatx4*>(a.y + m * a.Cout + n) = v;
        }
      }
    }
  }
  conv_report_nonfinite(a, chk);
  if (a.amax_out) {  // as conv_epilogue.h: wave maximum, look before the atomic, words spread over L2 channels
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0 && amax > 0.f) {
      const unsigned mine = __float_as_uint(amax);
      unsigned* const slot = a.amax_out + (blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride;
      if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
    }
  }
}

// PRE: 0 none, 1 BN + ReLU on the input (pre_scale / pre_shift [Cin]), 2 squeeze-excitation gate (pre_scale [n][Cin],
// EfficientNet projections: the gated input is formed in fp32 and then split, as the exact kernels form it)
// Register budget: double-buffered (NBUF 2) two workgroups per CU = 256 VGPRs; single-buffered (NBUF 1) four per CU = 128
// VGPRs for the 128-wide tile's cousins that fit, but THREE per CU (168 VGPRs) for the 64-wide tile: at 128 it spilled
// 36-220 B per lane to scratch (EfficientNet's narrow 1x1 layers, the detector), and a launch that uses scratch also keeps
// hipGraph replay off for the whole network.
// LIN: a 1x1 / stride-1 / pad-0 layer (EfficientNet's expansions and projections): K-tile t, chunk kc IS channel 32 t + 4 kc of
// the row's own pixel -- no look-up table.  The table entry is a global load that every activation load of a K-tile depends
// on: in program order the wave waits for it (and with it for everything issued before) at the head of every K-tile.
template <int BN, int PRE, int NBUF, bool POOL, bool LIN = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NBUF == 2 ? 2 : 3, NBUF == 2 ? 2 : 4))) void conv_igemm_split_f32(ConvArgs a) {
  constexpr int MT = 2, NT = BN / 64;  // 4 waves 2 x 2, wave tile 64 x BN/2
  constexpr int NA = 4, NB = BN / 32;  // staged 16-B pieces per thread and K-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* const As = reinterpret_cast<_Float16*>(lds_raw);  // [2][BM][LDH]
  _Float16* const Bs = As + NBUF * BM * LDH;                    // [NBUF][BN][LDH]

  int lin, slice;
  bool split;
  if (!splitk_decode(a, lin, slice, split)) return;
  const int tile_m = fdiv(lin, a.fd_tn), tile_n = lin - tile_m * a.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;
  // POOL: tile_m -> (image, pooled tile row, pooled tile column); fd_howo / fd_wo divide by tiles per image / per row
  int p_img = 0, p_oh0 = 0, p_ow0 = 0;
  if (POOL) {
    if (lin >= a.tiles_m * a.tiles_n) return;  // grid rounded up to the 8 XCDs
    p_img = fdiv(tile_m, a.fd_howo);
    const int rem = tile_m - p_img * (int)a.sk_S2;
    const int ty = fdiv(rem, a.fd_wo), tx = rem - ty * (int)a.sk_S3;
    p_oh0 = 2 * kPoolRows * ty - 1;
    p_ow0 = 2 * kPoolCols * tx - 1;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kc = tid & 7, r0 = tid >> 3;

  // per staged row (output pixel): input origin and offset
  const float* xrow[NA];
  int ih0[NA], iw0[NA];
  int gate_off[NA];  // PRE == 2: image index * Cin of each staged row
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = m0 + r0 + 32 * i;
    bool live = m < a.M;
    int img = 0, oh = 0, ow = 0;
    if (POOL) {
      const int r = r0 + 32 * i, dr = r / kPoolCW;
      img = p_img; oh = p_oh0 + dr; ow = p_ow0 + (r - dr * kPoolCW);
      live = r < kPoolCH * kPoolCW && (unsigned)oh < (unsigned)a.Ho && (unsigned)ow < (unsigned)a.Wo;
    } else if (live) {
      img = fdiv((int)m, a.fd_howo);
      const int rem = (int)m - img * HoWo;
      oh = fdiv(rem, a.fd_wo); ow = rem - oh * a.Wo;
    }
    gate_off[i] = live ? img * a.Cin : 0;
    if (live) {
      ih0[i] = oh * a.stride - a.pad;
      iw0[i] = ow * a.stride - a.pad;
      xrow[i] = a.x + (((int64_t)img * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
    } else {
      ih0[i] = -(1 << 28); iw0[i] = 0; xrow[i] = a.x;
    }
  }
  // LIN: the staged pixel of GEMM row m IS input pixel m: 32-bit byte offsets into a buffer resource over x, one per staged row;
  // rows beyond M and channels beyond Cin read out of bounds = zeros -- no bounds arithmetic, no 64-bit addresses, no masks
  constexpr unsigned kOob = 0xFFFFFFF0u;
  unsigned voff[NA];
  unsigned live_mask = 0;
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, LIN ? (int)(unsigned)(a.M * a.Cin * 4) : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = m0 + r0 + 32 * i;
    voff[i] = LIN && m < a.M ? (unsigned)((m * a.Cin + 4 * kc) * 4) : kOob;
    live_mask |= (m < a.M ? 1u : 0u) << i;
  }
  const _Float16* const wsplit = reinterpret_cast<const _Float16*>(a.w);
  float act_sx, act_inv;  // ConvArgs::amax_in: power-of-two scale of the staged activations and its inverse
  conv_act_scale(a, act_sx, act_inv);
  const _Float16* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wrow[i] = wsplit + (int64_t)(n0 + r0 + 32 * i) * a.Kpad * 2 + 8 * kc;

  _Float16* const Ast = As + r0 * LDH + 4 * kc;
  _Float16* const Bst = Bs + r0 * LDH + 8 * kc;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (BN / 2);
  const int frow = lane & 31, fk = 8 * (lane >> 5);
  const _Float16* const Afr = As + (wm + frow) * LDH + fk;
  // MFMA row i of a 32-channel block multiplies weight row sigma(i): the lane's 16 accumulator rows are 16 consecutive channels
  const int srow = 16 * ((frow >> 2) & 1) + 4 * (frow >> 3) + (frow & 3);
  const _Float16* const Bfr = Bs + (wn + srow) * LDH + fk;

  // staged K-tile in registers; two sets, so that the loads of K-tile t+2 are in flight while t is multiplied and
  // t+1 is written to LDS (these launches are latency-bound: a dozen MFMAs per K-tile and wave)
  struct Stage { floatx4 ra[NA]; halfx8 rbw[NB]; floatx4 ps, pb; floatx4 g[PRE == 2 ? NA : 1]; unsigned ok; };
  auto issue = [&](int t, Stage& st) {
    int4 e;  // {offset, kh, kw, channel}; kh < 0 marks K padding
    if (LIN) { const int c = 32 * t + 4 * kc; e = make_int4(c, c < a.Cin ? 0 : -1, 0, c); }
    else e = a.lut[t * 8 + kc];
    st.ok = 0;
    if constexpr (LIN) {
      const bool cok = e.y >= 0;
      st.ok = cok ? live_mask : 0u;  // (only the BN + ReLU prologue needs it: relu(0 * s + b) is not zero)
#pragma unroll
      for (int i = 0; i < NA; ++i)
        st.ra[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(cok ? voff[i] : kOob), t * 128, 0));
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int ih = ih0[i] + e.y, iw = iw0[i] + e.z;
        const bool in = (e.y >= 0) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        st.ra[i] = *reinterpret_cast<const floatx4*>(in ? xrow[i] + e.x : a.x);
        st.ok |= (in ? 1u : 0u) << i;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) st.rbw[i] = *reinterpret_cast<const halfx8*>(wrow[i] + (size_t)t * 64);
    if (PRE == 1) {
      const int c = e.y >= 0 ? e.w : 0;
      st.ps = *reinterpret_cast<const floatx4*>(a.pre_scale + c);
      st.pb = *reinterpret_cast<const floatx4*>(a.pre_shift + c);
    }
    if (PRE == 2) {
      const int c = e.y >= 0 ? e.w : 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) st.g[PRE == 2 ? i : 0] = *reinterpret_cast<const floatx4*>(a.pre_scale + gate_off[i] + c);
    }
  };
  auto store = [&](int buf, const Stage& st) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      floatx4 v = st.ra[i];
      if (PRE == 1) v = __builtin_elementwise_max(v * st.ps + st.pb, floatx4{0.f, 0.f, 0.f, 0.f});
      if (PRE == 2) v = v * st.g[PRE == 2 ? i : 0];
      v = v * act_sx;  // ConvArgs::amax_in: exact power of two (1 when the input's range is not tracked)
      if (!(LIN && PRE != 1) && !((st.ok >> i) & 1u)) v = floatx4{0.f, 0.f, 0.f, 0.f};  // (LIN: the load itself returned zeros)
      const halfx4 hi = __builtin_convertvector(v, halfx4);
      const halfx4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), halfx4);
      *reinterpret_cast<halfx4*>(Ast + buf * BM * LDH + 32 * i * LDH) = hi;
      *reinterpret_cast<halfx4*>(Ast + buf * BM * LDH + 32 * i * LDH + 32) = lo;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<halfx8*>(Bst + buf * BN * LDH + 32 * i * LDH) = st.rbw[i];
  };
  auto compute = [&](int buf) {
    const _Float16* Ab = Afr + buf * BM * LDH;
    const _Float16* Bb = Bfr + buf * BN * LDH;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      halfx8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ah[i] = *reinterpret_cast<const halfx8*>(Ab + i * 32 * LDH + kk * 16);
        al[i] = *reinterpret_cast<const halfx8*>(Ab + i * 32 * LDH + 32 + kk * 16);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        bh[i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH + kk * 16);
        bl[i] = *reinterpret_cast<const halfx8*>(Bb + i * 32 * LDH + 32 + kk * 16);
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], ah[mi], acc[mi][ni], 0, 0, 0);  // D[channel][pixel]
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], al[mi], acc[mi][ni], 0, 0, 0);
        }
    }
  };

  const int t_begin = split ? slice * a.ktiles / a.sk_S : 0;
  const int t_end = split ? (slice + 1) * a.ktiles / a.sk_S : a.ktiles;
  Stage s0, s1;
  issue(t_begin, s0);
  if (t_begin + 1 < t_end) issue(t_begin + 1, s1);
  store(0, s0);
  __syncthreads();
  // one K-tile: `cur` was written to LDS buffer `buf` (from the OTHER set), `nxt` holds t + 1 and takes t + 2 after it is stored
  // (a third register stage -- the loads of t + 3 in flight -- was measured on the 1x1 layers: no change, round 5)
  auto ktile = [&](int t, int buf, Stage& stored, Stage& nxt) {
    if (t + 2 < t_end) issue(t + 2, stored);  // its registers are free: tile t went to LDS before the last barrier
    compute(buf);
    if (NBUF == 1) __syncthreads();  // every wave is done reading the only buffer
    if (t + 1 < t_end) store(NBUF == 2 ? buf ^ 1 : 0, nxt);
    __syncthreads();
  };
  for (int t = t_begin; t < t_end; t += 2) {
    ktile(t, 0, s0, s1);
    if (t + 1 < t_end) ktile(t + 1, NBUF == 2 ? 1 : 0, s1, s0);
  }

  if (split && !splitk_reduce<BM, BN, MT, NT, kThreads>(a, acc, lin - a.sk_regular, slice)) return;

  // scale back + bias + residual + activation + store, straight from the accumulators
  const int rows_pad = a.tiles_n * BN;
  const float* const unscale = reinterpret_cast<const float*>(wsplit + (size_t)rows_pad * a.Kpad * 2);
  if (!POOL) {
    igs_epilogue_direct<MT, NT>(a, acc, unscale, act_inv, m0, n0, wm, wn, lane);
    return;
  }
  // pooled stem: scale back (a lane holds 16 consecutive channels of one conv pixel per tile)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const floatx4 s4 = *reinterpret_cast<const floatx4*>(unscale + n0 + wn + nt * 32 + 16 * (lane >> 5) + 4 * g) * act_inv;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[mt][nt][4 * g + q] *= s4[q];
    }
  // ---- pooled epilogue: conv tile -> LDS [row][BN + 4], then max over the 3 x 3 windows (bias and ReLU commute with max)
  float* const cl = reinterpret_cast<float*>(lds_raw);
  constexpr int LDC = BN + 4;
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g)  // row = the lane's conv pixel, 16 consecutive channels as four 16-B pieces
        *reinterpret_cast<floatx4*>(cl + (wm + mt * 32 + (lane & 31)) * LDC + wn + nt * 32 + 16 * (lane >> 5) + 4 * g) =
            floatx4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
  __syncthreads();
  constexpr int C4 = BN / 4;
  const int Hp = (a.Ho - 1) / 2 + 1, Wp = (a.Wo - 1) / 2 + 1;
  float pool_chk = 0.f;
  for (int it = tid; it < kPoolRows * kPoolCols * C4; it += kThreads) {
    const int c4 = it % C4, pp = it / C4, py = pp / kPoolCols, px = pp - py * kPoolCols;
    const int ph = (p_oh0 + 1) / 2 + py, pw = (p_ow0 + 1) / 2 + px;
    const int n = n0 + 4 * c4;
    if (ph >= Hp || pw >= Wp || n >= a.Cout) continue;
    floatx4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    floatx4 seen = {0.f, 0.f, 0.f, 0.f};  // v_max drops a NaN operand: the non-finite guard sums what the window reads
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int dr = 2 * py + dy, dc = 2 * px + dx;
        if ((unsigned)(p_oh0 + dr) < (unsigned)a.Ho && (unsigned)(p_ow0 + dc) < (unsigned)a.Wo)
        {
          const floatx4 cv = *reinterpret_cast<const floatx4*>(cl + (dr * kPoolCW + dc) * LDC + 4 * c4);
          best = __builtin_elementwise_max(best, cv);
          seen += cv;
        }
      }
    if (a.bias) best += *reinterpret_cast<const floatx4*>(a.bias + n);
    best = __builtin_elementwise_max(best, floatx4{0.f, 0.f, 0.f, 0.f});
    *reinterpret_cast<floatx4*>(a.y + (((int64_t)p_img * Hp + ph) * Wp + pw) * a.Cout + n) = best;
    pool_chk += (seen[0] + seen[1]) + (seen[2] + seen[3]);
  }
  conv_report_nonfinite(a, pool_chk);
}

template <int BN, int PRE, int NBUF>
int launch_igs_pool(ConvArgs args, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch& fl_) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_f32<BN, PRE, NBUF, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)igs_lds_bytes<BN, NBUF>()));
        fl_.spills = note_kernel(reinterpret_cast<const void*>(&conv_igemm_split_f32<BN, PRE, NBUF, true>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  const int Hp = (args.Ho - 1) / 2 + 1, Wp = (args.Wo - 1) / 2 + 1;
  const int tiles_y = (Hp + kPoolRows - 1) / kPoolRows, tiles_x = (Wp + kPoolCols - 1) / kPoolCols;
  const int n_img = (int)(args.M / ((int64_t)args.Ho * args.Wo));
  args.tiles_m = n_img * tiles_y * tiles_x;
  args.tiles_n = (args.Cout + BN - 1) / BN;
  args.fd_howo = make_fastdiv((unsigned)(tiles_y * tiles_x));  // tiles per image
  args.fd_wo = make_fastdiv((unsigned)tiles_x);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  args.sk_S2 = tiles_y * tiles_x;
  args.sk_S3 = tiles_x;
  const int nblk = args.tiles_m * args.tiles_n;
  args.sk_regular = (nblk + 7) / 8 * 8; args.sk_S = 1; args.sk_tail_items = 0; args.sk_slabs = nullptr; args.sk_counters = nullptr;
  args.M = (int64_t)args.tiles_m * BM;  // every GEMM row of a tile is addressed through the tile, not through M
  const size_t lds = igs_lds_bytes<BN, NBUF>();
  hipLaunchKernelGGL((conv_igemm_split_f32<BN, PRE, NBUF, true>), dim3(args.sk_regular), dim3(kThreads), lds, stream, args);
  return check_launch("conv_igemm_split_f32<pool>");
}

template <int BN, int PRE, int NBUF, bool LIN = false>
int launch_igs(ConvArgs args, hipStream_t stream) {
  static FirstLaunch fl;
  if (!LIN) {
    // a 1x1 / stride-1 / pad-0 layer (same map size without padding, K = the channels): the table-free instantiation
    // (not the single-buffered tile with a BN + ReLU prologue: its table-free instantiation spills 28 B per lane at 168 VGPRs)
    if (!(PRE == 1 && NBUF == 1) && args.pad == 0 && args.stride == 1 && args.Ho == args.H && args.Wo == args.W &&
        args.Kpad == (args.Cin + 31) / 32 * 32 && args.Cin % 4 == 0 && args.M * args.Cin * 4 < 0xFFFFFFF0ll)
      return launch_igs<BN, PRE, NBUF, true>(args, stream);
  }
  if (const int rc0 = fl.once([](FirstLaunch& fl_) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_f32<BN, PRE, NBUF, false, LIN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)igs_lds_bytes<BN, NBUF>()));
        fl_.spills = note_kernel(reinterpret_cast<const void*>(&conv_igemm_split_f32<BN, PRE, NBUF, false, LIN>));
        return HP_OK;
      }))
    return rc0;
  if (fl.spills) count_scratch_launch();
  args.tiles_m = (int)((args.M + BM - 1) / BM);
  args.tiles_n = (args.Cout + BN - 1) / BN;
  if (args.M >= (1ll << 31)) return fail(HP_ERR_ARG, "conv: more than 2^31 output pixels");
  args.fd_howo = make_fastdiv((unsigned)(args.Ho * args.Wo));
  args.fd_wo = make_fastdiv((unsigned)args.Wo);
  args.fd_tn = make_fastdiv((unsigned)args.tiles_n);
  const int nblk = args.tiles_m * args.tiles_n;
  int rc = conv_plan_split(args, nblk, igs_lds_bytes<BN, NBUF>(), args.ktiles, 1, stream);
  if (rc) return rc;
  const int per_xcd = args.sk_regular / 8 + (args.sk_tail_items + 7) / 8;
  const size_t lds = igs_lds_bytes<BN, NBUF>();
  hipLaunchKernelGGL((conv_igemm_split_f32<BN, PRE, NBUF, false, LIN>), dim3(8 * per_xcd), dim3(kThreads), lds, stream, args);
  return check_launch("conv_igemm_split_f32");
}

}  // namespace

// rows = Cout padded to whole tiles (64, or 128 when a multiple of 128)
size_t conv_igemm_split_weight_bytes(int rows_pad, int Kpad) { return (size_t)rows_pad * Kpad * 2 * 2 + (size_t)rows_pad * 4; }

int conv_igemm_split_transform_weights(const float* d_w, void* d_ws, int rows_pad, int Kpad, hipStream_t stream) {
  _Float16* ws = reinterpret_cast<_Float16*>(d_ws);
  float* unscale = reinterpret_cast<float*>(ws + (size_t)rows_pad * Kpad * 2);
  hipLaunchKernelGGL(split_weights_generic_kernel, dim3(rows_pad), dim3(256), 0, stream, d_w, ws, unscale, Kpad);
  return check_launch("split_weights_generic_kernel");
}

// a.w = weights split by conv_igemm_split_transform_weights over cout_pad rows (variant 0: cout_pad % 128 == 0 -> 128-wide
// tiles, variant 1: 64-wide); a.pre_scale with a.pre_shift = BN + ReLU prologue, a.pre_scale alone = squeeze-excitation gate
bool conv_igemm_split_launchable(const ConvArgs& a) {
  if (a.pre_scale && !a.pre_shift && a.Cin % 4) return false;
  return a.Kpad % 32 == 0;
}

template <int BN, int NBUF>
static int launch_igs_pre(const ConvArgs& a, hipStream_t stream) {
  if (!a.pre_scale) return launch_igs<BN, 0, NBUF>(a, stream);
  return a.pre_shift ? launch_igs<BN, 1, NBUF>(a, stream) : launch_igs<BN, 2, NBUF>(a, stream);
}

int launch_conv_igemm_split(const ConvArgs& a, int variant, hipStream_t stream) {
  if (variant == 0) return launch_igs_pre<128, 2>(a, stream);
  // short K: single-buffered, three workgroups per CU -- except the squeeze-excitation-gated input (PRE 2), whose gate
  // arithmetic does not fit 168 VGPRs without scratch: those layers take the double-buffered tile
  if (a.ktiles <= 8 && !(a.pre_scale && !a.pre_shift)) return launch_igs_pre<64, 1>(a, stream);
  return launch_igs_pre<64, 2>(a, stream);
}

// conv + ReLU + 3x3 / stride-2 / pad-1 max-pool in one launch (a.y = the pooled map [n][Hp][Wp][Cout]); 64-wide tiles only
// Only for short K (the 6-channel CosyPose stem, 5 K-tiles: stem + pool 481 -> 353 us): the 24 % of recomputed conv
// pixels cost more than the saved traffic on the 7x7 / 32-channel MegaPose stem (49 K-tiles; C3 2440 -> 2312 poses/s).
bool conv_igemm_split_pool_launchable(const ConvArgs& a, int cout_pad) {
  return conv_igemm_split_launchable(a) && cout_pad % 128 != 0 && a.relu == HP_ACT_RELU && !a.residual && !a.pre_scale &&
         a.ktiles <= 8;
}

int launch_conv_igemm_split_pool(const ConvArgs& a, hipStream_t stream) { return launch_igs_pool<64, 0, 1>(a, stream); }

}  // namespace hp
