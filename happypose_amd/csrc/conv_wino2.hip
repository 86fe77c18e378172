// 3x3 / stride-1 / pad-1 convolution as Winograd F(2x2, 3x3) on the fp16 matrix path with fp32-level accuracy
// (split operands: three v_mfma_f32_32x32x16_f16 per product, conv_split.hip), tiled FOR that path.
//
// The first attempt (conv_wino.hip, SPLIT) kept the shape of the fp32 kernel -- a wave = 16 tiles x 32 couts x 8
// positions on 16x16 MFMAs -- and lost to the direct split kernels: at fp16 rates a chunk's matrix time is a third of the
// LDS time of re-reading a 32-KB weight tile per wave for 16 tiles.  Here:
//   item  = 64 output tiles (2x2 pixels each, tile-linear over batch x tile rows x tile columns) x 64 output channels;
//   wave  = (tile half th: 32 tiles) x (position row pg: 4 of the 16 transform positions) x all 64 couts:
//           4 positions x 2 cout blocks of 32 = 8 accumulator tiles of v_mfma_f32_32x32x16_f16 = 128 registers;
//   lane  = (tile t = lane & 31, channel octet o = lane >> 5): it reads the two pixel rows its position row needs
//           (4 columns x 8 channels: 16 ds_read_b128), forms V = (B^T d)_row B for its 8 channels in registers, splits it
//           into fp16 hi / lo -- and that IS the B operand of the MFMA (column = tile lane & 31, k = 8 (lane >> 5) .. + 7);
//   weights U = G g G^T are the A operand (rows = couts): a lane's accumulator elements are then groups of four consecutive
//           couts of ONE tile, stored as float4 without a transpose.  Per chunk (16 input channels) a wave reads 16 KB of
//           weight fragments for 32 tiles x 64 couts x 4 positions: 4x less per output than the first attempt.
//   Probe (tools/probes/wino2_chunk_model.hip, the instruction mix without staging): 2794 cycles per chunk at 1.74 GHz =
//   1.6 us against 3456 cycles of pure MFMA time the direct split kernel needs for the same outputs at 100 % pipe use.
// Output transform Y = A^T M A: every wave reduces its position row to (z0, z1) = (m0 + m1 + m2, m1 - m2 - m3) per
// (tile, cout), the four position rows of a tile half meet in LDS, and wave (th, pg) finishes output pixel
// (pg >> 1, pg & 1) of its 32 tiles: Y[0][.] = z(row0) + z(row1) + z(row2), Y[1][.] = z(row1) - z(row2) - z(row3).
// One workgroup per item, one per CU (up to 148 KB of LDS).  Schedule: two wave groups (position rows 0, 1 / 2, 3; one wave of
// each per SIMD) run the same loop one barrier apart, so that one wave of a SIMD multiplies while the other transforms; a
// group's half of the weight tile is reloaded by LDS-direct loads (global_load_lds_dwordx4: no staging registers -- with
// them the kernel spilled) during the interval in which the group transforms; pixels are staged through registers during
// the multiply interval.
//
// MEASURED (MI355X, WideResNet-34 at 128 x 240 x 320, per launch): correct (tests/test_gpu_kernels.py
// winograd_split) and NOT faster than the direct split kernels -- 15x20 maps 137 us vs 128, 30x40 143 vs 139, 8x10 164 vs
// 140, 60x80 193 vs 150 -- so it is selectable (HP_CONV_ALGO_WINO_SPLIT, HP_WINO2 mask) and off by default.  Ablations
// (HP_W2ABL_*, tools/w2_ablate.sh; 15x20 layer): no output transform -22 us (the LDS exchange: 64 ds_write_b32 + 128
// ds_read_b32 per lane and item); no staging -26; skeleton (barriers + weight-fragment reads) 45; transform alone on top of
// the skeleton +12, MFMAs alone +10, both +45: the two waves of a SIMD do not overlap as hoped -- beside a 32-cycle MFMA
// only ~5 VALU issues hide (MI355X_MICROARCH.md), the transform is ~210 VALU per 24 MFMAs (128 of them the fp16 hi / lo
// split), and with one workgroup per CU nothing else fills the LDS / barrier latencies.  2.25x fewer MFMA cycles buy nothing
// when the matrix pipe is not what the direct kernels wait for either.
#include <algorithm>
#include <cstdlib>

#include "conv.h"

namespace hp {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

constexpr int kThr = 512;
constexpr int CK2 = 16;        // input channels per chunk
constexpr int BNC = 64;        // output channels per item
constexpr int TPI = 64;        // tiles per item
constexpr int U_BYTES = 16 * 2 * 2 * 64 * 16;  // weight tile of a chunk: [pos row half 2][pos col 4][pos row in half 2][nblk 2][hi | lo][lane 64] x 16 B = 64 KB
constexpr int Z_BYTES = 2 * 4 * 2 * 32 * 64 * 4;  // exchange of the output transform: [th][pg][ox][e 32][lane] floats = 128 KB
constexpr int kMaxP = 640;     // staged pixels of an item (whole image rows): 2 buffers x 4 planes x 16 B x (P + 1); <= 128 x 5 (NLD = 6 spills)

struct W2Div { FastDiv per, tw, nb; };  // by TH * TW, TW, cout blocks

struct W2Geom { int TH, TW, T, Pmax; bool ok; };

__host__ __device__ inline void w2_range(int bm, int T, int TH, int TW, int H, int W, int& lo, int& P) {
  const int per = TH * TW;
  const int t0 = bm * TPI, t1 = (t0 + TPI - 1 < T - 1) ? t0 + TPI - 1 : T - 1;
  const int i0 = t0 / per, th0 = (t0 - i0 * per) / TW;
  const int i1 = t1 / per, th1 = (t1 - i1 * per) / TW;
  const int r0 = 2 * th0 - 1 > 0 ? 2 * th0 - 1 : 0;
  const int r1 = 2 * th1 + 2 < H - 1 ? 2 * th1 + 2 : H - 1;
  lo = (i0 * H + r0) * W;
  P = (i1 * H + r1) * W + W - lo;
}

// NLD: 16-B pixel loads per thread and chunk (the staged range holds <= 128 NLD pixels)
template <bool PRE, int NLD>
__global__ __launch_bounds__(kThr) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino2_split(
    ConvArgs a, int TH, int TW, int T, int n_items, int Pp, W2Div fd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* const ul = lds_raw;                                        // U_BYTES
  floatx4* const pixl = reinterpret_cast<floatx4*>(lds_raw + U_BYTES);      // [2 buffers][4 planes][Pp] float4, slot Pp - 1 = zeros
  float* const pl = reinterpret_cast<float*>(pixl + 2 * 4 * Pp);           // PRE: [Cin] scale, [Cin] shift

  // XCD-aware numbering: dispatch order b -> XCD b % 8; an XCD walks a contiguous item range with the couts fastest
  const int per_xcd = (n_items + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= n_items) return;
  const int n_nb = a.Cout / BNC;
  const int bm = fdiv(lin, fd.nb), nb64 = lin - bm * n_nb;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int th = wave & 1, pg = wave >> 1, grp = wave >> 2;  // waves w and w + 4 share a SIMD: one of each group
  const int H = a.H, W = a.W, Cin = a.Cin;
  const int nchunks = Cin / CK2;
  const int ZS = Pp - 1;

  float act_sx, act_inv;
  conv_act_scale(a, act_sx, act_inv);
  if (a.amax_in) { act_sx *= 0.25f; act_inv *= 4.f; }  // |B^T d B| <= 4 max|d|

  int lo, P;
  w2_range(bm, T, TH, TW, H, W, lo, P);

  // zero slots of both pixel buffers, prologue constants
  if (tid < 8) pixl[tid * Pp + ZS] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (PRE)
    for (int i = tid; i < Cin; i += kThr) { pl[i] = a.pre_scale[i] * act_sx; pl[Cin + i] = a.pre_shift[i] * act_sx; }

  // ---- staging: pixel p of the range, channel quad q = index % 4
  const int n_img = (int)(a.M / ((int64_t)a.Ho * a.Wo));
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((int64_t)n_img * H * W * Cin * 4), 0x00020000);
  const int q4 = tid & 3, prow = tid >> 2;  // thread stages pixels prow + 128 k
  const unsigned char* const ug = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)nb64 * U_BYTES);
  const size_t u_chunk_stride = (size_t)n_nb * U_BYTES;
  floatx4 sp[NLD];
  auto issue_pix = [&](int c) {
#ifdef HP_W2ABL_NO_PIX
    if (c > 1) return;
#endif
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int p = prow + 128 * k;
      sp[k] = p < P ? __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ((lo + p) * Cin + c * CK2 + 4 * q4) * 4, 0, 0))
                    : floatx4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_pix = [&](int c, int buf) {
#ifdef HP_W2ABL_NO_PIX
    if (c > 1) return;
#endif
    floatx4 ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};
    if (PRE) {
      ps = *reinterpret_cast<const floatx4*>(pl + c * CK2 + 4 * q4);
      pb = *reinterpret_cast<const floatx4*>(pl + Cin + c * CK2 + 4 * q4);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int p = prow + 128 * k;
      if (p < P) {
        floatx4 v = sp[k];
        if (PRE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], ps[e], pb[e]), 0.f);
        } else {
          v = v * act_sx;
        }
        pixl[(buf * 4 + q4) * Pp + p] = v;
      }
    }
  };
  // weight tile: LDS-direct loads (no registers): a chunk's tile is four 16-KB steps (position column p of every position
  // row); the eight waves move a step with two 1-KB instructions each.  The compiler does not see these loads: every
  // barrier that publishes them is preceded by an explicit s_waitcnt vmcnt(0).  M0 (the LDS base of the transfer) is
  // written inside the asm statement and cannot be named as a clobber (hipcc: reserved register); nothing else in this
  // kernel uses it -- LDS instructions do not read M0 on gfx9+, and the ISA of every instantiation was checked for it.
  const unsigned ul_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ul;
  auto dma_half = [&](int c, int half) {
#ifdef HP_W2ABL_NO_DMA
    if (c > 0) return;
#endif
    const unsigned char* gsrc = ug + (size_t)c * u_chunk_stride + half * 32768 + wave * 4096 + lane * 16;
    const unsigned dst = ul_addr + half * 32768 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                   :: "s"(dst + i * 1024), "v"(gsrc + i * 1024) : "memory");
  };
  auto publish = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
  };
  // ---- the lane's tile: its two pixel rows (position row pg) x 4 columns as LDS slots (padding -> the zero slot)
  const int t = lane & 31, oct = lane >> 5;
  const int g = bm * TPI + th * 32 + t;
  const bool t_in = g < T;
  const int gg = t_in ? g : 0;
  const int img = fdiv(gg, fd.per), rr = gg - img * (TH * TW);
  const int tth = fdiv(rr, fd.tw), ttw = rr - tth * TW;
  const int ih0 = 2 * tth - 1, iw0 = 2 * ttw - 1;
  // position row pg of B^T d: rows (ra, rb, sign): 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
  const int ra = pg == 0 ? 0 : (pg == 2 ? 2 : 1), rb = pg == 0 ? 2 : (pg == 1 ? 2 : (pg == 2 ? 1 : 3));
  const float sgn = pg == 1 ? 1.f : -1.f;
  int doff[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int pr = k < 4 ? ra : rb, pc = k & 3;
    const bool ok = t_in & ((unsigned)(ih0 + pr) < (unsigned)H) & ((unsigned)(iw0 + pc) < (unsigned)W);
    doff[k] = ok ? (img * H + ih0 + pr) * W + iw0 + pc - lo : ZS;
  }

  floatx16 acc[4][2];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][nb][r] = 0.f;

  // ---- prologue: chunk 0's pixels
  issue_pix(0);
  if (PRE) __syncthreads();  // pl
  store_pix(0, 0);
  __syncthreads();

  // ---- main loop, two intervals per chunk, one barrier each.  The waves of group 0 (position rows 0, 1) transform chunk c
  // in interval 2c and multiply it in interval 2c + 1; group 1 (rows 2, 3) runs one interval behind -- so that on every SIMD
  // one wave feeds the matrix pipe while the other does the VALU / LDS work of the transform.  A group's half of the weight
  // tile is single-buffered: it is reloaded (LDS-direct) during the interval in which the group transforms.
  const halfx8* const ufr = reinterpret_cast<const halfx8*>(ul) + ((grp * 4 * 2 + (pg & 1)) * 4) * 64 + lane;  // + ((p * 2) * 4 + nb * 2 + hl) * 64
  halfx8 vh[4], vl[4];
  auto transform = [&](int c) {
    const floatx4* const pb = pixl + ((c & 1) * 4 + 2 * oct) * Pp;
#ifdef HP_W2ABL_NO_XFORM
    {
      const floatx4 d = pb[doff[0]];
      const halfx4 q = __builtin_convertvector(d, halfx4);
      for (int p = 0; p < 4; ++p) { vh[p] = __builtin_shufflevector(q, q, 0, 1, 2, 3, 4, 5, 6, 7); vl[p] = vh[p]; }
      return;
    }
#endif
    halfx4 qh[4][2], ql[4][2];  // [position][channel quad]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      floatx4 tr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const floatx4 da = pb[h * Pp + doff[j]], db = pb[h * Pp + doff[4 + j]];
        tr[j] = da + db * sgn;
      }
      const floatx4 V0 = tr[0] - tr[2], V1 = tr[1] + tr[2], V2 = tr[2] - tr[1], V3 = tr[1] - tr[3];
      qh[0][h] = __builtin_convertvector(V0, halfx4); ql[0][h] = __builtin_convertvector(V0 - __builtin_convertvector(qh[0][h], floatx4), halfx4);
      qh[1][h] = __builtin_convertvector(V1, halfx4); ql[1][h] = __builtin_convertvector(V1 - __builtin_convertvector(qh[1][h], floatx4), halfx4);
      qh[2][h] = __builtin_convertvector(V2, halfx4); ql[2][h] = __builtin_convertvector(V2 - __builtin_convertvector(qh[2][h], floatx4), halfx4);
      qh[3][h] = __builtin_convertvector(V3, halfx4); ql[3][h] = __builtin_convertvector(V3 - __builtin_convertvector(qh[3][h], floatx4), halfx4);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      vh[p] = __builtin_shufflevector(qh[p][0], qh[p][1], 0, 1, 2, 3, 4, 5, 6, 7);
      vl[p] = __builtin_shufflevector(ql[p][0], ql[p][1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto multiply = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const halfx8 wh = ufr[(p * 8 + nb * 2 + 0) * 64], wl = ufr[(p * 8 + nb * 2 + 1) * 64];
        floatx16 cc = acc[p][nb];
#ifdef HP_W2ABL_NO_MFMA
        cc[0] += (float)wh[0] * (float)vh[p][0] + (float)wl[0] * (float)vl[p][0];
        acc[p][nb] = cc;
        continue;
#endif
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vh[p], cc, 0, 0, 0);  // D[cout][tile]
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vl[p], cc, 0, 0, 0);
        acc[p][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, vh[p], cc, 0, 0, 0);
      }
  };
  // Both groups run the SAME loop; group 1 enters it one barrier late and group 0 leaves it one barrier late.  What a
  // thread stages in an interval depends on the interval's parity in WORKGROUP time (even: group 0's weights and the pixel
  // loads, odd: group 1's weights and the pixel stores), hence on its group.
  if (grp == 1) {
    dma_half(0, 0);
    if (1 < nchunks) { issue_pix(1); store_pix(1, 1); }
    publish();
  }
  for (int c = 0; c < nchunks; ++c) {
    dma_half(c, grp);
    transform(c);
    publish();
    // pixels travel while the thread multiplies (their registers are free of the transform's temporaries then): group 0
    // stages chunk c + 1 here, group 1 -- an interval later in workgroup time -- chunk c + 2
    const int cn = c + 1 + grp;
    if (grp == 0) dma_half(c, 1);
    else if (c + 1 < nchunks) dma_half(c + 1, 0);
    if (cn < nchunks) issue_pix(cn);
    multiply();
    if (cn < nchunks) store_pix(cn, cn & 1);
    publish();
  }
  if (grp == 0) publish();

#ifdef HP_W2ABL_NO_EPI
  if (acc[0][0][0] != 12345.f) return;
#endif
  // ---- output transform, step 1: this position row -> (z0, z1) per (tile, cout); meet the other rows in LDS
  __syncthreads();  // all waves are done with the operand buffers: the exchange area aliases them
  float* const zl = reinterpret_cast<float*>(lds_raw);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r];
      const int e = nb * 16 + r;
      zl[(((th * 4 + pg) * 2 + 0) * 32 + e) * 64 + lane] = (m0 + m1) + m2;
      zl[(((th * 4 + pg) * 2 + 1) * 32 + e) * 64 + lane] = (m1 - m2) - m3;
    }
  __syncthreads();
  // ---- step 2: wave (th, pg) finishes output pixel (oy, ox) = (pg >> 1, pg & 1) of its 32 tiles, all 64 couts
  const int oy = pg >> 1, ox = pg & 1;
  const int oh = 2 * tth + oy, ow = 2 * ttw + ox;
  const bool e_ok = t_in & (oh < a.Ho) & (ow < a.Wo);
  const int64_t e_off = (((int64_t)img * a.Ho + oh) * a.Wo + ow) * a.Cout + nb64 * BNC + 4 * oct;
  const float* const unscale = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.w) + (size_t)16 * a.Cout * Cin * 4);
  float chk = 0.f, amax = 0.f;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int jg = 0; jg < 4; ++jg) {
      floatx4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = nb * 16 + jg * 4 + i;
        const float* zp = zl + ((th * 4 * 2 + ox) * 32 + e) * 64 + lane;  // + row * (2 * 32 * 64)
        const float z0 = zp[0 * 4096], z1 = zp[1 * 4096], z2 = zp[2 * 4096], z3 = zp[3 * 4096];
        v[i] = oy == 0 ? (z0 + z1) + z2 : (z1 - z2) - z3;
      }
      const int col = nb * 32 + jg * 8;  // + 4 oct (in e_off) + i
      if (e_ok) {
        v = v * (*reinterpret_cast<const floatx4*>(unscale + nb64 * BNC + col + 4 * oct) * act_inv);
        if (a.bias) v += *reinterpret_cast<const floatx4*>(a.bias + nb64 * BNC + col + 4 * oct);
        if (a.residual) v += *reinterpret_cast<const floatx4*>(a.residual + e_off + col);
        chk += (v[0] + v[1]) + (v[2] + v[3]);
        if (a.relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        *reinterpret_cast<floatx4*>(a.y + e_off + col) = v;
      }
    }
  if (a.status && !(fabsf(chk) <= 3.0e38f)) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (a.amax_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0 && amax > 0.f) {
      const unsigned mine = __float_as_uint(amax);
      unsigned* const slot = a.amax_out + (blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride;
      if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
    }
  }
}

// w [Cout][Kpad] with K ordered (kh, kw, c) (BN folded)  ->  U = G g G^T scaled per cout by a power of two into [2^13, 2^14),
// split into fp16 hi / lo, in the LDS image of a (chunk, 64-cout block): [chunk][block][pos row half 2][pos col 4][pos row in half 2][nblk 2][hi | lo][lane 64][8 halves],
// lane = 32 (k / 8) + cout % 32, element k % 8; unscale[Cout] (fp32) behind it.  One block per cout.
__global__ __launch_bounds__(256) void wino2_weight_transform(const float* __restrict__ w, unsigned char* __restrict__ out, int Cout, int Cin, int Kpad) {
  const int o = blockIdx.x;
  __shared__ float red[256];
  auto xform = [&](int ci, float (&u)[16]) {
    float gk[3][3];
    for (int y = 0; y < 3; ++y)
      for (int x = 0; x < 3; ++x) gk[y][x] = w[(int64_t)o * Kpad + (y * 3 + x) * Cin + ci];
    float tg[4][3];
    for (int x = 0; x < 3; ++x) {
      tg[0][x] = gk[0][x];
      tg[1][x] = 0.5f * (gk[0][x] + gk[1][x] + gk[2][x]);
      tg[2][x] = 0.5f * (gk[0][x] - gk[1][x] + gk[2][x]);
      tg[3][x] = gk[2][x];
    }
    for (int i = 0; i < 4; ++i) {
      u[4 * i + 0] = tg[i][0];
      u[4 * i + 1] = 0.5f * (tg[i][0] + tg[i][1] + tg[i][2]);
      u[4 * i + 2] = 0.5f * (tg[i][0] - tg[i][1] + tg[i][2]);
      u[4 * i + 3] = tg[i][2];
    }
  };
  float mx = 0.f;
  for (int ci = threadIdx.x; ci < Cin; ci += 256) {
    float u[16];
    xform(ci, u);
    for (int p = 0; p < 16; ++p) mx = fmaxf(mx, fabsf(u[p]));
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);
  const int sh = mx > 0.f ? 14 - e : 0;
  float* const unscale = reinterpret_cast<float*>(out + (size_t)16 * Cout * Cin * 4);
  if (threadIdx.x == 0) unscale[o] = ldexpf(1.f, -sh);
  _Float16* const Uh = reinterpret_cast<_Float16*>(out);
  const int n_nb = Cout / BNC, nb64 = o / BNC, nb = (o % BNC) / 32, cl = o % 32;
  for (int ci = threadIdx.x; ci < Cin; ci += 256) {
    float u[16];
    xform(ci, u);
    const int chunk = ci / CK2, k = ci % CK2, ln = 32 * (k / 8) + cl, kk = k % 8;
    for (int p = 0; p < 16; ++p) {
      const float v = ldexpf(u[p], sh);
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      const size_t base = ((size_t)(chunk * n_nb + nb64) * U_BYTES) / 2;  // halves
      const int ps = (((p >> 3) * 4 + (p & 3)) * 2) + ((p >> 2) & 1);  // [row half][column][row in half]
      Uh[base + (size_t)(((ps * 2 + nb) * 2 + 0) * 64 + ln) * 8 + kk] = hi;
      Uh[base + (size_t)(((ps * 2 + nb) * 2 + 1) * 64 + ln) * 8 + kk] = lo;
    }
  }
}

W2Geom w2_geom(int H, int W, int64_t n_img) {
  W2Geom g{};
  g.TH = (H + 1) / 2; g.TW = (W + 1) / 2;
  const int64_t T64 = n_img * g.TH * g.TW;
  g.ok = T64 > 0 && T64 < (1 << 30) && n_img * H * W < (1ll << 30);
  if (!g.ok) return g;
  g.T = (int)T64;
  const int tiles_m = (g.T + TPI - 1) / TPI;
  const int look = std::min(tiles_m, g.TH * g.TW + 1);
  int lo, P;
  for (int bm = 0; bm < look; ++bm) {
    w2_range(bm, g.T, g.TH, g.TW, H, W, lo, P);
    g.Pmax = std::max(g.Pmax, P);
  }
  w2_range(tiles_m - 1, g.T, g.TH, g.TW, H, W, lo, P);
  g.Pmax = std::max(g.Pmax, P);
  return g;
}

const W2Geom& w2_cached_geom(int H, int W, int64_t n_img) {
  struct Entry { int H, W; int64_t n; W2Geom g; };
  static Entry cache[16];
  static int used = 0, next = 0;
  for (int i = 0; i < used; ++i)
    if (cache[i].H == H && cache[i].W == W && cache[i].n == n_img) return cache[i].g;
  const int slot = used < 16 ? used++ : (next++ % 16);
  cache[slot] = Entry{H, W, n_img, w2_geom(H, W, n_img)};
  return cache[slot].g;
}

int w2_plane(int Pmax) {  // float4 slots per plane: the range + the zero slot, odd
  int n = Pmax + 1;
  if (n % 2 == 0) ++n;
  return n;
}

size_t w2_lds_bytes(int Pmax, int Cin) {
  const size_t need = (size_t)U_BYTES + (size_t)2 * 4 * w2_plane(Pmax) * 16 + (size_t)2 * Cin * 4;
  return std::max(need, (size_t)Z_BYTES);
}

}  // namespace

bool conv_wino2_launchable(const ConvArgs& a) {
  if (a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.H < 2 || a.W < 2) return false;
  if (a.Cin % CK2 != 0 || a.Cin < 2 * CK2 || a.Cout % BNC != 0 || a.relu == HP_ACT_SWISH || (a.pre_scale && !a.pre_shift)) return false;
  const int64_t n_img = a.M / ((int64_t)a.Ho * a.Wo);
  if (n_img * a.H * a.W * a.Cin * 4 >= (1ll << 31) || a.M * a.Cout >= (1ll << 31)) return false;  // 32-bit buffer / element offsets
  const W2Geom& g = w2_cached_geom(a.H, a.W, n_img);
  return g.ok && g.Pmax <= kMaxP && w2_lds_bytes(g.Pmax, a.Cin) <= 160 * 1024;
}

size_t conv_wino2_weight_bytes(int cout, int cin) { return (size_t)16 * cout * cin * 4 + (size_t)cout * 4; }

int conv_wino2_transform_weights(const float* d_w, void* d_U, int cout, int cin, int Kpad, hipStream_t stream) {
  hipLaunchKernelGGL(wino2_weight_transform, dim3(cout), dim3(256), 0, stream, d_w, reinterpret_cast<unsigned char*>(d_U), cout, cin, Kpad);
  return check_launch("wino2_weight_transform");
}

// a.w must point at the weights of conv_wino2_transform_weights
int launch_conv_wino2(const ConvArgs& a, hipStream_t stream) {
  if (!conv_wino2_launchable(a)) return fail(HP_ERR_ARG, "conv3x3_wino2_split: geometry not supported (check conv_wino2_launchable)");
  const W2Geom& g = w2_cached_geom(a.H, a.W, a.M / ((int64_t)a.Ho * a.Wo));
  const int tiles_m = (g.T + TPI - 1) / TPI, n_nb = a.Cout / BNC;
  const int n_items = tiles_m * n_nb;
  const W2Div fd{make_fastdiv((unsigned)(g.TH * g.TW)), make_fastdiv((unsigned)g.TW), make_fastdiv((unsigned)n_nb)};
  const size_t lds = w2_lds_bytes(g.Pmax, a.Cin);
  const dim3 grid(8 * ((n_items + 7) / 8));
  const int nld = (g.Pmax + 127) / 128;
  typedef void (*K)(ConvArgs, int, int, int, int, int, W2Div);
  static const K kern[2][3] = {{conv3x3_wino2_split<false, 3>, conv3x3_wino2_split<false, 4>, conv3x3_wino2_split<false, 5>},
                               {conv3x3_wino2_split<true, 3>, conv3x3_wino2_split<true, 4>, conv3x3_wino2_split<true, 5>}};
  const int ni = nld <= 3 ? 0 : nld - 3, pi = a.pre_scale ? 1 : 0;
  static bool opted[2][3] = {{false, false, false}, {false, false, false}};
  if (!opted[pi][ni]) {
    HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern[pi][ni]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    opted[pi][ni] = true;
  }
  hipLaunchKernelGGL(kern[pi][ni], grid, dim3(kThr), lds, stream, a, g.TH, g.TW, g.T, n_items, w2_plane(g.Pmax), fd);
  return check_launch("conv3x3_wino2_split");
}

}  // namespace hp
