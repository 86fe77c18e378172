// 3x3 / stride-1 / pad-1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// 85 % of the backbone FLOPs are 3x3 stride-1 convolutions and the direct kernels
// (conv_patch.hip) already sit at ~90 % of what the fp32 MFMA pipe sustains, so the remaining
// lever is the amount of matrix work itself: Y = A^T [ (G g G^T) . (B^T d B) ] A produces a 2x2
// output tile from a 4x4 input tile with 16 multiplies per (cin, cout) instead of 36 -- 2.25x
// fewer MFMA FLOPs for the same result up to fp32 round-off (the transform constants are 0, +-1,
// +-1/2, so the error stays within ~2x of the direct kernel's; tests/test_gpu_kernels.py).
//
// GEMM view: 16 independent products, one per transform position p = (xi, nu):
//   M_p[tile][cout] = sum_cin V_p[tile][cin] * U_p[cin][cout]
// Everything is fused in ONE kernel -- no transformed tensor ever touches HBM:
//   * a wave owns 16 consecutive output tiles (linear index over batch x tile-rows x tile-cols,
//     so any feature-map size fills the tile exactly) x 32 output channels x ALL 16 positions,
//     with v_mfma_f32_16x16x4_f32: the accumulators of a lane are then the 16 positions of the
//     same (tile, cout) -> the output transform A^T M A is pure register arithmetic;
//   * the block (4 waves = 64 tiles) stages the pixel range its tiles touch -- a contiguous
//     range of the pixel-linear NHWC index, whole image rows -- for 16 input channels at a time
//     in LDS with coalesced 16-B loads (a first version let every lane fetch its own 4x4 pixels
//     from global memory: 64 separate 16-B accesses per wave instruction, far slower).  The
//     pre-activation BN+ReLU prologue is applied on the way into LDS.  Lane (tile = lane & 15,
//     kg = lane >> 4) then reads the 4x4 pixels of its tile for channels 4kg..4kg+3
//     (16 ds_read_b128; padding pixels read a row of zeros) and forms B^T d B with 32 packed
//     float4 adds; the float4 feeds 4 MFMA k-steps (k-step j multiplies channel 4 kg + j; A and
//     B use the same order, so the permutation is free);
//   * transformed weights U (device pre-pass, BN folded) are packed so that the [16 positions]
//     [4 kg][32 couts] float4 image of one (chunk, cout block) is copied to LDS verbatim
//     (double buffered) and shared by the 4 waves of the block;
//   * LDS layouts are chosen for the lane groups of ds_read_b128 ({0-3,12-15,20-27}, ... :
//     MI355X_MICROARCH.md, LDS): weights [pos][kg][cout] float4 -> the 16 lanes of a group hit
//     16 different 16-B slots whatever their kg; pixels as 4 channel-quad planes [kg][P'] float4
//     with P' odd, so the stride-2-pixel reads of neighbouring tiles in two kg planes interleave;
//   * ~400 VGPRs+AGPRs (128 accumulators): one wave per SIMD, one block per CU, so latency must be
//     hidden inside the wave: the stage (item, chunk) two ahead is in flight from global memory
//     while the one after the current is written to LDS after three quarters of the current
//     chunk's 128 MFMAs; its LDS reads fly under the last quarter.  The staging code is branch
//     free (clamped addresses, a dump row) so that the compiler can interleave it with the MFMAs.
//     Blocks are PERSISTENT (one per CU, static round-robin over the work items of its XCD), the
//     load cursor simply runs on into the next item, so fill latency hides behind the epilogue.
// Item = 64 tiles (256 output pixels) x 32 couts; an XCD walks a contiguous item range with
// the couts fastest, so the blocks that re-read the same input share an L2.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv.h"

namespace hp {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef _Float16 whalfx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr int CK = 16;        // channels per chunk
constexpr int BN = 32;        // output channels per item
constexpr int TPB = 64;       // tiles per item (16 per wave)
constexpr int U_BUF = 16 * 4 * BN * 4;  // floats per weight buffer: [pos][kg][cout] float4

// (FastDiv / fdiv: conv.h)
struct WinoDiv { FastDiv per, tw, tn; };  // by TH*TW, TW, tiles_n

// pixel-linear range [lo, lo + P) that the tiles of item row bm touch (whole image rows)
__device__ __forceinline__ void item_range_dev(int bm, int T, int TH, int TW, int H, int W, const WinoDiv& fd, int& lo, int& P) {
  const int per = TH * TW;
  const int t0 = bm * TPB, t1 = (t0 + TPB - 1 < T - 1) ? t0 + TPB - 1 : T - 1;
  const int i0 = fdiv(t0, fd.per), th0 = fdiv(t0 - i0 * per, fd.tw);
  const int i1 = fdiv(t1, fd.per), th1 = fdiv(t1 - i1 * per, fd.tw);
  const int r0 = 2 * th0 - 1 > 0 ? 2 * th0 - 1 : 0;
  const int r1 = 2 * th1 + 2 < H - 1 ? 2 * th1 + 2 : H - 1;
  lo = (i0 * H + r0) * W;
  P = (i1 * H + r1) * W + W - lo;
}
__host__ __device__ inline void item_range(int bm, int T, int TH, int TW, int H, int W, int& lo, int& P) {
  const int per = TH * TW;
  const int t0 = bm * TPB, t1 = (t0 + TPB - 1 < T - 1) ? t0 + TPB - 1 : T - 1;
  const int i0 = t0 / per, th0 = (t0 - i0 * per) / TW;
  const int i1 = t1 / per, th1 = (t1 - i1 * per) / TW;
  const int r0 = 2 * th0 - 1 > 0 ? 2 * th0 - 1 : 0;
  const int r1 = 2 * th1 + 2 < H - 1 ? 2 * th1 + 2 : H - 1;
  lo = (i0 * H + r0) * W;
  P = (i1 * H + r1) * W + W - lo;
}

// plane length (float4 slots) for a staged range of up to Pmax pixels: + zero row + dump row,
// odd and = 3 (mod 8) (read and write bank spread, see the header)
__host__ __device__ inline int plane_len(int Pmax) {
  int n = Pmax + 2;
  while (n % 8 != 3) ++n;
  return n;
}

// plane length of the two-waves-per-SIMD kernel: 128 NLD staged rows + the zero slot, = 3 (mod 8)
__host__ __device__ constexpr int plane_len8(int nld) {
  int n = 128 * nld + 1;
  while (n % 8 != 3) ++n;
  return n;
}

// a - b on the packed-f32 pipe (hipcc packs float adds but not subtractions)
__device__ __forceinline__ floatx4 sub4(floatx4 a, floatx4 b) {
  floatx2 lo, hi;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(a.xy), "v"(b.xy));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(a.zw), "v"(b.zw));
  return floatx4{lo.x, lo.y, hi.x, hi.y};
}

#ifdef HP_WABL_TIMING  // per-block time stamps (diagnostic build only): [block][0 start, 1 first K loop, 2+i end of item i]
constexpr int kStampSlots = 64;
__device__ long long g_wino_stamps[512 * kStampSlots];
#define HP_STAMP(slot) do { if (threadIdx.x == 0 && (slot) < kStampSlots) g_wino_stamps[blockIdx.x * kStampSlots + (slot)] = wall_clock64(); } while (0)
#else
#define HP_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ floatx2 add2(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx2 sub2(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// a * s + c with a wave-uniform s (an SGPR pair): two v_pk_fma_f32 per float4
__device__ __forceinline__ floatx4 pk_fma4(floatx4 a, floatx2 s, floatx4 c) {
  floatx2 lo, hi;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.xy), "s"(s), "v"(c.xy));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.zw), "s"(s), "v"(c.zw));
  return floatx4{lo.x, lo.y, hi.x, hi.y};
}

// float4 add / subtract of the input transform: packed (two v_pk_add_f32).  Spelling them out as
// single-lane-op instructions (the rule for bf16 MFMA kernels, MI355X_MICROARCH.md) doubled the
// instruction count and was slower here: every instruction beside an f32 MFMA costs its issue time.
__device__ __forceinline__ floatx4 add4s(floatx4 a, floatx4 b) { return a + b; }
__device__ __forceinline__ floatx4 sub4s(floatx4 a, floatx4 b) { return sub4(a, b); }

template <bool PRE, int NLD>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_wino_f32(
    ConvArgs a, int TH, int TW, int T, int tiles_n, int n_items, int Pmax, WinoDiv fd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Pp = plane_len(Pmax);
  float* const rawl = lds;               // [4 kg][Pp] float4; slot Pmax = zeros, slot Pmax+1 = dump
  float* const ul = lds + 4 * Pp * 4;    // [2][16 pos][4 kg][BN] float4

  const int nslot = gridDim.x / 8;
  const int ipx = (n_items + 7) / 8;
  const int item_begin = (blockIdx.x % 8) * ipx;
  const int item_end = item_begin + ipx < n_items ? item_begin + ipx : n_items;
  int item = item_begin + blockIdx.x / 8;
  if (item >= item_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = lane >> 4;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const int nchunks = Cin / CK;
  HP_STAMP(0);
#ifdef HP_WABL_TIMING
  int stamp_slot = 2;
  if (threadIdx.x == 0) g_wino_stamps[blockIdx.x * kStampSlots + 62] = clock64();  // shader-clock counter at the start
#endif
  const int c4 = tid & 3, srow = tid >> 2;  // channel quad / first pixel row this thread stages
  if (tid < 16) rawl[((tid >> 2) * Pp + Pmax) * 4 + (tid & 3)] = 0.f;

  // ---- global loads go through buffer descriptors: the per-lane byte offset is a constant, the
  //      stage-dependent part is wave-uniform (soffset), so a load costs no vector ALU work; rows
  //      past the end of the tensor read as zero instead of faulting
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((int64_t)a.M / ((int64_t)a.Ho * a.Wo) * H * W * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t ursrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 16 * a.Cout * Cin * 4, 0x00020000);
  const int x_voff = (srow * Cin + 4 * c4) * 4;
  const int x_kstride = 64 * Cin * 4;  // bytes between the rows a thread stages
  // weights: the [16][4][32] float4 image of a (chunk, cout block) is 16 segments of 2 KB
  const int u_pos_stride = (a.Cout / BN) * (4 * BN * 16);  // bytes between positions
  const int u_chunk_stride = 16 * u_pos_stride;
  const int u_voff = (tid >> 7) * u_pos_stride + (tid & 127) * 16;  // + nb*2048 + 2 i pos
  float* const udst = ul + tid * 4;                                  // + 1024 i floats
  const float* const ufr = ul + (kg * BN + (lane & 15)) * 4;         // + (pos*128 + nt*16) * 4
  float* const rdst = rawl + (c4 * Pp) * 4;

  // ---- two load cursors run ahead of the compute cursor over the stages (item, chunk): the
  //      pixel cursor three stages (two register sets, alternating with the parity of the chunk:
  //      pixel data comes from HBM / the far L2 and needs more than one chunk of MFMAs to arrive),
  //      the weight cursor two (L2 resident).  Past the last stage they keep re-reading it
  //      (harmless, keeps the loop branch free).
  struct Cursor { int item, c, lo, P, nb; };  // nb = cout block of the item
  auto locate = [&](Cursor& k, bool range) {
    const int bm = fdiv(k.item, fd.tn);
    k.nb = k.item - bm * tiles_n;
    if (range) item_range_dev(bm, T, TH, TW, H, W, fd, k.lo, k.P);
  };
  auto advance = [&](Cursor& k, bool range) {
    const bool wrap = k.c + 1 == nchunks;
    const bool more = !wrap || k.item + nslot < item_end;
    if (more) {
      k.c = wrap ? 0 : k.c + 1;
      if (wrap) {
        k.item += nslot;
        locate(k, range);
      }
    }
  };
  Cursor rc{item, 0, 0, 0, 0}, uc{item, 0, 0, 0, 0};
  locate(rc, true);
  locate(uc, false);
  floatx4 rsA[NLD], rsB[NLD], us[8];
  floatx4 psA = {1.f, 1.f, 1.f, 1.f}, pbA = {0.f, 0.f, 0.f, 0.f}, psB = psA, pbB = pbA;  // prologue of the held chunks
  int heldA = 0, heldB = 0;  // P of the stage held in each set
  // the loads of a stage are issued in 8 parts (one per step): a burst of 4 waves x 14-18 loads
  // backs up the texture-address FIFO and the waves stall at issue with the matrix pipe idle
  auto issue_raw = [&](auto set, int part) {
    floatx4(&rs)[NLD] = decltype(set)::value ? rsB : rsA;
    const int xs = (rc.lo * Cin + rc.c * CK) * 4;
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (part < 0 || k % 8 == part)
        rs[k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, x_voff, xs + k * x_kstride, 0));
    if (part >= 0 && part < 7) return;
    if (PRE) {
      (decltype(set)::value ? psB : psA) = *reinterpret_cast<const floatx4*>(a.pre_scale + rc.c * CK + 4 * c4);
      (decltype(set)::value ? pbB : pbA) = *reinterpret_cast<const floatx4*>(a.pre_shift + rc.c * CK + 4 * c4);
    }
    (decltype(set)::value ? heldB : heldA) = rc.P;
    advance(rc, true);
  };
  auto issue_u = [&](int part) {
    const int s = uc.nb * (4 * BN * 16) + uc.c * u_chunk_stride;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (part < 0 || i == part)
        us[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, u_voff, s + 2 * i * u_pos_stride, 0));
    if (part < 0 || part == 7) advance(uc, false);
  };
  // held stage -> LDS (pixels: the single buffer; weights: buffer ubuf), in eight parts (or all: -1)
  auto store_held = [&](auto set, int ubuf, int part) {
    floatx4(&rs)[NLD] = decltype(set)::value ? rsB : rsA;
    const floatx4 ps = decltype(set)::value ? psB : psA, pb = decltype(set)::value ? pbB : pbA;
    const int held_P = decltype(set)::value ? heldB : heldA;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (part >= 0 && k % 8 != part) continue;
      const int row = srow + 64 * k;
      floatx4 v = rs[k];
      if (PRE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(v[q], ps[q], pb[q]), 0.f);
      }
      *reinterpret_cast<floatx4*>(rdst + (row < held_P ? row : Pmax + 1) * 4) = v;
    }
    float* d = udst + ubuf * U_BUF;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (part < 0 || i == part) *reinterpret_cast<floatx4*>(d + 1024 * i) = us[i];
  };
  constexpr std::integral_constant<int, 0> SET_A{};
  constexpr std::integral_constant<int, 1> SET_B{};

  int ubuf = 0;  // weight buffer holding the chunk about to be consumed
  issue_raw(SET_A, -1);  // stage 0
  issue_u(-1);
  store_held(SET_A, ubuf, -1);
  __syncthreads();
  issue_raw(SET_B, -1);  // stage 1: stored during chunk 0
  issue_raw(SET_A, -1);  // stage 2: stored during chunk 1
  issue_u(-1);           // stage 1

  // ---- the lane's input tile: LDS offsets of its 4x4 pixels (padding -> the zero slot), in three
  //      parts so that the set-up of the NEXT item can be dealt out under the last chunk's MFMAs
  int doff[16];
  int s_prow = 0, s_ih0 = 0, s_iw0 = 0;
  bool s_in = false;
  auto setup_tile = [&](int it) {
    const int bm = fdiv(it, fd.tn);
    int lo, P;
    item_range_dev(bm, T, TH, TW, H, W, fd, lo, P);
    const int g = bm * TPB + wave * 16 + (lane & 15);
    const int gg = g < T ? g : 0;
    const int img = fdiv(gg, fd.per), r = gg - img * (TH * TW);
    const int th = fdiv(r, fd.tw), tw = r - th * TW;
    s_ih0 = 2 * th - 1;
    s_iw0 = 2 * tw - 1;
    s_prow = (img * H + s_ih0) * W + s_iw0 - lo;
    s_in = g < T;
  };
  auto setup_doff = [&](int half) {
#pragma unroll
    for (int p = 8 * half; p < 8 * half + 8; ++p) {
      const bool ok = s_in & ((unsigned)(s_ih0 + p / 4) < (unsigned)H) & ((unsigned)(s_iw0 + p % 4) < (unsigned)W);
      doff[p] = (kg * Pp + (ok ? s_prow + (p / 4) * W + (p % 4) : Pmax)) * 4;
    }
  };
  floatx4 d[16], V[16];
  auto read_d_rows = [&](const float* rb, int r0, int r1) {  // pixel rows r0 and r1 of the 4x4 patch
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[4 * r0 + j] = *reinterpret_cast<const floatx4*>(rb + doff[4 * r0 + j]);
      if (r1 != r0) d[4 * r1 + j] = *reinterpret_cast<const floatx4*>(rb + doff[4 * r1 + j]);
    }
  };
  // V row i = (B^T d) row i times B
  auto xform_row = [&](int i) {
    floatx4 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      t[j] = i == 0 ? sub4(d[0 + j], d[8 + j]) : i == 1 ? d[4 + j] + d[8 + j] : i == 2 ? sub4(d[8 + j], d[4 + j]) : sub4(d[4 + j], d[12 + j]);
#ifdef HP_WABL_NO_XFORM
    V[4 * i + 0] = d[4 * i + 0]; V[4 * i + 1] = d[4 * i + 1]; V[4 * i + 2] = d[4 * i + 2]; V[4 * i + 3] = d[4 * i + 3];
#else
    V[4 * i + 0] = sub4(t[0], t[2]);
    V[4 * i + 1] = t[1] + t[2];
    V[4 * i + 2] = sub4(t[2], t[1]);
    V[4 * i + 3] = sub4(t[1], t[3]);
#endif
  };
  setup_tile(item);
  setup_doff(0);
  setup_doff(1);
  read_d_rows(rawl, 0, 2);
  read_d_rows(rawl, 1, 3);
  __syncthreads();  // every wave holds its pixels of chunk 0: the pixel buffer may be refilled
  HP_STAMP(1);

  for (;;) {
    const int bm = fdiv(item, fd.tn), n0 = (item - bm * tiles_n) * BN;
    const bool has_next = item + nslot < item_end;
#ifdef HP_WABL_TIMING
    HP_STAMP(stamp_slot);  // K loop starts
#endif
    floatx4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[p][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    // one chunk; `odd` = parity of the chunk = register set of the stage two ahead (the set of
    // the next stage is the other one)
    // LAST (compile time: the steps must stay branch free, a branch inside a step splits the
    // scheduling region): the chunk that ends the item; it sets up the tile of the NEXT item
    auto chunk = [&](int c, auto odd, auto is_last) {
      constexpr std::integral_constant<int, 1 - decltype(odd)::value> nxt{};
      constexpr bool LAST = decltype(is_last)::value;
#ifdef HP_WABL_TIMING
      if (stamp_slot == 5) HP_STAMP(32 + c);  // chunk starts of the block's second item
#endif
      // One chunk = 16 steps (one transform position each: 2 weight fragment reads for the next
      // step, 8 MFMAs), with the rest of the work dealt out between them so that the matrix pipe
      // never waits: input transform rows 1-3 under steps 0/4/8, the next stage's LDS stores under
      // steps 9-11 (pixel buffer free since the barrier after the previous chunk's reads, weights
      // go to the other buffer), the loads of the stage after that at step 12, the next chunk's
      // pixel reads under steps 12-13.
      const float* ub = ufr + ubuf * U_BUF;
      floatx4 bf[2][2];
      xform_row(0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bf[0][nt] = *reinterpret_cast<const floatx4*>(ub + nt * 16 * 4);
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        if (p + 1 < 16) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            bf[(p + 1) & 1][nt] = *reinterpret_cast<const floatx4*>(ub + ((p + 1) * 128 + nt * 16) * 4);
        }
        if (p == 0) xform_row(1);
        if (p == 4) xform_row(2);
        if (p == 8) xform_row(3);
#if !defined(HP_WABL_NO_STAGE) && !defined(HP_WABL_NO_LSTORE)
        if (p < 8) store_held(nxt, ubuf ^ 1, p);
#endif
#if !defined(HP_WABL_NO_STAGE) && !defined(HP_WABL_NO_GLOAD)
        if (p >= 8) {  // registers of the stage just stored are free: stage + 3 pixels, stage + 2 weights
          issue_raw(nxt, p - 8);
          issue_u(p - 8);
        }
#endif
        // the last chunk of an item sets up the tile of the NEXT item (its first stage is what the
        // barrier below publishes), so an item boundary costs no pixel-read latency
        // (after the block's last item: the same item again, read and never used)
        if (LAST) {
          if (p == 9) setup_tile(has_next ? item + nslot : item);
          if (p == 10) setup_doff(0);
          if (p == 11) setup_doff(1);
        }
        if (p == 12) {
#ifndef HP_WABL_NO_BARRIER
          __syncthreads();  // the next stage is in LDS
#endif

        }
#ifndef HP_WABL_NO_READD
        if (p == 12) read_d_rows(rawl, 0, 0);
        if (p == 13) read_d_rows(rawl, 2, 2);
        if (p == 14) read_d_rows(rawl, 1, 1);
        if (p == 15) read_d_rows(rawl, 3, 3);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p][j], bf[p & 1][nt][j], acc[p][nt], 0, 0, 0);
#ifndef HP_WABL_NO_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      ubuf ^= 1;
#ifndef HP_WABL_NO_BARRIER
      // every wave holds its pixels of chunk c+1 and is done with this chunk's weights: both
      // buffers may be refilled (a third weight buffer + a second pixel buffer would save this
      // barrier, but the 60x80 layers have no LDS left for them)
      __syncthreads();
#endif
    };
    for (int c = 0; c + 2 < nchunks; c += 2) {
      chunk(c, SET_A, std::false_type{});
      chunk(c + 1, SET_B, std::false_type{});
    }
    chunk(nchunks - 2, SET_A, std::false_type{});
    chunk(nchunks - 1, SET_B, std::true_type{});
#ifdef HP_WABL_TIMING
    HP_STAMP(stamp_slot + 1);  // K loop done
#endif
    // ---- output transform Y = A^T M A per (tile row i of the lane, cout tile), epilogue.
    //      A lane holds the 2x2 pixels of ONE cout; a 4x4 transpose inside each lane quad (two
    //      DPP butterfly stages) turns that into 4 consecutive couts of ONE pixel, so bias,
    //      residual and store are 16-B accesses instead of 4-B ones.
    const int lq = lane & 3;                       // after the transpose: the pixel this lane stores
    const int ncol = n0 + ((lane & 15) & ~3);      // first of its 4 couts (+ nt * 16)
    // tile of accumulator row i = 0 decoded once, then stepped (integer divisions cost ~25 VALU each)
    const int go0 = bm * TPB + wave * 16 + 4 * kg;  // uniform over the 16 lanes of a kg group
    int e_img, e_th, e_tw;
    {
      const int gc = go0 < T ? go0 : 0;
      e_img = fdiv(gc, fd.per);
      const int r = gc - e_img * (TH * TW);
      e_th = fdiv(r, fd.tw);
      e_tw = r - e_th * TW;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int go = go0 + i;
      const int img = e_img, th = e_th, tw = e_tw;
      if (++e_tw == TW) { e_tw = 0; if (++e_th == TH) { e_th = 0; ++e_img; } }
      const int oh = 2 * th + (lq >> 1), ow = 2 * tw + (lq & 1);
      const bool ok = (go < T) & (oh < a.Ho) & (ow < a.Wo);
      const int64_t obase = (((int64_t)img * a.Ho + oh) * a.Wo + ow) * a.Cout + ncol;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t0[j] = acc[0 + j][nt][i] + acc[4 + j][nt][i] + acc[8 + j][nt][i];
          t1[j] = acc[4 + j][nt][i] - acc[8 + j][nt][i] - acc[12 + j][nt][i];
        }
        float y[4];  // pixels (0,0) (0,1) (1,0) (1,1) of this lane's cout
        y[0] = t0[0] + t0[1] + t0[2];
        y[1] = t0[1] - t0[2] - t0[3];
        y[2] = t1[0] + t1[1] + t1[2];
        y[3] = t1[1] - t1[2] - t1[3];
        // stage 1: lanes l, l^1 swap the off-diagonal of each 2x2 block
        {
          const bool odd = lq & 1;
          const float s0 = odd ? y[0] : y[1], s1 = odd ? y[2] : y[3];
          const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
          const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
          if (odd) { y[0] = x0; y[2] = x1; } else { y[1] = x0; y[3] = x1; }
        }
        // stage 2: lanes l, l^2 swap the off-diagonal 2x2 blocks
        {
          const bool hi = lq & 2;
          const float s0 = hi ? y[0] : y[2], s1 = hi ? y[1] : y[3];
          const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
          const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
          if (hi) { y[0] = x0; y[1] = x1; } else { y[2] = x0; y[3] = x1; }
        }
        // y[c] = pixel lq, cout ncol + nt*16 + c
#ifdef HP_WABL_NO_EPI
        if (ok && y[0] == 123.456f) {
#else
        if (ok) {
#endif
          floatx4 v = {y[0], y[1], y[2], y[3]};
          if (a.bias) v += *reinterpret_cast<const floatx4*>(a.bias + ncol + nt * 16);
          if (a.residual) v += *reinterpret_cast<const floatx4*>(a.residual + obase + nt * 16);
          if (a.relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
          }
          *reinterpret_cast<floatx4*>(a.y + obase + nt * 16) = v;
        }
        // one (row, cout tile) group at a time: the next item's pixels and the staged loads stay
        // live across the epilogue, there are no registers for reading all accumulators up front
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef HP_WABL_TIMING
    HP_STAMP(stamp_slot + 2);  // epilogue done
    if (stamp_slot == 2 && threadIdx.x == 0) g_wino_stamps[blockIdx.x * kStampSlots + 63] = clock64();  // ... and after item 0
    stamp_slot += 3;
#endif
    item += nslot;
    if (item >= item_end) break;
  }
}


// ------------------------------------------------------------------------------------------------
// Two waves per SIMD.  With one wave per SIMD the matrix pipe idles whenever that wave waits (a
// barrier, an LDS read that is not back yet): the wave cannot queue more than one MFMA ahead, so
// every stall longer than 32 cycles is a bubble -- 27 % of a chunk in the kernel above.  Here the
// block has 8 waves: wave w and wave w + 4 share the 16 tiles of tile group w & 3 and split the 16
// transform positions BY ROWS of the 4x4 position grid (role 0: rows 0-1, role 1: rows 2-3), each
// for all 32 couts: 64 accumulators per wave, <= 256 registers, and one wave's waits are covered
// by the other wave's MFMAs.  Splitting by rows (not by couts) keeps the LDS traffic and the
// transform work per SIMD the same as above: V rows i need only three pixel rows of the 4x4 patch
//   role 0: t0 = d0 - d2, t1 = d1 + d2        role 1: t3 = d1 - d3, t2 = d2 - d1
// With role 1 reading its pixel rows in reversed order (E0,E1,E2) = (d3,d2,d1) both roles compute
//   first = E0 - E2  (= t0 | -t3),   second = E1 + s E2  (s = +1 | -1)  (= t1 | t2)
// so the K loop is the same code for both; role 1's first row is -V row 3, which the output
// transform absorbs.  The output transform Y = A^T M A is linear in the rows of M, so each wave
// reduces its two rows to a 2x2 partial result per (tile, cout), the partners swap half of it
// through LDS (role 0 finishes couts 0-15 of the item, role 1 couts 16-31) and each stores half.
constexpr int kThreads8 = 512;
constexpr int X_BUF = 8 * 8 * 64 * 2;  // floats of the exchange buffer: [wave][8][lane] float2

// NT = cout tiles (of 16) per item: 2 = the whole 32-cout block; 1 = HALF items, used for the last,
// partially filled round of a launch (see launch8): an item is then split into two 16-cout halves that
// go to different blocks, and the partner waves split the finishing work by tile pairs instead.
template <bool PRE, int NLD, int SETS, int NT>
__global__ __launch_bounds__(kThreads8) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino8_f32(
    ConvArgs a, int TH, int TW, int T, int tiles_n, int n_items, int Pmax, WinoDiv fd, int range_off, int range_len) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // every staged row has its own LDS slot (128 NLD of them, the zero slot after them), so the
  // stores need no clamping -- VALU work beside the f32 MFMAs is not free (see the chunk loop)
  constexpr int ZS = 128 * NLD;          // index of the zero slot
  constexpr int Pp = plane_len8(NLD);
  float* const rawl = lds;               // [4 kg][Pp] float4
  float* const ul = lds + 4 * Pp * 4;    // [2][16 pos][4 kg][BN] float4
  float* const xl = ul + 2 * U_BUF;      // [8 waves][8][64 lanes] float2
  float* const pl = xl + X_BUF;          // PRE: [Cin] scale, [Cin] shift of the prologue

  // items [rb, re) of this XCD's share; `item` counts VIRTUAL items from 0: whole items (NT = 2) or halves
  const int nslot = gridDim.x / 8;
  const int ipx = (n_items + 7) / 8;
  const int xb = (blockIdx.x % 8) * ipx;
  const int xe = xb + ipx < n_items ? xb + ipx : n_items;
  const int rb = xb + range_off < xe ? xb + range_off : xe;
  const int re = rb + range_len < xe ? rb + range_len : xe;
  const int item_end = (re - rb) * (3 - NT);
  int item = blockIdx.x / 8;
  if (item >= item_end) return;
  auto real_item = [&](int v) { return rb + (NT == 1 ? v >> 1 : v); };
  auto half_off = [&](int v) { return NT == 1 ? (v & 1) * 16 * 4 : 0; };  // floats: second 16 couts of the block

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave & 3, role = wave >> 2;
  const int kg = lane >> 4;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const int nchunks = Cin / CK;
  HP_STAMP(0);
#ifdef HP_WABL_TIMING
  int stamp_slot = 2;
  if (threadIdx.x == 0) g_wino_stamps[blockIdx.x * kStampSlots + 62] = clock64();
#endif
  const int c4 = tid & 3, srow = tid >> 2;  // channel quad / first pixel row this thread stages
  if (tid < 16) rawl[((tid >> 2) * Pp + ZS) * 4 + (tid & 3)] = 0.f;
  if (PRE) {  // the prologue constants live in LDS: held in registers they would stay live across the epilogue
    for (int i = tid; i < Cin; i += kThreads8) {
      pl[i] = a.pre_scale[i];
      pl[Cin + i] = a.pre_shift[i];
    }
  }

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((int64_t)a.M / ((int64_t)a.Ho * a.Wo) * H * W * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t ursrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 16 * a.Cout * Cin * 4, 0x00020000);
  const int x_voff = (srow * Cin + 4 * c4) * 4;
  const int x_kstride = 128 * Cin * 4;  // bytes between the rows a thread stages
  const int u_pos_stride = (a.Cout / BN) * (4 * BN * 16);  // bytes between positions
  const int u_chunk_stride = 16 * u_pos_stride;
  const int u_voff = (tid >> 7) * u_pos_stride + (tid & 127) * 16;  // + nb*2048 + 4 i pos
  float* const udst = ul + ((tid >> 7) * 128 + (tid & 127)) * 4;    // + 2048 i floats
  // weight fragments: local step q multiplies position q (role 0) | 12 + q for q < 4, 4 + q for q >= 4 (role 1)
  const float* const ufr_lo = ul + (kg * BN + (lane & 15)) * 4 + role * 12 * 128 * 4;
  const float* const ufr_hi = ul + (kg * BN + (lane & 15)) * 4 + role * 4 * 128 * 4;
  float* const rdst = rawl + (c4 * Pp) * 4;
  const float s_second = role ? -1.f : 1.f;
  const floatx2 s_second2 = {s_second, s_second};

  struct Cursor { int item, c, lo, P, nb; };
  auto locate = [&](Cursor& k, bool range) {
    const int it = real_item(k.item);
    const int bm = fdiv(it, fd.tn);
    k.nb = it - bm * tiles_n;
    if (range) item_range_dev(bm, T, TH, TW, H, W, fd, k.lo, k.P);
  };
  auto advance = [&](Cursor& k, bool range) {
    const bool wrap = k.c + 1 == nchunks;
    const bool more = !wrap || k.item + nslot < item_end;
    if (more) {
      k.c = wrap ? 0 : k.c + 1;
      if (wrap) {
        k.item += nslot;
        locate(k, range);
      }
    }
  };
  Cursor rc{item, 0, 0, 0, 0}, uc{item, 0, 0, 0, 0};
  locate(rc, true);
  locate(uc, false);
  floatx4 rsA[NLD], rsB[NLD], us[4];
  floatx4 ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};  // prologue constants of the stage being stored
  auto load_pre = [&](int c) {  // chunk c of the input channels
    ps = *reinterpret_cast<const floatx4*>(pl + c * CK + 4 * c4);
    pb = *reinterpret_cast<const floatx4*>(pl + Cin + c * CK + 4 * c4);
  };
  auto issue_raw = [&](auto set, int part) {  // part 0..3 (or -1: all)
    floatx4(&rs)[NLD] = (SETS == 2 && decltype(set)::value) ? rsB : rsA;
    const int xs = (rc.lo * Cin + rc.c * CK) * 4;
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (part < 0 || k % 4 == part)
        rs[k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, x_voff, xs + k * x_kstride, 0));
    if (part < 0 || part == 3) advance(rc, true);
  };
  auto issue_u = [&](int part) {
    const int s = uc.nb * (4 * BN * 16) + uc.c * u_chunk_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (part < 0 || i == part)
        us[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, u_voff, s + 4 * i * u_pos_stride, 0));
    if (part < 0 || part == 3) advance(uc, false);
  };
  auto store_held = [&](auto set, int ubuf, int part) {
    floatx4(&rs)[NLD] = (SETS == 2 && decltype(set)::value) ? rsB : rsA;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (part >= 0 && k % 4 != part) continue;
      floatx4 v = rs[k];
      if (PRE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(v[q], ps[q], pb[q]), 0.f);
      }
      *reinterpret_cast<floatx4*>(rdst + (srow + 128 * k) * 4) = v;  // rows past the item's range: never read
    }
    float* d = udst + ubuf * U_BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (part < 0 || i == part) *reinterpret_cast<floatx4*>(d + 2048 * i) = us[i];
  };
  constexpr std::integral_constant<int, 0> SET_A{};
  constexpr std::integral_constant<int, 1> SET_B{};

  int ubuf = 0;
  // pixel stages run two ahead of the one being stored in two alternating register sets, or
  // (SETS == 1: the large staged ranges, where two sets would spill) one ahead in a single set.
  // With two sets the loads of stage 1 leave together with those of stage 0: one memory latency
  // at the head of the launch instead of two.
  issue_raw(SET_A, -1);                 // stage 0
  issue_u(-1);
  if (SETS == 2) issue_raw(SET_B, -1);  // stage 1: stored during chunk 0
  if (PRE) {
    __syncthreads();
    load_pre(0);
  }
  store_held(SET_A, ubuf, -1);
  __syncthreads();
  if (PRE) load_pre(1);  // chunk 0 stores stage 1
  if (SETS == 2) issue_raw(SET_A, -1);  // stage 2: stored during chunk 1
  else issue_raw(SET_B, -1);            // stage 1 (the single set is free again)
  issue_u(-1);                          // stage 1

  // ---- the lane's three pixel rows (E0,E1,E2) of its tile: LDS offsets (padding -> the zero slot)
  int doff[12];
  int s_prow = 0, s_ih0 = 0, s_iw0 = 0;
  bool s_in = false;
  auto setup_tile = [&](int v) {
    const int bm = fdiv(real_item(v), fd.tn);
    int lo, P;
    item_range_dev(bm, T, TH, TW, H, W, fd, lo, P);
    const int g = bm * TPB + tg * 16 + (lane & 15);
    const int gg = g < T ? g : 0;
    const int img = fdiv(gg, fd.per), r = gg - img * (TH * TW);
    const int th = fdiv(r, fd.tw), tw = r - th * TW;
    s_ih0 = 2 * th - 1;
    s_iw0 = 2 * tw - 1;
    s_prow = (img * H + s_ih0) * W + s_iw0 - lo;
    s_in = g < T;
  };
  auto setup_doff = [&]() {
#pragma unroll
    for (int p = 0; p < 12; ++p) {
      const int pr = role ? 3 - p / 4 : p / 4;  // pixel row of E_(p/4)
      const bool ok = s_in & ((unsigned)(s_ih0 + pr) < (unsigned)H) & ((unsigned)(s_iw0 + p % 4) < (unsigned)W);
      doff[p] = (kg * Pp + (ok ? s_prow + pr * W + (p % 4) : ZS)) * 4;
    }
  };
  floatx4 d[12], V[8];
  auto read_d = [&](int p0, int p1) {
#pragma unroll
    for (int p = p0; p < p1; ++p) d[p] = *reinterpret_cast<const floatx4*>(rawl + doff[p]);
  };
  auto col_xform = [&](const floatx4 (&t)[4], int row) {
#ifdef HP_WABL_NO_XFORM
    V[4 * row + 0] = t[0]; V[4 * row + 1] = t[1]; V[4 * row + 2] = t[2]; V[4 * row + 3] = t[3];
    return;
#endif
    V[4 * row + 0] = sub4s(t[0], t[2]);
    V[4 * row + 1] = add4s(t[1], t[2]);
    V[4 * row + 2] = sub4s(t[2], t[1]);
    V[4 * row + 3] = sub4s(t[1], t[3]);
  };
  // row part of both transforms as soon as the pixels are there (E0, E1, E2 die, the second row
  // waits as 4 float4 for V[4..7] to be consumed); column parts: V[0..3] at once, V[4..7] in step 0
  floatx4 tS[4];
  auto xform_first = [&]() {  // needs E0, E2; E0 dies
    floatx4 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef HP_WABL_NO_XFORM
      t[j] = d[0 + j];
#else
      t[j] = sub4s(d[0 + j], d[8 + j]);
#endif
    }
    col_xform(t, 0);
  };
  auto xform_second_rows = [&]() {  // needs E1, E2; both die
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef HP_WABL_NO_XFORM
      tS[j] = d[4 + j];
#else
      tS[j] = pk_fma4(d[8 + j], s_second2, d[4 + j]);
#endif
    }
  };
  auto xform_second = [&]() { col_xform(tS, 1); };
  floatx4 bf[2][NT];
  setup_tile(item);
  setup_doff();
  read_d(0, 12);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const floatx4*>(ufr_lo + half_off(item) + nt * 16 * 4);
  xform_first();
  xform_second_rows();
  __syncthreads();  // every wave holds its pixels of chunk 0: the pixel buffer may be refilled
  HP_STAMP(1);

  for (;;) {
    const int bm = fdiv(real_item(item), fd.tn), n0 = (real_item(item) - bm * tiles_n) * BN;
    const bool has_next = item + nslot < item_end;
    const int hoff = half_off(item), hoff_next = half_off(has_next ? item + nslot : item);
#ifdef HP_WABL_TIMING
    HP_STAMP(stamp_slot);
#endif
    floatx4 acc[8][NT];

    // one chunk = 8 steps (one position each: 2 weight-fragment reads for the next step, 8 MFMAs);
    // the next stage's LDS stores go under steps 0-3, the loads of the stages after it under
    // steps 4-7, the next chunk's pixel reads under steps 4-7 after the barrier that publishes them
    // FIRST: the chunk that starts the item -- its first MFMA per accumulator takes C = 0 instead of
    // 64 register clears per wave and item
    auto chunk = [&](int c, auto odd, auto is_first, auto is_last) {
      constexpr std::integral_constant<int, 1 - decltype(odd)::value> nxt{};
      constexpr bool FIRST = decltype(is_first)::value;
      constexpr bool LAST = decltype(is_last)::value;
#ifdef HP_WABL_TIMING
      if (stamp_slot == 5) HP_STAMP(32 + c);
#endif
      const float* ub_lo = ufr_lo + ubuf * U_BUF + hoff;
      const float* ub_hi = ufr_hi + ubuf * U_BUF + hoff;
      // bf[0] and V[0..3] of this chunk were prepared under the last step of the previous one (the
      // weights were published by its mid-chunk barrier), so the MFMAs of step 0 issue right away
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q + 1 < 8) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bf[(q + 1) & 1][nt] = *reinterpret_cast<const floatx4*>((q + 1 < 4 ? ub_lo : ub_hi) + ((q + 1) * 128 + nt * 16) * 4);
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bf[0][nt] = *reinterpret_cast<const floatx4*>(ufr_lo + (ubuf ^ 1) * U_BUF + (LAST ? hoff_next : hoff) + nt * 16 * 4);
        }
        if (q == 0) xform_second();
#if !defined(HP_WABL_NO_STAGE) && !defined(HP_WABL_NO_LSTORE)
        if (q < 4) store_held(nxt, ubuf ^ 1, q);
#endif
#if !defined(HP_WABL_NO_STAGE) && !defined(HP_WABL_NO_GLOAD)
        if (q >= 4) {
          issue_raw(nxt, q - 4);
          issue_u(q - 4);
        }
#endif
        if (LAST) {  // tile of the NEXT item (after the block's last item: the same again, never used)
          if (q == 2) setup_tile(has_next ? item + nslot : item);
          if (q == 3) setup_doff();
        }
#ifndef HP_WABL_NO_BARRIER
        if (q == 4) __syncthreads();  // the next stage is in LDS
#endif
#ifndef HP_WABL_NO_READD
        if (q == 4) read_d(0, 4);     // E0
        if (q == 5) read_d(8, 12);    // E2
        if (q == 6) read_d(4, 8);     // E1
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[q & 1][nt][j], V[q][j],  // D[cout][tile]
                                                              FIRST && j == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[q][nt], 0, 0, 0);
        if (q == 7) {  // V[0..3] of the next chunk (this chunk's were consumed by steps 0-3) and the rows of V[4..7]
          xform_first();
          xform_second_rows();
        }
        if (PRE && q == 4) load_pre(c + 2 < nchunks ? c + 2 : c + 2 - nchunks);  // stage stored during the next chunk
        __builtin_amdgcn_sched_barrier(0);
      }
      ubuf ^= 1;
#ifndef HP_WABL_NO_BARRIER
      __syncthreads();  // pixels of the next chunk held by every wave, this chunk's weights consumed
#endif
    };
    constexpr std::false_type NO{};
    constexpr std::true_type YES{};
    chunk(0, SET_A, YES, NO);  // nchunks >= 4 (the launcher sends Cin = 32 to the other kernel)
    chunk(1, SET_B, NO, NO);
    for (int c = 2; c + 2 < nchunks; c += 2) {
      chunk(c, SET_A, NO, NO);
      chunk(c + 1, SET_B, NO, NO);
    }
    chunk(nchunks - 2, SET_A, NO, NO);
    chunk(nchunks - 1, SET_B, NO, YES);
#ifdef HP_WABL_TIMING
    HP_STAMP(stamp_slot + 1);
#endif

    // ---- where the lane's results go.  The weights are the A operand of the MFMAs, so a lane's four
    //      accumulator elements are 4 CONSECUTIVE COUTS (4 kg .. 4 kg + 3 of the 16-cout tile) of ONE
    //      tile (lane & 15): after the output transform it holds the 2x2 pixels of that tile for those
    //      couts and stores them as 16-B accesses with no cross-lane transpose.  Whole items: all four
    //      pixels for the cout half nt = role; half items: pixels 2 role, 2 role + 1 of the one cout
    //      tile.  The residual is requested NOW, so that its latency (HBM for the 60x80 layers) passes
    //      under the output transform and the exchange instead of inside the store loop.
    constexpr int NPX = 2 * NT;  // pixels this wave finishes
    const int ncol = n0 + (NT == 2 ? role * 16 : hoff / 4) + 4 * kg;
    int e_off[NPX];
    bool e_ok[NPX];
    floatx4 e_res[NPX];
    {
      const int g = bm * TPB + tg * 16 + (lane & 15);
      const int gc = g < T ? g : 0;
      const int e_img = fdiv(gc, fd.per);
      const int r = gc - e_img * (TH * TW);
      const int e_th = fdiv(r, fd.tw), e_tw = r - e_th * TW;
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int px = NT == 2 ? i : 2 * role + i;
        const int oh = 2 * e_th + (px >> 1), ow = 2 * e_tw + (px & 1);
        e_ok[i] = (g < T) & (oh < a.Ho) & (ow < a.Wo);
        e_off[i] = ((e_img * a.Ho + oh) * a.Wo + ow) * a.Cout + ncol;  // < 2^31: conv_wino_launchable
        e_res[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        // (not in the PRE instantiations: they are at the register limit, and the layers that take the
        // BN+ReLU prologue -- the first conv of a residual block -- have no residual input)
        if (!PRE && a.residual && e_ok[i]) e_res[i] = *reinterpret_cast<const floatx4*>(a.residual + e_off[i]);
      }
    }
    // ---- output transform.  Local rows L0 = acc[0..3], L1 = acc[4..7]:
    //      role 0: L0 = M row 0, L1 = M row 1   ->  partial of Y row 0: L0 + L1,  of Y row 1: L1
    //      role 1: L0 = -M row 3, L1 = M row 2  ->  partial of Y row 0: L1,       of Y row 1: L0 - L1
    //      then the column transform; the partners swap what they do not finish through LDS.
    //      (two copies of the code behind a wave-uniform branch: with the role known at compile
    //      time half of the row arithmetic and all keep/send selects disappear)
    //      All of it on the register PAIRS (couts c, c+1) the MFMA results already sit in, with
    //      explicit packed adds: left to itself hipcc SLP-packs these adds across other dimensions
    //      and spends more v_mov on building the pairs than it saves.
    floatx2 yk[2][NPX];  // [cout pair][pixel this wave finishes]
    auto partial = [&](auto role_c) {
      constexpr int R = decltype(role_c)::value;
      auto half = [&](auto nt_c, floatx2(&y)[2][4]) {
        constexpr int nt = decltype(nt_c)::value;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          floatx2 p0[4], p1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const floatx2 l0 = ip ? acc[j][nt].zw : acc[j][nt].xy, l1 = ip ? acc[4 + j][nt].zw : acc[4 + j][nt].xy;
            p0[j] = R == 0 ? add2(l0, l1) : l1;
            p1[j] = R == 0 ? l1 : sub2(l0, l1);
          }
          y[ip][0] = add2(add2(p0[0], p0[1]), p0[2]);
          y[ip][1] = sub2(sub2(p0[1], p0[2]), p0[3]);
          y[ip][2] = add2(add2(p1[0], p1[1]), p1[2]);
          y[ip][3] = sub2(sub2(p1[1], p1[2]), p1[3]);
        }
      };
      if constexpr (NT == 2) {
        {  // the half the partner finishes goes out first, so its registers are free again
          floatx2 ys[2][4];
          half(std::integral_constant<int, 1 - R>{}, ys);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            *reinterpret_cast<floatx2*>(xl + ((wave * 8 + k) * 64 + lane) * 2) = ys[k >> 2][k & 3];
        }
        __builtin_amdgcn_sched_barrier(0);
        half(std::integral_constant<int, R>{}, yk);
      } else {  // half item: one cout tile; role r finishes pixels 2r, 2r + 1 and sends the other two
        floatx2 y[2][4];
        half(std::integral_constant<int, 0>{}, y);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<floatx2*>(xl + ((wave * 8 + k) * 64 + lane) * 2) = y[k >> 1][2 * (1 - R) + (k & 1)];
          yk[k >> 1][k & 1] = y[k >> 1][2 * R + (k & 1)];
        }
      }
    };
    if (role == 0) partial(std::integral_constant<int, 0>{});
    else partial(std::integral_constant<int, 1>{});
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2 * NPX; ++k)  // slot k = [cout pair k / NPX][pixel k % NPX]: the layout both sides wrote
      yk[k / NPX][k % NPX] = add2(yk[k / NPX][k % NPX], *reinterpret_cast<const floatx2*>(xl + (((wave ^ 4) * 8 + k) * 64 + lane) * 2));
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      if (e_ok[i]) {
        floatx4 v = {yk[0][i].x, yk[0][i].y, yk[1][i].x, yk[1][i].y};
        if (a.bias) v += *reinterpret_cast<const floatx4*>(a.bias + ncol);
        if (PRE) { if (a.residual) v += *reinterpret_cast<const floatx4*>(a.residual + e_off[i]); }
        else v += e_res[i];
        if (a.relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        *reinterpret_cast<floatx4*>(a.y + e_off[i]) = v;
      }
    }
#ifdef HP_WABL_TIMING
    HP_STAMP(stamp_slot + 2);
    if (stamp_slot == 2 && threadIdx.x == 0) g_wino_stamps[blockIdx.x * kStampSlots + 63] = clock64();
    stamp_slot += 3;
#endif
    item += nslot;
    if (item >= item_end) break;
  }
}

// w [Cout][Kpad] with K ordered (kh, kw, c) (BN already folded)  ->
// U [Cin/16][16 positions][Cout/32][4 kg][32 couts][4 channels],  U = G g G^T,
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void wino_weight_transform(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Kpad) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cout * Cin) return;
  const int o = (int)(idx / Cin), ci = (int)(idx % Cin);
  float g[3][3];
  for (int y = 0; y < 3; ++y)
    for (int x = 0; x < 3; ++x) g[y][x] = w[(int64_t)o * Kpad + (y * 3 + x) * Cin + ci];
  float tg[4][3];
  for (int x = 0; x < 3; ++x) {
    tg[0][x] = g[0][x];
    tg[1][x] = 0.5f * (g[0][x] + g[1][x] + g[2][x]);
    tg[2][x] = 0.5f * (g[0][x] - g[1][x] + g[2][x]);
    tg[3][x] = g[2][x];
  }
  const int chunk = ci / CK, kgi = (ci % CK) / 4, j4 = ci % 4, nb = o / BN, n = o % BN;
  for (int i = 0; i < 4; ++i) {
    const float u[4] = {tg[i][0], 0.5f * (tg[i][0] + tg[i][1] + tg[i][2]), 0.5f * (tg[i][0] - tg[i][1] + tg[i][2]), tg[i][2]};
    for (int j = 0; j < 4; ++j)
      U[(((((int64_t)chunk * 16 + (4 * i + j)) * (Cout / BN) + nb) * 4 + kgi) * BN + n) * 4 + j4] = u[j];
  }
}

struct WinoGeom { int TH, TW, T, Pmax; bool ok; };

// tile counts and the largest staged pixel range of any item (the pattern of item starts
// repeats after TH*TW items, so that many, plus the ragged last one, are enough to look at)
WinoGeom wino_geom(int H, int W, int64_t n_img) {
  WinoGeom g{};
  g.TH = (H + 1) / 2; g.TW = (W + 1) / 2;
  const int64_t T64 = n_img * g.TH * g.TW;
  g.ok = T64 > 0 && T64 < (1 << 30) && n_img * H * W < (1ll << 30);
  if (!g.ok) return g;
  g.T = (int)T64;
  const int tiles_m = (g.T + TPB - 1) / TPB;
  const int look = std::min(tiles_m, g.TH * g.TW + 1);
  int lo, P;
  for (int bm = 0; bm < look; ++bm) {
    item_range(bm, g.T, g.TH, g.TW, H, W, lo, P);
    g.Pmax = std::max(g.Pmax, P);
  }
  item_range(tiles_m - 1, g.T, g.TH, g.TW, H, W, lo, P);
  g.Pmax = std::max(g.Pmax, P);
  return g;
}

constexpr int kMaxNld = 12;  // staged pixel range <= 768 pixels
size_t wino_lds_bytes(int Pmax) { return ((size_t)4 * plane_len(Pmax) * 4 + 2 * (size_t)U_BUF) * sizeof(float); }

template <bool PRE, int NLD>
int launch(const ConvArgs& a, const WinoGeom& g, hipStream_t stream) {
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch&) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<PRE, NLD>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return HP_OK;
      }))
    return rc0;
  const int cus = conv_num_cus();
  const int tiles_m = (g.T + TPB - 1) / TPB, tiles_n = a.Cout / BN;
  const int n_items = tiles_m * tiles_n;
  // one persistent block per CU, an equal number per XCD
  const int ipx = (n_items + 7) / 8;
  const int slots = std::max(1, std::min(ipx, cus / 8));
  hipLaunchKernelGGL((conv3x3_wino_f32<PRE, NLD>), dim3(8 * slots), dim3(kThreads), wino_lds_bytes(g.Pmax), stream, a,
                     g.TH, g.TW, g.T, tiles_n, n_items, g.Pmax,
                     WinoDiv{make_fastdiv((unsigned)(g.TH * g.TW)), make_fastdiv((unsigned)g.TW), make_fastdiv((unsigned)tiles_n)});
  return check_launch("conv3x3_wino_f32");
}

int wino8_nld(int Pmax) { return std::max(2, (Pmax + 127) / 128); }
size_t wino8_lds_bytes(int Pmax, int Cin = 512) {
  return ((size_t)4 * plane_len8(wino8_nld(Pmax)) * 4 + 2 * (size_t)U_BUF + (size_t)X_BUF + 2 * (size_t)Cin) * sizeof(float);
}

template <bool PRE, int NLD>
int launch8(const ConvArgs& a, const WinoGeom& g, hipStream_t stream) {
  constexpr int SETS = NLD >= 6 ? 1 : 2;
  static FirstLaunch fl;
  if (const int rc0 = fl.once([](FirstLaunch&) {
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino8_f32<PRE, NLD, SETS, 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino8_f32<PRE, NLD, SETS, 1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return HP_OK;
      }))
    return rc0;
  const int cus = conv_num_cus();
  const int tiles_m = (g.T + TPB - 1) / TPB, tiles_n = a.Cout / BN;
  const int n_items = tiles_m * tiles_n;
  const int ipx = (n_items + 7) / 8;  // items per XCD
  const int per_xcd = std::max(1, cus / 8);
  const WinoDiv fd{make_fastdiv((unsigned)(g.TH * g.TW)), make_fastdiv((unsigned)g.TW), make_fastdiv((unsigned)tiles_n)};
  const size_t lds = wino8_lds_bytes(g.Pmax, a.Cin);
  // Rounds: every block of an XCD walks ipx / per_xcd items.  A last round that fills at most half of
  // the blocks (8x10 maps at batch 128: 2.5 rounds) would cost a whole round; its items are split into
  // two 16-cout halves instead and run as a second launch on twice as many blocks -- a half item costs
  // ~0.6 of a whole one (half the MFMAs, the same staging).
  const int full = (ipx / per_xcd) * per_xcd, tail = ipx - full;
  if (full > 0 && tail > 0 && 2 * tail <= per_xcd) {
    hipLaunchKernelGGL((conv3x3_wino8_f32<PRE, NLD, SETS, 2>), dim3(8 * per_xcd), dim3(kThreads8), lds, stream, a, g.TH, g.TW,
                       g.T, tiles_n, n_items, g.Pmax, fd, 0, full);
    hipLaunchKernelGGL((conv3x3_wino8_f32<PRE, NLD, SETS, 1>), dim3(8 * 2 * tail), dim3(kThreads8), lds, stream, a, g.TH, g.TW,
                       g.T, tiles_n, n_items, g.Pmax, fd, full, tail);
  } else {
    const int slots = std::max(1, std::min(ipx, per_xcd));
    hipLaunchKernelGGL((conv3x3_wino8_f32<PRE, NLD, SETS, 2>), dim3(8 * slots), dim3(kThreads8), lds, stream, a, g.TH, g.TW,
                       g.T, tiles_n, n_items, g.Pmax, fd, 0, ipx);
  }
  return check_launch("conv3x3_wino8_f32");
}

// the one-wave-per-SIMD schedule on request (hp_conv_select_algo / HP_WINO_V1), else two waves per SIMD
bool wino_use_v1(const ConvArgs& a) { return a.algo == HP_CONV_ALGO_WINOGRAD_1WAVE; }

template <bool PRE>
int launch_nld(const ConvArgs& a, const WinoGeom& g, hipStream_t stream) {
  if (!wino_use_v1(a) && a.Cin >= 4 * CK && wino8_nld(g.Pmax) <= 6 && wino8_lds_bytes(g.Pmax, a.Cin) <= 160 * 1024) {
    const int nld8 = wino8_nld(g.Pmax);
    if (nld8 <= 2) return launch8<PRE, 2>(a, g, stream);
    if (nld8 <= 3) return launch8<PRE, 3>(a, g, stream);
    if (nld8 <= 4) return launch8<PRE, 4>(a, g, stream);
    if (nld8 <= 5) return launch8<PRE, 5>(a, g, stream);
    return launch8<PRE, 6>(a, g, stream);
  }
  const int nld = (g.Pmax * 4 + kThreads - 1) / kThreads;
  if (nld <= 4) return launch<PRE, 4>(a, g, stream);
  if (nld <= 6) return launch<PRE, 6>(a, g, stream);
  if (nld <= 8) return launch<PRE, 8>(a, g, stream);
  if (nld <= 10) return launch<PRE, 10>(a, g, stream);
  return launch<PRE, 12>(a, g, stream);
}

}  // namespace

bool conv_wino_applicable(const ConvArgs& a, int kh, int kw) {
  if (kh != 3 || kw != 3 || a.stride != 1 || a.pad != 1 || a.Cin % (2 * CK) != 0 || a.Cout % BN != 0) return false;
  if (a.H < 2 || a.W < 2 || a.Ho != a.H || a.Wo != a.W) return false;
  // the staged pixel range of 64 consecutive tiles must fit LDS next to the weight buffers
  // (batch independent once there are a few images: look at a long virtual batch)
  const WinoGeom g = wino_geom(a.H, a.W, 64);
  return g.ok && g.Pmax <= kMaxNld * 64 && wino_lds_bytes(g.Pmax) <= 160 * 1024;
}

size_t conv_wino_weight_floats(int cout, int cin) { return (size_t)16 * cout * cin; }

int conv_wino_transform_weights(const float* d_w, float* d_U, int cout, int cin, int Kpad, hipStream_t stream) {
  const int64_t n = (int64_t)cout * cin;
  hipLaunchKernelGGL(wino_weight_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_w, d_U, cout, cin, Kpad);
  return check_launch("wino_weight_transform");
}

// geometry of the last few (H, W, batch) seen: the look-ahead loop of wino_geom is not free
const WinoGeom& cached_geom(int H, int W, int64_t n_img) {
  struct Entry { int H, W; int64_t n; WinoGeom g; };
  static Entry cache[16];
  static int used = 0, next = 0;
  for (int i = 0; i < used; ++i)
    if (cache[i].H == H && cache[i].W == W && cache[i].n == n_img) return cache[i].g;
  const int slot = used < 16 ? used++ : (next++ % 16);
  cache[slot] = Entry{H, W, n_img, wino_geom(H, W, n_img)};
  return cache[slot].g;
}

// batch-dependent part of the applicability test: 32-bit buffer offsets, staged range fits
bool conv_wino_launchable(const ConvArgs& a) {
  const int64_t n_img = a.M / ((int64_t)a.Ho * a.Wo);
  if (n_img * a.H * a.W * a.Cin * 4 >= (1ll << 31)) return false;  // 32-bit buffer offsets
  if (a.M * a.Cout >= (1ll << 31)) return false;                  // 32-bit output element offsets
  const WinoGeom& g = cached_geom(a.H, a.W, n_img);
  return g.ok && g.Pmax <= kMaxNld * 64 && wino_lds_bytes(g.Pmax) <= 160 * 1024;
}

// a.w must point at the transformed weights U
int launch_conv_wino(const ConvArgs& a, hipStream_t stream) {
  if (!conv_wino_launchable(a))
    return fail(HP_ERR_ARG, "conv3x3_wino_f32: geometry not supported (check conv_wino_launchable)");
  const WinoGeom& g = cached_geom(a.H, a.W, a.M / ((int64_t)a.Ho * a.Wo));
  return a.pre_scale ? launch_nld<true>(a, g, stream) : launch_nld<false>(a, g, stream);
}

}  // namespace hp

#ifdef HP_WABL_TIMING
extern "C" int hp_debug_wino_stamps(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hp::g_wino_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#endif
