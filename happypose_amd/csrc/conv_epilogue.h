// Shared epilogue of the MFMA conv kernels: bias (folded BN), residual add, ReLU, NHWC store.
//
// The 32x32 MFMA accumulator layout gives a lane 16 values of ONE output channel (column) in
// 16 different pixels (rows), so a direct store is 4 bytes per lane per instruction and the
// residual read likewise.  Instead the block tile is transposed through LDS (free at this
// point): accumulators -> LDS [row][BN+4] with ds_write_b32 (conflict free: a half-wave writes
// 32 consecutive columns of a row), then every thread moves 16-B pieces: ds_read_b128, float4
// bias, float4 residual load, ReLU, float4 global store -- 4x fewer and fully coalesced
// vector-memory instructions (a 128-channel row = 512 contiguous bytes per 32 lanes).
#pragma once

#include "conv.h"

namespace hp {

typedef float epi_floatx16 __attribute__((ext_vector_type(16)));
typedef float epi_floatx4 __attribute__((ext_vector_type(4)));

// Non-finite guard of the split-fp16 kernels (ConvArgs::status): an activation beyond the fp16 range becomes inf when
// it is split and inf / NaN in the accumulators; ReLU would hide a NaN (fmaxf(NaN, 0) = 0), so the epilogues sum what
// they are about to store BEFORE the activation and report a non-finite sum.  The store only happens in the failure case.
__device__ __forceinline__ void conv_report_nonfinite(const ConvArgs& a, float chk) {
  if (a.status && !(fabsf(chk) <= 3.0e38f)) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int BM, int BN>
constexpr int epilogue_lds_floats() { return BM * (BN + 4); }

template <int BM, int BN, int MT, int NT, int THREADS>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, float* lds, epi_floatx16 (&acc)[MT][NT],
                                              int64_t m0, int n0, int wm, int wn) {
  constexpr int LDC = BN + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int C4 = BN / 4;                       // 16-B pieces per row
  constexpr int ITERS = BM * C4 / THREADS;
  const int c4 = tid % C4;
  const int n = n0 + 4 * c4;
  const bool n_ok = n < a.Cout;  // the last tile of a layer whose Cout is not a multiple of BN
  // the residual pieces of this thread are fetched FIRST, all of them, so that their latency runs under the LDS
  // transpose; fetched inside the store loop they were ITERS dependent round trips to HBM (60x80 layers with a
  // residual: 224 -> 170 us)
  epi_floatx4 res[ITERS];
  if (a.residual) {
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const int64_t m = m0 + tid / C4 + k * (THREADS / C4);
      const bool ok = m < a.M && n_ok;
      res[k] = *reinterpret_cast<const epi_floatx4*>(a.residual + (ok ? m * a.Cout + n : 0));
    }
  }
  __syncthreads();  // every wave is done reading the operand tiles
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        lds[row * LDC + wn + nt * 32 + (lane & 31)] = acc[mt][nt][r];
      }
  __syncthreads();
  epi_floatx4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias = *reinterpret_cast<const epi_floatx4*>(a.bias + n);  // padded to whole tiles by the planner
  float chk = 0.f;  // running sum of everything this thread stores, before the activation: inf / NaN are sticky in it
  float amax = 0.f; // largest magnitude this thread stores (ConvArgs::amax_out)
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const int row = tid / C4 + k * (THREADS / C4);
    const int64_t m = m0 + row;
    if (m < a.M && n_ok) {
      epi_floatx4 v = *reinterpret_cast<const epi_floatx4*>(lds + row * LDC + 4 * c4);
      v += bias;
      if (a.residual) v += res[k];
      chk += (v[0] + v[1]) + (v[2] + v[3]);
      if (a.relu == HP_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
      } else if (a.relu == HP_ACT_SWISH) {  // x * sigmoid(x)  (MemoryEfficientSwish, CP/models/efficientnet_utils.py)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.f + __expf(-v[q]));  // v_rcp_f32 (1 ulp), as mbconv_front.hip:
        // the IEEE division was ~10 of the ~15 VALU instructions per stored element, and EfficientNet's wide expansions are
        // bound by exactly those (SQ counters, round 5: VALU active 0.24 of the wave cycles at 3 - 4 waves per SIMD)
      }
#ifndef HP_EABL_NO_AMAX_VALU
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#endif
      *reinterpret_cast<epi_floatx4*>(a.y + m * a.Cout + n) = v;
    }
  }
  conv_report_nonfinite(a, chk);
#ifndef HP_EABL_NO_AMAX_REDUCE
  if (a.amax_out) {
    // max is order-independent: deterministic.  The word only grows, so a wave first LOOKS (a plain device-scope load)
    // and sends its atomic only when it would raise the value: after the first few tiles nearly every wave skips it --
    // thousands of same-address atomics per launch serialise in one L2 channel (+5 us per launch, measured)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0 && amax > 0.f) {
      const unsigned mine = __float_as_uint(amax);
      unsigned* const slot = a.amax_out + (blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride;  // spread over L2 channels
      if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
    }
  }
#endif
}

}  // namespace hp
