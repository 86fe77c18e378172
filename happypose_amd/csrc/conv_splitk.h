// Tail split-K of the conv kernels: the tiles of the last, partially filled round of the grid are
// split along K into slices that run as separate workgroups; slices meet through a "last arriver
// reduces" protocol (cdna_hip_programming.md G16): each slice parks its partial accumulators in a
// slab (coalesced 16 B per lane), publishes with an agent-scope release + ticket, and the slice that
// takes the last ticket acquires and sums ALL slabs in slice order -- bitwise reproducible whoever
// arrives last -- before running the epilogue.  The planning side is conv_plan_split (conv_patch.hip).
#pragma once

#include "conv.h"

namespace hp {

typedef float sk_floatx16 __attribute__((ext_vector_type(16)));
typedef float sk_floatx4 __attribute__((ext_vector_type(4)));

// returns true in the workgroup that must run the epilogue (acc then holds the full sum)
template <int BM, int BN, int MT, int NT, int THREADS>
__device__ __forceinline__ bool splitk_reduce(const ConvArgs& a, sk_floatx16 (&acc)[MT][NT], int tail_tile, int slice) {
  __shared__ int ticket_s;
  const int tid = threadIdx.x;
  float* const slab = a.sk_slabs + ((size_t)tail_tile * a.sk_S + slice) * (BM * BN);
  // register layout -> [mt][nt][r4][thread][4]: every store is a coalesced 16 B per lane
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        sk_floatx4 v = {acc[mt][nt][4 * r4], acc[mt][nt][4 * r4 + 1], acc[mt][nt][4 * r4 + 2], acc[mt][nt][4 * r4 + 3]};
        *reinterpret_cast<sk_floatx4*>(slab + ((((mt * NT + nt) * 4 + r4) * THREADS) + tid) * 4) = v;
      }
  // publish: stores drained -> agent-scope release -> ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ticket_s = __hip_atomic_fetch_add(a.sk_counters + tail_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (ticket_s != a.sk_S - 1) return false;  // not the last slice of this tile
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // every slice has taken its ticket: re-arm for the next launch.  An AGENT-scope atomic store like the tickets themselves:
    // the slices' atomics run at the memory side, a plain store sits in THIS XCD's L2 until something writes it back -- between
    // two launches replayed from a hipGraph nothing did, the next launch's tickets then started from sk_S, no slice saw
    // sk_S - 1, the tile's epilogue never ran and its rows kept the previous forward's values (round 5: the co-scheduling
    // "non-determinism" of two-lane graph replays, tools/probes/two_lane_repro.py)
    __hip_atomic_store(a.sk_counters + tail_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int s2 = 0; s2 < a.sk_S; ++s2) {
    const float* other = a.sk_slabs + ((size_t)tail_tile * a.sk_S + s2) * (BM * BN);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const sk_floatx4 v = *reinterpret_cast<const sk_floatx4*>(other + ((((mt * NT + nt) * 4 + r4) * THREADS) + tid) * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[mt][nt][4 * r4 + q] += v[q];
        }
  }
  return true;
}

// The same hand-off with write-through (sc1) slab stores and sc1 slab loads instead of plain accesses bracketed by
// agent-scope release / acquire fences (cdna_hip_programming.md, split-K recipe): the release fence writes back every
// dirty line of the XCD's L2 -- mostly other workgroups' output -- once per slice, which cost more than the slices saved
// on the 8x10 layers.  Slab offsets are 32-bit (the planner keeps the slab area below 2 GB).
template <int BM, int BN, int MT, int NT, int THREADS>
__device__ __forceinline__ bool splitk_reduce_sc1(const ConvArgs& a, sk_floatx16 (&acc)[MT][NT], int tail_tile, int slice) {
  typedef unsigned int sk_uintx4 __attribute__((ext_vector_type(4)));
  __shared__ int ticket_s1;
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.sk_slabs, 0, 0x7FFFFFFF, 0x00020000);
  const unsigned slab = (unsigned)(((size_t)tail_tile * a.sk_S + slice) * (BM * BN) * 4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const sk_floatx4 v = {acc[mt][nt][4 * r4], acc[mt][nt][4 * r4 + 1], acc[mt][nt][4 * r4 + 2], acc[mt][nt][4 * r4 + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_uintx4, v), rsrc,
                                               (int)(slab + ((((mt * NT + nt) * 4 + r4) * THREADS) + tid) * 16), 0, 16 /* sc1 */);
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) ticket_s1 = __hip_atomic_fetch_add(a.sk_counters + tail_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket_s1 != a.sk_S - 1) return false;  // not the last slice of this tile
  if (tid == 0) __hip_atomic_store(a.sk_counters + tail_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm (an atomic: see splitk_reduce)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int s2 = 0; s2 < a.sk_S; ++s2) {
    const unsigned other = (unsigned)(((size_t)tail_tile * a.sk_S + s2) * (BM * BN) * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const sk_floatx4 v = __builtin_bit_cast(sk_floatx4, __builtin_amdgcn_raw_buffer_load_b128(
              rsrc, (int)(other + ((((mt * NT + nt) * 4 + r4) * THREADS) + tid) * 16), 0, 16 /* sc1 */));
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[mt][nt][4 * r4 + q] += v[q];
        }
  }
  return true;
}

// block index -> work item: per XCD (dispatch puts block b on XCD b % 8) first its share of the
// regular tiles (whole K), then its share of the tail items (tile, slice).  Returns false for
// the padding blocks of the grid.
__device__ __forceinline__ bool splitk_decode(const ConvArgs& a, int& tile, int& slice, bool& split) {
  const int rpx = a.sk_regular / 8, tpx = (a.sk_tail_items + 7) / 8;
  const int xcd = blockIdx.x % 8, li = blockIdx.x / 8;
  slice = 0; split = false;
  if (li < rpx) { tile = xcd * rpx + li; return true; }
  const int ti = xcd * tpx + (li - rpx);
  if (li - rpx >= tpx || ti >= a.sk_tail_items) return false;
  tile = a.sk_regular + ti / a.sk_S;
  slice = ti % a.sk_S;
  split = a.sk_S > 1;
  return true;
}

// host: fill a.sk_* for T tiles whose K loop has k_units splittable units of ktiles_per_unit K-tiles
// (32 deep) each; lds_bytes decides whether one or two workgroups fit a CU.
// Grid = 8 * (sk_regular / 8 + ceil(sk_tail_items / 8)) blocks.
int conv_plan_split(ConvArgs& a, int T, size_t lds_bytes, int k_units, int ktiles_per_unit, hipStream_t stream);

}  // namespace hp
