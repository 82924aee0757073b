// Order-preserving (stable) stream compaction building blocks.
//
// Surfel ids are buffer positions and association keys, so every pass that removes or appends
// surfels must keep their relative order (the reference gets this from OpenGL transform
// feedback).  Three stream-ordered launches:
//   flags  : each 256-thread block owns kScanChunk = 256 consecutive elements, writes one
//            keep-byte per element and the block's kept count;
//   scan   : one 1024-thread block turns the per-block counts into exclusive offsets and
//            publishes the total;
//   scatter: each block re-reads its keep-bytes, ranks them with wave ballots and writes the
//            survivors at offset + rank.
#pragma once
#include <hip/hip_runtime.h>

namespace dms {

// sum of one unsigned per thread over a 256-thread block; result valid in every thread
__device__ __forceinline__ unsigned block_sum_u32(unsigned v) {
  __shared__ unsigned s_w[4];
  __shared__ unsigned s_tot;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (lane == 0) s_w[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) s_tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  const unsigned t = s_tot;
  __syncthreads();
  return t;
}

// exclusive rank of `flag` among the 256 threads of the block (thread order), and the total
__device__ __forceinline__ unsigned block_exclusive_rank(bool flag, unsigned& total) {
  __shared__ unsigned s_w[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long mask = __ballot(flag);
  const unsigned below = __popcll(mask & ((1ull << lane) - 1ull));
  __syncthreads();  // protect s_w from the previous call's readers
  if (lane == 0) s_w[wid] = __popcll(mask);
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wid; ++w) base += s_w[w];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + below;
}

// exclusive scan of nb per-block counts by one 1024-thread block; out_count[0] = min(total, cap)
static __global__ __launch_bounds__(1024) void k_scan_blocks(const unsigned* __restrict__ counts, unsigned* __restrict__ offsets, int nb,
                                                      unsigned* __restrict__ out_count, unsigned cap, unsigned* __restrict__ out_count2) {
  __shared__ unsigned s_w[16];
  __shared__ unsigned s_carry;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned v = i < nb ? counts[i] : 0u;
    unsigned incl = v;  // inclusive scan inside the wave
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < wid; ++w) wbase += s_w[w];
    const unsigned carry = s_carry;
    if (i < nb) offsets[i] = carry + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const unsigned c = s_carry < cap ? s_carry : cap;
    out_count[0] = c;
    if (out_count2) out_count2[0] = c;  // e.g. the frame context's result block
  }
}

// The same scan for long count arrays (maps of a million surfels and more), one 1024-thread block per 1024
// counts: each block first sums every count before its chunk itself (no inter-block wait; the counts
// are a few hundred kilobytes in L2), then scans its own chunk.  The last block stores the total.
// `first_touched` (optional): atomicMin of the first block index b that does not satisfy
// counts[b] == full && (b + 1) * full <= limit[0] — the end of the prefix that the suffix-mode clean
// leaves in place; the word is set to 0xFFFFFFFF by an earlier kernel on the stream.
static __global__ __launch_bounds__(1024) void k_scan_blocks_par(const unsigned* __restrict__ counts, unsigned* __restrict__ offsets, int nb,
                                                          unsigned* __restrict__ out_count, unsigned cap,
                                                          unsigned* __restrict__ out_count2, unsigned* __restrict__ first_touched,
                                                          const unsigned* __restrict__ limit, unsigned full) {
  __shared__ unsigned s_w[16];
  __shared__ unsigned s_prefix;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int chunk0 = blockIdx.x * 1024;
  unsigned below = 0;
  for (int i = threadIdx.x; i < chunk0; i += 1024) below += counts[i];
  for (int off = 32; off > 0; off >>= 1) below += __shfl_down(below, off, 64);
  if (lane == 0) s_w[wid] = below;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < 16; ++w) t += s_w[w];
    s_prefix = t;
  }
  __syncthreads();
  const unsigned prefix = s_prefix;
  __syncthreads();
  const int i = chunk0 + threadIdx.x;
  const unsigned v = i < nb ? counts[i] : 0u;
  unsigned incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_w[wid] = incl;
  __syncthreads();
  unsigned wbase = 0;
  for (int w = 0; w < wid; ++w) wbase += s_w[w];
  if (i < nb) {
    offsets[i] = prefix + wbase + incl - v;
    if (first_touched) {
      const bool in_place = v == full && ((unsigned)i + 1u) * full <= limit[0];
      if (!in_place) atomicMin(first_touched, (unsigned)i);
    }
  }
  if (i == nb - 1) {
    const unsigned tot = prefix + wbase + incl;
    const unsigned c = tot < cap ? tot : cap;
    out_count[0] = c;
    if (out_count2) out_count2[0] = c;
  }
}

}  // namespace dms
