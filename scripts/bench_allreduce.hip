// Microbenchmark of the grid-wide "sum 58 values over all blocks, every block gets the totals" step of the
// resident tracker kernels (track.hip, k_gn_level), 512-thread blocks, at most one per CU.
//
//   mode 0  records + barrier + gather (round-1 protocol): every block stores a 256-byte fp32 record (sc1), drains,
//           arrives at a sharded payload barrier whose last shard arriver forwards to a top word, polls the top word,
//           then gathers ALL records with 16-byte sc1 loads.
//   mode 1  the same, but nobody forwards: pollers read the 8 shard words themselves (one hop less).
//   mode 2  integer all-reduce in memory-side atomics: lane k of wave 0 adds its value (fixed point, low 56 bits)
//           plus one arrival (top 8 bits) to word [k][block % 8] with a non-returning 64-bit atomic; lane k then
//           polls its own 8 shard words until they show all arrivals — the totals are then in its registers.
//           No record, no separate barrier, no gather.  Integer adds are order free => deterministic.
//   mode 3  mode 2 with the words of one shard contiguous ([shard][k]) instead of the shards of one value.
//
// Prints microseconds per iteration and checks every total in every block and iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NV = 58;
constexpr int REC = 64;
constexpr int TPB = 512;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__device__ __forceinline__ int value_of(int it, int b, int k) { return ((it * 131 + b * 7 + k * 3) % 2001) - 1000; }

// ---- round-1 barrier (payload word, 8 shards, forward hop) ----
__device__ __forceinline__ void arrive_fwd(u64* b, u64 payload) {
  const u64 one = 1ull << 54, low = one - 1ull;
  const int sh = blockIdx.x & 7;
  const u64 members = (gridDim.x - sh + 7) >> 3;
  const u64 old = __hip_atomic_fetch_add(b + 16 * (sh + 1), one | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((old >> 54) + 1ull == members) __hip_atomic_fetch_add(b, one | (((old & low) + payload) & low), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 wait_fwd(u64* b, int* err) {
  __shared__ u64 s_word;
  if (threadIdx.x == 0) {
    u64 cur;
    unsigned spins = 0;
    for (;;) {
      cur = __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((cur >> 54) >= 8ull) break;
      if (++spins > (1u << 22)) {
        *err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    s_word = cur;
  }
  __syncthreads();
  return s_word;
}
// ---- no forward hop: 8 lanes poll the 8 shard words (8 words of one 64-byte line) ----
__device__ __forceinline__ void arrive_nf(u64* b, u64 payload) {
  __hip_atomic_fetch_add(b + 16 * (blockIdx.x & 7), (1ull << 54) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 wait_nf(u64* b, int* err) {
  __shared__ u64 s_word;
  if (threadIdx.x < 64) {
    unsigned spins = 0;
    for (;;) {
      u64 cur = (threadIdx.x < 8) ? __hip_atomic_load(b + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      // sum over lanes 0..7 (wave-uniform result via readlane)
      u64 tot = 0;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)cur, l), hi = __builtin_amdgcn_readlane((unsigned)(cur >> 32), l);
        tot += ((u64)hi << 32) | lo;
      }
      if ((tot >> 54) >= (u64)gridDim.x) {
        if (threadIdx.x == 0) s_word = tot;
        break;
      }
      if (++spins > (1u << 22)) {
        *err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return s_word;
}

__global__ __launch_bounds__(TPB) void k_records(float* rec, u64* words, int iters, int* err, int* bad_out, int nofwd, int check) {
  __shared__ double s_grp[32][16][4];
  __shared__ float s_sums[REC];
  const int nb = gridDim.x, tid = threadIdx.x;
  int bad = 0;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, nb * REC * 4 * 2, 0x00020000);
  for (int it = 0; it < iters; ++it) {
    const int par = it & 1;
    float* my = rec + ((size_t)par * nb + blockIdx.x) * REC;
    if (tid < NV * 8 && (tid & 7) == 0) __hip_atomic_store(my + (tid >> 3), (float)value_of(it, blockIdx.x, tid >> 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u64* w = words + (size_t)it * 160;
    if (nofwd) {
      if (tid == 0) arrive_nf(w, 1);
      wait_nf(w, err);
    } else {
      if (tid == 0) arrive_fwd(w, 1);
      wait_fwd(w, err);
    }
    const int k4 = tid & 15, g = tid >> 4;
    double f[4] = {0., 0., 0., 0.};
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = g + 32 * u;
      const int bc = b < nb ? b : 0;
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((par * nb + bc) * REC + k4 * 4) * 4, 0, 16);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (g + 32 * u < nb)
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] += (double)__uint_as_float(v[u][c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) s_grp[g][k4][c] = f[c];
    __syncthreads();
    if (tid < REC) {
      double t = 0;
      for (int gg = 0; gg < 32; ++gg) t += s_grp[gg][tid >> 2][tid & 3];
      s_sums[tid] = (float)t;
    }
    __syncthreads();
    if (check && tid < NV) {
      long long expect = 0;
      for (int b = 0; b < nb; ++b) expect += value_of(it, b, tid);
      if ((long long)s_sums[tid] != expect) bad = 1;
    }
  }
  if (bad || s_sums[3] == 123456.f) atomicAdd(bad_out, 1);
}

// layout 0: word(k, sh) = k * 8 + sh (the 8 shards of a value share a 64-byte line); layout 1: sh * 64 + k
template <int NVT, int NS>
__global__ __launch_bounds__(TPB) void k_atomic_t(u64* words, int iters, int* err, int* bad_out, int check) {
  __shared__ long long s_tot[64];
  const int nb = gridDim.x, tid = threadIdx.x;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    u64* w = words + (size_t)it * (NS * 64);
    const int sh = blockIdx.x % NS;
    if (tid < NVT) {
      const long long v = value_of(it, blockIdx.x, tid);
      const u64 add = (u64)(v + (1ll << 46)) | (1ull << 58);
      __hip_atomic_fetch_add(w + sh * 64 + tid, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {
      unsigned spins = 0;
      long long tot = 0;
      for (;;) {
        u64 arr = 0, fld = 0;
        if (tid < NVT) {
          u64 q[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) q[s] = __hip_atomic_load(w + s * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            arr += q[s] >> 58;
            fld += q[s] & ((1ull << 52) - 1ull);
          }
        }
        const bool done = tid >= NVT || arr == (u64)nb;
        if (__builtin_amdgcn_ballot_w64(done) == ~0ull) {
          tot = (long long)fld - (long long)arr * (1ll << 46);
          break;
        }
        if (++spins > (1u << 14)) {
          *err = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      s_tot[tid] = tot;
    }
    __syncthreads();
    if (check && tid < NVT) {
      long long expect = 0;
      for (int b = 0; b < nb; ++b) expect += value_of(it, b, tid);
      if (s_tot[tid] != expect) bad = 1;
    }
    __syncthreads();
  }
  if (bad || s_tot[3] == 123456789) atomicAdd(bad_out, 1);
}

// mode 9/10: the same all-reduce, but only ONE block per XCD (the first to take a ticket on its XCD's election word)
// polls the memory-side words; it hands the totals to the other blocks of its XCD through 8-byte {value, tag} granules
// in a per-XCD mailbox: plain (workgroup-scope) stores keep the lines in that XCD's L2, the followers' sc1 loads
// bypass their L1 and hit it.  Readers per hot memory-side line: 8 instead of every block.
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF); }

template <int NVT>
__global__ __launch_bounds__(TPB) void k_atomic_xcd(u64* words, u64* mailbox, unsigned* elect, int iters, int* err, int* bad_out, int check) {
  __shared__ long long s_tot[64];
  __shared__ int s_leader;
  const int nb = gridDim.x, tid = threadIdx.x;
  const int xcd = xcc_id() & 7;
  if (tid == 0) s_leader = atomicAdd(elect + xcd, 1u) == 0u ? 1 : 0;
  __syncthreads();
  const bool leader = s_leader != 0;
  u64* mb = mailbox + xcd * 64;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    u64* w = words + (size_t)it * (8 * 64);
    const int sh = blockIdx.x & 7;
    const unsigned tag = (unsigned)it + 1u;
    if (tid < NVT) {
      const long long v = value_of(it, blockIdx.x, tid);
      __hip_atomic_fetch_add(w + sh * 64 + tid, (u64)(v + (1ll << 46)) | (1ull << 58), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {
      unsigned spins = 0;
      if (leader) {
        long long tot = 0;
        for (;;) {
          u64 arr = 0, fld = 0;
          if (tid < NVT) {
            u64 q[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) q[s] = __hip_atomic_load(w + s * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              arr += q[s] >> 58;
              fld += q[s] & ((1ull << 52) - 1ull);
            }
          }
          const bool done = tid >= NVT || arr == (u64)nb;
          if (__builtin_amdgcn_ballot_w64(done) == ~0ull) {
            tot = (long long)fld - (long long)arr * (1ll << 46);
            break;
          }
          if (++spins > (1u << 14)) {
            *err = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (tid < NVT) {
          const u64 g = ((u64)tag << 32) | (u64)__float_as_uint((float)tot);
          __hip_atomic_store(mb + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        s_tot[tid] = tot;
      } else {
        u64 g = 0;
        for (;;) {
          if (tid < NVT) g = __hip_atomic_load(mb + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool done = tid >= NVT || (unsigned)(g >> 32) == tag;
          if (__builtin_amdgcn_ballot_w64(done) == ~0ull) break;
          if (++spins > (1u << 16)) {
            *err = 2;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        s_tot[tid] = (long long)__uint_as_float((unsigned)g);
      }
    }
    __syncthreads();
    if (check && tid < NVT) {
      long long expect = 0;
      for (int b = 0; b < nb; ++b) expect += value_of(it, b, tid);
      if (s_tot[tid] != expect) bad = 1;
    }
    __syncthreads();
  }
  if (bad || s_tot[3] == 123456789) atomicAdd(bad_out, 1);
}

__global__ __launch_bounds__(TPB) void k_atomic(u64* words, int iters, int* err, int* bad_out, int layout, int check) {
  __shared__ long long s_tot[64];
  const int nb = gridDim.x, tid = threadIdx.x;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    u64* w = words + (size_t)it * 512;
    const int sh = blockIdx.x & 7;
    if (tid < NV) {
      // word = arrivals (top 6 bits, <= 32 per shard word) | field (58 bits) holding sum of (v + 2^52), |v| < 2^52
      const long long v = value_of(it, blockIdx.x, tid);
      const u64 add = (u64)(v + (1ll << 52)) | (1ull << 58);
      __hip_atomic_fetch_add(w + (layout ? sh * 64 + tid : tid * 8 + sh), add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {
      unsigned spins = 0;
      long long tot = 0;
      for (;;) {
        u64 arr = 0, fld = 0;
        if (tid < NV) {
          u64 q[8];
#pragma unroll
          for (int s = 0; s < 8; ++s) q[s] = __hip_atomic_load(w + (layout ? s * 64 + tid : tid * 8 + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            arr += q[s] >> 58;
            fld += q[s] & ((1ull << 58) - 1ull);
          }
        }
        const bool done = tid >= NV || arr == (u64)nb;
        if (__builtin_amdgcn_ballot_w64(done) == ~0ull) {
          tot = (long long)fld - (long long)arr * (1ll << 52);
          break;
        }
        if (++spins > (1u << 14)) {
          *err = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      s_tot[tid] = tot;
    }
    __syncthreads();
    if (check && tid < NV) {
      long long expect = 0;
      for (int b = 0; b < nb; ++b) expect += value_of(it, b, tid);
      if (s_tot[tid] != expect) bad = 1;
    }
    __syncthreads();
  }
  if (bad || s_tot[3] == 123456789) atomicAdd(bad_out, 1);
}

int main() {
  float* rec;
  u64* words;
  int *err, *bad;
  const int iters = 400;
  const size_t wbytes = (size_t)iters * 1024 * 8;
  hipMalloc(&rec, 2 * 256 * REC * 4);
  hipMalloc(&words, wbytes);
  hipMalloc(&err, 4);
  hipMalloc(&bad, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  u64* mailbox;
  unsigned* elect;
  hipMalloc(&mailbox, 8 * 64 * 8);
  hipMalloc(&elect, 64);
  for (int mode = 3; mode < 11; ++mode)
    for (int nb : {38, 75, 150, 200, 256}) {
      float best = 1e9f;
      int e = 0, bd = 0;
      for (int rep = 0; rep < 4; ++rep) {
        const int check = rep == 0;
        hipMemset(words, 0, wbytes);
        hipMemset(err, 0, 4);
        hipMemset(mailbox, 0, 8 * 64 * 8);
        hipMemset(elect, 0, 64);
        hipMemset(bad, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        if (mode < 2)
          hipLaunchKernelGGL(k_records, dim3(nb), dim3(TPB), 0, 0, rec, words, iters, err, bad, mode, check);
        else if (mode < 4)
          hipLaunchKernelGGL(k_atomic, dim3(nb), dim3(TPB), 0, 0, words, iters, err, bad, mode - 2, check);
        else if (mode == 4)
          hipLaunchKernelGGL((k_atomic_t<29, 8>), dim3(nb), dim3(TPB), 0, 0, words, iters, err, bad, check);
        else if (mode == 5)
          hipLaunchKernelGGL((k_atomic_t<58, 8>), dim3(nb), dim3(TPB), 0, 0, words, iters, err, bad, check);
        else if (mode == 6)
          hipLaunchKernelGGL((k_atomic_t<58, 16>), dim3(nb), dim3(TPB), 0, 0, words, iters, err, bad, check);
        else if (mode == 7)
          hipLaunchKernelGGL((k_atomic_t<2, 8>), dim3(nb), dim3(TPB), 0, 0, words, iters, err, bad, check);
        else if (mode == 8)
          hipLaunchKernelGGL((k_atomic_t<11, 8>), dim3(nb), dim3(TPB), 0, 0, words, iters, err, bad, check);
        else if (mode == 9)
          hipLaunchKernelGGL((k_atomic_xcd<58>), dim3(nb), dim3(TPB), 0, 0, words, mailbox, elect, iters, err, bad, check);
        else
          hipLaunchKernelGGL((k_atomic_xcd<2>), dim3(nb), dim3(TPB), 0, 0, words, mailbox, elect, iters, err, bad, check);
        hipEventRecord(b, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (!check && ms < best) best = ms;
        int e1, b1;
        hipMemcpy(&e1, err, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&b1, bad, 4, hipMemcpyDeviceToHost);
        e |= e1;
        bd += b1;
      }
      printf("mode %d blocks %3d: %.2f us / iteration  timeout=%d wrong-blocks=%d\n", mode, nb, best * 1e3 / iters, e, bd);
      fflush(stdout);
    }
  return 0;
}
