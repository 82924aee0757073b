// Order-free ("canonical") cross-pixel sums of the tracker's reductions (reference: the fp32 tree sums of
// reduce.cu:86-147 behind icpStep / rgbStep / so3Step, whose order depends on the launch shape).
//
//     S_ij = sum over pixels of floor(row_i * row_j / g_ij),      value_ij = (float)(S_ij * g_ij)
//
// an exact integer sum of the exact products of two floats, each rounded DOWN to a power-of-two grid g_ij that every
// participant derives from the same numbers.  Integer addition is associative: lanes, waves, blocks and the memory-side
// atomics may add in any order and the totals are the same bits — in every block, from run to run, on any grid size, and
// in the CPU oracle (oracle/orc_canon.c states the definition with plain integers).
//
// How a lane computes floor(a * b / g) without an integer in sight: its accumulator starts at +-B with B = 1.5 * 2^52 * g,
// so the accumulator's unit in the last place IS g, and `v_fma_f64 acc, a, b, acc` with the fp64 rounding mode set toward
// -infinity (MODE.FP_ROUND[3:2] = 2, s_setreg) leaves acc + floor_g(a * b) — one instruction per product, as many as the
// fp32 multiply-adds it replaces.  Everything above that is exact in any rounding mode (sums of multiples of g below
// 2^53 g): the wave tree (v_permlane32_swap / v_permlane16_swap / DPP), the fp64 sum over the waves, the conversion to
// a 53-bit integer and the 64-bit atomic.  Lanes 0-31 start at +B and lanes 32-63 at -B: the tree's first exchange pairs
// lane l with lane l + 32 and the biases cancel there.
//
// Grid.  Column c of a reduction (Jacobian columns 0..N-1, residual column N) carries an exponent E_c with the claim
// T_cc = sum row_c^2 < 2^E_c.  By Cauchy-Schwarz every partial sum of every product (i, j) is below sqrt(T_ii T_jj), so
// e_ij = ceil((E_i + E_j) / 2) + 1 bounds everything that is ever added for value (i, j) by 2^e_ij, and g_ij = 2^(e_ij - 51).
// The claim is checked on the totals (S_cc >= 2^50 is a violation): the reduction is then repeated with all exponents
// raised by 8.  The exponents come from the previous iteration's diagonal totals + kMargin bits, + 2 on the step to a
// finer level, or from a static table for the first reduction of a call.  All of it must match oracle/orc_canon.c.
#pragma once
#include "common.hpp"

namespace dms {
namespace canon {

constexpr int kMargin = 6;       // bits of headroom over the previous diagonal totals
constexpr int kEmin = -40;       // smallest / largest column exponent
constexpr int kEmax = 100;
constexpr int kRetryStep = 8;    // exponents raised by this much when a diagonal total does not fit
constexpr int kMaxRetries = 16;  // per reduction; then the sums are zero (the count is kept)

// smallest e with d < 2^e for a positive finite float; -127 for anything else (bounds nothing)
__host__ __device__ __forceinline__ int exp_of(float d) {
  if (!(d > 0.f) || !(d < 3.0e38f)) return -127;
  int e;
  (void)frexpf(d, &e);
  return e;
}
__host__ __device__ __forceinline__ int clamp_e(int e) { return e < kEmin ? kEmin : (e > kEmax ? kEmax : e); }

// Layout of a reduction with N Jacobian columns (6: JtJJtrSE3, 3: JtJJtrSO3, types.cuh:123-197): value k < NP is the product
// (vi, vj) of the upper triangle of the (N+1)-column system, k = NP the residual's square, k = NP + 1 the inlier count.
template <int N>
struct Layout {
  static constexpr int NP = N * (N + 3) / 2;
  static constexpr int NV = NP + 2;
  static constexpr int SLOTS = NV > 16 ? 32 : 16;  // leaves of the wave tree
  __host__ __device__ static constexpr int vi(int k) {
    if (k >= NP) return N;
    int off = 0, r = 0;
    for (int i = 0; i < N; ++i) {
      const int len = N + 1 - i;
      if (k >= off && k < off + len) r = i;
      off += len;
    }
    return r;
  }
  __host__ __device__ static constexpr int vj(int k) {
    if (k >= NP) return N;
    int off = 0, r = 0;
    for (int i = 0; i < N; ++i) {
      const int len = N + 1 - i;
      if (k >= off && k < off + len) r = i + (k - off);
      off += len;
    }
    return r;
  }
  __host__ __device__ static constexpr int diag(int c) { return c == N ? NP : (N + 1) * c - (c * (c - 1)) / 2; }
  __host__ __device__ static constexpr bool is_diag(int k) { return k <= NP && vi(k) == vj(k); }
};

// bound exponent e_k of value k (grid exponent = e_k - 51); the count is exact on the unit grid
template <int N>
__host__ __device__ __forceinline__ int value_exp(const int* E, int k) {
  if (k > Layout<N>::NP) return 51;
  return ((E[Layout<N>::vi(k)] + E[Layout<N>::vj(k)] + 1) >> 1) + 1;
}

// E in: the exponents this reduction ended with; a column without any contribution keeps its exponent (an empty iteration
// says nothing about the next one)
template <int N>
__host__ __device__ __forceinline__ void next_exponents(const float* sums, int* E) {
#pragma unroll
  for (int c = 0; c <= N; ++c) {
    const int e = exp_of(sums[Layout<N>::diag(c)]);
    if (e != -127) E[c] = clamp_e(e + kMargin);
  }
}
template <int N>
__host__ __device__ __forceinline__ void level_step(int* E) {
#pragma unroll
  for (int c = 0; c <= N; ++c) E[c] = clamp_e(E[c] + 2);
}
template <int N>
__host__ __device__ __forceinline__ void retry_step(int* E) {
#pragma unroll
  for (int c = 0; c <= N; ++c) E[c] = clamp_e(E[c] + kRetryStep);
}
// The static guesses below assume geometry within a few metres (rotation columns |v x n|^2 <= 2^4).  With a far depth cut-off
// (--d 40 for the KITTI configuration) the first reduction of most calls outgrew them and was repeated (0.34-0.67 repeated
// reductions per frame at 1241 x 376 / 40 m); the frame step therefore raises them by this bias, a function of the depth cut-off
// alone - 0 up to 4 m, so nothing changes for the indoor configurations; +2 already removes every repeat at 40 m
// (scripts/kitti_retries.py), the rule gives +8 there (exp_of(40) = 6).  oracle/orc_pipeline.py applies the same rule.
// Clamped to the range the tracker accepts (a "no cut-off" of 1e20 or FLT_MAX is a legal option value; any guess is legal, the
// totals check corrects it).
__host__ __device__ __forceinline__ int depth_exp_bias(float depthCut) {
  const int b = depthCut > 4.0f ? 2 * (exp_of(depthCut) - 2) : 0;
  return b > 100 ? 100 : b;
}
// first reduction of a call: per-pixel magnitude guesses times the pixel count (any guess is legal: the check above corrects it)
__host__ __device__ __forceinline__ void static_icp(int npix, int* E) {
  const int en = exp_of((float)npix);
  const int m2[7] = {0, 0, 0, 4, 4, 4, -6};
#pragma unroll
  for (int c = 0; c < 7; ++c) E[c] = clamp_e(m2[c] + en);
}
__host__ __device__ __forceinline__ void static_rgb(int npix, float fx_level, int rgbOnly, int* E) {
  const int ef = exp_of(fx_level), en = rgbOnly ? exp_of((float)npix) : 0;
#pragma unroll
  for (int c = 0; c < 6; ++c) E[c] = clamp_e(2 * ef + 8 + en);
  E[6] = clamp_e(10 + en);
}
__host__ __device__ __forceinline__ void static_so3(int npix, int* E) {
  const int en = exp_of((float)npix);
#pragma unroll
  for (int c = 0; c < 3; ++c) E[c] = clamp_e(22 + en);
  E[3] = clamp_e(10 + en);
}

#if defined(__HIPCC__)
// ---- device side ------------------------------------------------------------------------------------------------------

// fp64 rounding mode of this wave: toward -infinity / back to nearest-even (MODE.FP_ROUND[3:2]; fp32 is not affected)
__device__ __forceinline__ void round_down_begin() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2"); }
__device__ __forceinline__ void round_down_end() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"); }
// acc + floor_g(a * b) between round_down_begin() and round_down_end() (volatile: stays between the two mode switches)
__device__ __forceinline__ double fma_down(double a, double b, double acc) {
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}

// Bias table in LDS: s_bias[h][k] = (h ? -1 : +1) * 1.5 * 2^(e_k + 1) for value k (2^52 grid units above the grid).
// Called by threads tid0 .. tid0 + 31 (k = tid - tid0), followed by a block barrier before the table is read.
template <int N>
__device__ __forceinline__ void write_bias(const int* E, double (*s_bias)[32], int k) {
  const double B = k < Layout<N>::NV ? ldexp(1.5, value_exp<N>(E, k) + 1) : 0.0;
  s_bias[0][k] = B;
  s_bias[1][k] = -B;
}

__device__ __forceinline__ void split(double v, unsigned& lo, unsigned& hi) {
  const long long b = __double_as_longlong(v);
  lo = (unsigned)b;
  hi = (unsigned)(b >> 32);
}
__device__ __forceinline__ double join(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }

// x <- [x(l) + x(l + 32) in lanes 0-31 | y(l - 32) + y(l) in lanes 32-63]   (exact on the grid)
__device__ __forceinline__ double swap32_add(double x, double y) {
  unsigned xl, xh, yl, yh;
  split(x, xl, xh);
  split(y, yl, yh);
  const auto r0 = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
  return join(r0[0], r1[0]) + join(r0[1], r1[1]);
}
// the same across lane ^ 16: rows 0, 2 (lane bit 4 clear) keep x, rows 1, 3 keep y
__device__ __forceinline__ double swap16_add(double x, double y) {
  unsigned xl, xh, yl, yh;
  split(x, xl, xh);
  split(y, yl, yh);
  const auto r0 = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
  return join(r0[0], r1[0]) + join(r0[1], r1[1]);
}
template <int XOR>
__device__ __forceinline__ double mov_xor(double v) {
  unsigned lo, hi;
  split(v, lo, hi);
  if constexpr (XOR == 8) {  // row_ror:8 == lane ^ 8 inside a row of 16
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, true);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, true);
  } else if constexpr (XOR == 4) {
    lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)lo, 0x101F);
    hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)hi, 0x101F);
  } else if constexpr (XOR == 2) {
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xf, 0xf, true);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xf, 0xf, true);
  } else {
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xf, 0xf, true);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, true);
  }
  return join(lo, hi);
}
// lanes with the bit clear keep x and receive their partner's x; lanes with it set keep y and receive y
template <int XOR>
__device__ __forceinline__ double sel_add(double x, double y, bool hi) {
  return (hi ? y : x) + mov_xor<XOR>(hi ? x : y);
}

// slot (value index) whose wave total lane `lane` holds after wave_sum (both lanes of a pair l, l ^ 1 hold it)
template <int SLOTS>
__device__ __forceinline__ int slot_of_lane(int lane) {
  int s = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3);
  if (SLOTS == 32) s |= ((lane >> 1) & 1) << 4;
  return s;
}

// steps 2..6 of the wave tree (lane ^ 16, ^ 8, ^ 4, ^ 2, ^ 1) on the SL / 2 registers the first step left
template <int SL>
__device__ __forceinline__ double tree_rest(const double (&o1)[SL / 2]) {
  const int lane = threadIdx.x & 63;
  double o2[SL / 4];
#pragma unroll
  for (int m = 0; m < SL / 4; ++m) o2[m] = swap16_add(o1[2 * m], o1[2 * m + 1]);
  double o3[SL / 8];
#pragma unroll
  for (int m = 0; m < SL / 8; ++m) o3[m] = sel_add<8>(o2[2 * m], o2[2 * m + 1], (lane & 8) != 0);
  double r;
  if constexpr (SL == 32) {
    double o4[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) o4[m] = sel_add<4>(o3[2 * m], o3[2 * m + 1], (lane & 4) != 0);
    r = sel_add<2>(o4[0], o4[1], (lane & 2) != 0);
  } else {
    r = sel_add<4>(o3[0], o3[1], (lane & 4) != 0);
    r += mov_xor<2>(r);
  }
  r += mov_xor<1>(r);
  return r;
}

// Canonical sum over the wave of the NV values of P pixels per lane: rows[p] = (N Jacobian entries, residual), all zero for
// a pixel without correspondence; found[p] counts.  Returns, in lane l, the wave total of value slot_of_lane(l) as a
// multiple of that value's grid (zero for slots >= NV).  s_bias: the block's bias table (write_bias).  The accumulators are
// formed two at a time and handed to the tree's first step at once (32 live registers instead of 64).
template <int N, int P>
__device__ __forceinline__ double wave_sum(const float (&rows)[P][N + 1], const bool (&found)[P], const double (*s_bias)[32]) {
  using Lay = Layout<N>;
  constexpr int SL = Lay::SLOTS;
  const int lane = threadIdx.x & 63;
  const double* bias = s_bias[lane >> 5];
  double r64[P][N + 1];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int c = 0; c <= N; ++c) r64[p][c] = (double)rows[p][c];
  // Groups of four values: the biases of the next group are fetched from LDS (into what will be its accumulators) before the
  // current group's multiply-adds are issued, so no LDS round trip sits in front of a multiply-add chain; inside a group the
  // four chains are interleaved pixel by pixel (the volatile multiply-adds keep the order written here).
  constexpr int G = 4, NG = (Lay::NV + G - 1) / G;
  double o1[SL / 2];
#pragma unroll
  for (int m = 0; m < SL / 2; ++m) o1[m] = 0.0;
  double cur[G], nxt[G];
#pragma unroll
  for (int h = 0; h < G; ++h) cur[h] = h < Lay::NV ? bias[h] : 0.0;
  round_down_begin();
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int h = 0; h < G; ++h) nxt[h] = ((g + 1) * G + h < Lay::NV) ? bias[(g + 1) * G + h] : 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
      for (int h = 0; h < G; ++h) {
        const int k = g * G + h;
        if (k <= Lay::NP)
          cur[h] = fma_down(r64[p][Lay::vi(k)], r64[p][Lay::vj(k)], cur[h]);
        else if (k == Lay::NP + 1)
          cur[h] += found[p] ? 1.0 : 0.0;  // (integers on the unit grid: exact)
      }
    }
#pragma unroll
    for (int h = 0; h < G; h += 2)
      if (g * G + h < Lay::NV) o1[(g * G + h) / 2] = swap32_add(cur[h], cur[h + 1]);
#pragma unroll
    for (int h = 0; h < G; ++h) cur[h] = nxt[h];
  }
  round_down_end();
  return tree_rest<SL>(o1);
}

// The same in three steps for loops with a variable number of pixels per lane (the launch-per-phase kernels): all NV biased
// accumulators live across the loop.
template <int N>
__device__ __forceinline__ void acc_init(double (&acc)[Layout<N>::NV], const double (*s_bias)[32]) {
  const double* bias = s_bias[(threadIdx.x & 63) >> 5];
#pragma unroll
  for (int k = 0; k < Layout<N>::NV; ++k) acc[k] = bias[k];
}
template <int N>
__device__ __forceinline__ void acc_add(double (&acc)[Layout<N>::NV], const float (&row)[N + 1], bool found) {
  using Lay = Layout<N>;
  double r64[N + 1];
#pragma unroll
  for (int c = 0; c <= N; ++c) r64[c] = (double)row[c];
  round_down_begin();
#pragma unroll
  for (int k = 0; k <= Lay::NP; ++k) acc[k] = fma_down(r64[Lay::vi(k)], r64[Lay::vj(k)], acc[k]);
  round_down_end();
  acc[Lay::NP + 1] += found ? 1.0 : 0.0;
}
template <int N>
__device__ __forceinline__ double wave_tree(const double (&acc)[Layout<N>::NV]) {
  using Lay = Layout<N>;
  constexpr int SL = Lay::SLOTS;
  double o1[SL / 2];
#pragma unroll
  for (int m = 0; m < SL / 2; ++m)
    o1[m] = (2 * m < Lay::NV) ? swap32_add(acc[2 * m], (2 * m + 1 < Lay::NV) ? acc[2 * m + 1] : 0.0) : 0.0;
  return tree_rest<SL>(o1);
}

// the wave totals of one value per lane (wave_sum / wave_tree) -> block totals in threads 0 .. NV-1
template <int N, int NW>
__device__ __forceinline__ double block_fold(double r, double (*s_red)[32]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // s_red may still be read from its previous use
  if ((lane & 1) == 0) s_red[wid][slot_of_lane<Layout<N>::SLOTS>(lane)] = r;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x < Layout<N>::NV) {
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[w][threadIdx.x];
  }
  return t;
}

// Block-wide canonical sum of P pixels per thread: wave_sum, then the wave totals through LDS; thread k < NV ends with the
// exact block total of value k (other threads: 0).  s_red: [NW][32].  Two block barriers.
template <int N, int P, int NW>
__device__ __forceinline__ double block_sum(const float (&rows)[P][N + 1], const bool (&found)[P], const double (*s_bias)[32],
                                            double (*s_red)[32]) {
  return block_fold<N, NW>(wave_sum<N, P>(rows, found, s_bias), s_red);
}

// block total (a multiple of 2^(e - 51), |t| < 2^e unless the exponents were too small) -> grid units, clamped to 52 bits;
// anything that does not fit (NaN included) becomes the largest value, which the diagonal check then reports
__device__ __forceinline__ long long to_units(double t, int e) {
  const double x = ldexp(t, 51 - e);
  return fabs(x) < 2251799813685248.0 ? (long long)x : 2251799813685247ll;  // 2^51
}
__device__ __forceinline__ float from_units(long long S, int e) { return (float)ldexp((double)S, e - 51); }
constexpr long long kViolation = 1ll << 50;  // a diagonal total of this many grid units or more: exponents too small

// memory-side all-reduce word: arrivals [63:58] | sum of (S + 2^52) [57:0], <= 32 blocks per word
__device__ __forceinline__ unsigned long long pack_word(long long S) { return (1ull << 58) + (unsigned long long)(S + (1ll << 52)); }
#endif  // __HIPCC__

}  // namespace canon
}  // namespace dms
