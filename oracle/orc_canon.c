/*
 * orc_canon.c — CPU ORACLE (test infrastructure; see the header of orc_track.c).
 *
 * Order-free ("canonical") cross-pixel sums of the tracker's reductions.
 *
 * The reference adds the per-pixel products row_i * row_j in fp32, in an order that depends on its launch shape
 * (reduce.cu:86-147, :325-341, :412-427); any restatement that adds them in floating point inherits an order, and
 * two different orders differ in the last bits of A and b — which the Gauss-Newton loop, re-selecting 300 000
 * discrete correspondences from the float pose every iteration, amplifies to ~0.01 degrees on unlucky frames.  The
 * sum is therefore DEFINED here so that it has no order at all:
 *
 *     S_ij = sum over pixels of floor(row_i * row_j / g_ij),   value_ij = (float)(S_ij * g_ij),
 *
 * an exact integer sum of the exact products (two floats: 48 significant bits) rounded DOWN to a power-of-two grid
 * g_ij = 2^(e_ij - 51) that every participant derives from the same numbers.  Integer addition is associative, so any
 * partition of the pixels over threads / lanes / waves / blocks / atomics gives the same bits; the product computes
 * exactly this in HIP (fp64 fused multiply-adds rounding toward -inf onto the grid, exact adds above), so its sums are
 * bit-identical to the ones below.
 *
 * Grid.  Column c (Jacobian columns 0..n-1, residual column n) carries an exponent E_c that is asserted to bound its
 * diagonal total: T_cc = sum row_c^2 < 2^E_c.  Cauchy-Schwarz bounds every partial sum of every product by
 * sqrt(T_ii T_jj), so with e_ij = ceil((E_i + E_j) / 2) + 1 nothing that is added anywhere exceeds 2^e_ij: 51 bits
 * below that is the grid.  The assertion is CHECKED on the result: if a diagonal total reaches 2^E_c (S_cc >= 2^50)
 * the whole reduction is repeated with every exponent raised by 8 (at most 16 times) — the same rule on both sides,
 * so the outcome stays canonical whatever the initial guess was.  The guess itself: the previous iteration's diagonal
 * totals plus kMargin bits (orc_canon_next_exponents), four times as much on the step to a finer pyramid level, and
 * a static table for the first iteration of a call (orc_canon_static_*).
 *
 * The rounding toward -inf biases a sum by at most (number of pixels) * g, i.e. < 2^-32 of the bound: below the noise
 * of the reference's own fp32 summation by more than two orders of magnitude.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#include "orc.h"

#define CANON_MARGIN 6      /* bits of headroom over the previous diagonal totals (product: kArMargin) */
#define CANON_EMIN (-40)    /* smallest column exponent */
#define CANON_EMAX 100
#define CANON_RETRY_STEP 8
#define CANON_MAX_RETRIES 16

/* smallest e with d < 2^e for a positive finite float; -127 for anything else (bounds nothing) */
int orc_canon_exp_of(float d) {
  if (!(d > 0.f) || !(d < 3.0e38f)) return -127;
  int e;
  (void)frexpf(d, &e);
  return e;
}

static int clamp_e(int e) { return e < CANON_EMIN ? CANON_EMIN : (e > CANON_EMAX ? CANON_EMAX : e); }
int orc_canon_clamp_e(int e) { return clamp_e(e); }

/* index of the diagonal value of column c in the (n+1)-column upper-triangle layout (the residual column's square is
 * stored after the n rows: index n(n+3)/2) */
static int diag_index(int n, int c) { return c == n ? n * (n + 3) / 2 : (n + 1) * c - (c * (c - 1)) / 2; }

/* exponents for the next reduction from the totals of this one (E in: the exponents this reduction ended with; a column
 * without any contribution keeps its exponent: an empty iteration says nothing about the next one) */
void orc_canon_next_exponents(int n, const float* sums, int* E) {
  for (int c = 0; c <= n; ++c) {
    const int e = orc_canon_exp_of(sums[diag_index(n, c)]);
    if (e != -127) E[c] = clamp_e(e + CANON_MARGIN);
  }
}
/* step to the next finer pyramid level: four times the pixels */
void orc_canon_level_step(int n, int* E) {
  for (int c = 0; c <= n; ++c) E[c] = clamp_e(E[c] + 2);
}

/* first reduction of a call (nothing measured yet): per-pixel magnitude guesses times the pixel count */
void orc_canon_static_icp(int npix, int* E) { /* |n| <= 1, |v x n| <= a few metres, |residual| <= distThres */
  const int en = orc_canon_exp_of((float)npix);
  static const int m2[7] = {0, 0, 0, 4, 4, 4, -6};
  for (int c = 0; c < 7; ++c) E[c] = clamp_e(m2[c] + en);
}
void orc_canon_static_rgb(int npix, float fx_level, int rgbOnly, int* E) { /* J ~ w * gradient / 8 * f / z, w ~ 1 / sqrt(count) */
  const int ef = orc_canon_exp_of(fx_level), en = rgbOnly ? orc_canon_exp_of((float)npix) : 0;
  for (int c = 0; c < 6; ++c) E[c] = clamp_e(2 * ef + 8 + en);
  E[6] = clamp_e(10 + en);
}
void orc_canon_static_so3(int npix, int* E) { /* J ~ f * gradient, residual = intensity difference */
  const int en = orc_canon_exp_of((float)npix);
  for (int c = 0; c < 3; ++c) E[c] = clamp_e(22 + en);
  E[3] = clamp_e(10 + en);
}

/*
 * rows: [npix][n + 1] floats (all zero where found == 0), found: [npix].  E: n + 1 column exponents, raised in place
 * by retries.  sums: n(n+3)/2 products, then residual^2, then the count (the JtJJtrSE3 / JtJJtrSO3 layouts of
 * types.cuh:123-197).  Returns the number of retries (CANON_MAX_RETRIES + 1 = gave up: sums are zero, count kept).
 */
int orc_canon_reduce(int n, const float* rows, const unsigned char* found, long npix, int* E, float* sums) {
  const int np = n * (n + 3) / 2, w = n + 1;
  int vi[32], vj[32];
  {
    int k = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i; j <= n; ++j) {
        vi[k] = i;
        vj[k] = j;
        ++k;
      }
    vi[k] = vj[k] = n; /* residual^2 */
  }
  long long count = 0;
  for (long p = 0; p < npix; ++p) count += found[p] ? 1 : 0;
  int retries = 0;
  for (;; ++retries) {
    if (retries > CANON_MAX_RETRIES) {
      for (int k = 0; k <= np; ++k) sums[k] = 0.f;
      sums[np + 1] = (float)count;
      return retries;
    }
    int shift[32]; /* grid exponent s_k: g = 2^s_k */
    double inv_g[32];
    for (int k = 0; k <= np; ++k) {
      const int e = ((E[vi[k]] + E[vj[k]] + 1) >> 1) + 1;
      shift[k] = e - 51;
      inv_g[k] = ldexp(1.0, -shift[k]);
    }
    __int128 S[32];
    memset(S, 0, sizeof(S));
    int viol = 0;
#pragma omp parallel
    {
      __int128 loc[32];
      memset(loc, 0, sizeof(loc));
      int lv = 0;
#pragma omp for schedule(static) nowait
      for (long p = 0; p < npix; ++p) {
        if (!found[p]) continue;
        const float* r = rows + (size_t)p * w;
        for (int k = 0; k <= np; ++k) {
          const double scaled = ((double)r[vi[k]] * (double)r[vj[k]]) * inv_g[k]; /* exact: 48-bit product, power-of-two scale */
          if (!(fabs(scaled) < 0x1p62)) {
            lv = 1; /* (its column's diagonal is out of range too: Cauchy-Schwarz) */
            continue;
          }
          loc[k] += (long long)floor(scaled);
        }
      }
#pragma omp critical
      {
        for (int k = 0; k <= np; ++k) S[k] += loc[k];
        viol |= lv;
      }
    }
    for (int c = 0; c <= n; ++c)
      if (S[diag_index(n, c)] >= ((__int128)1 << 50)) viol = 1;
    if (viol) {
      for (int c = 0; c <= n; ++c) E[c] = clamp_e(E[c] + CANON_RETRY_STEP);
      continue;
    }
    for (int k = 0; k <= np; ++k) sums[k] = (float)ldexp((double)(long long)S[k], shift[k]); /* |S| < 2^52: exact before the float rounding */
    sums[np + 1] = (float)count;
    return retries;
  }
}
