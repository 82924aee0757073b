/* ORACLE (test infrastructure only): the host half of the reference's NID scores, computeNIDImg / computeNIDDepth
 * (Cuda/cudafuncs.cu:1556-1612 and :1838-1894): histogram / num_points in float, float marginals summed in loop order,
 * the three entropy sums accumulated in a FLOAT variable from DOUBLE terms (`joint_entropy += h * log2(h)`: log2 of a float
 * argument resolves to the double function, the product is a double, the compound assignment rounds the double sum to
 * float once per term), mutual information and the score in float.
 * PINNED: bit for bit what the reference's own code returned on an MI355X host for 20 histograms (tests/golden/ref_cudafuncs.npz
 * `nid`, tests/test_ref_cf_pin_cpu.py).  log2 is the C library's (glibc: the same library the reference's host code links). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* out[0] = nid, out[1] = joint entropy, out[2] = key-frame (column) marginal entropy, out[3] = live (row) marginal entropy */
void orc_nid_score(const uint32_t* hist, int num_bins, int num_points, float* out) {
  const size_t nb = (size_t)num_bins;
  if (num_points == 0) { /* :1548-1551 */
    out[0] = 1.0f;
    out[1] = out[2] = out[3] = 0.0f;
    return;
  }
  float* h = (float*)malloc(nb * nb * sizeof(float));
  float* PA = (float*)malloc(nb * sizeof(float));
  float* PB = (float*)malloc(nb * sizeof(float));
  for (size_t k = 0; k < nb * nb; ++k) h[k] = (float)hist[k] / (float)num_points; /* counts below 2^24: exact as floats */
  for (size_t b = 0; b < nb; ++b) { /* column-wise marginals, :1571-1578 */
    float pb = 0.0f;
    for (size_t a = 0; a < nb; ++a) pb += h[nb * a + b];
    PB[b] = pb;
  }
  for (size_t a = 0; a < nb; ++a) { /* row-wise marginals, :1580-1587 */
    float pa = 0.0f;
    for (size_t b = 0; b < nb; ++b) pa += h[nb * a + b];
    PA[a] = pa;
  }
  float joint = 0.0f, kf = 0.0f, cf = 0.0f;
  for (size_t k = 0; k < nb * nb; ++k) joint = (float)((double)joint + (double)h[k] * (h[k] == 0.0f ? 0.0 : log2((double)h[k])));
  for (size_t b = 0; b < nb; ++b) kf = (float)((double)kf + (double)PB[b] * (PB[b] == 0.0f ? 0.0 : log2((double)PB[b])));
  for (size_t a = 0; a < nb; ++a) cf = (float)((double)cf + (double)PA[a] * (PA[a] == 0.0f ? 0.0 : log2((double)PA[a])));
  cf = -cf;
  kf = -kf;
  joint = -joint;
  const float mi = kf + cf - joint; /* :1609-1610 */
  out[0] = (joint - mi) / joint;    /* :1612 */
  out[1] = joint;
  out[2] = kf;
  out[3] = cf;
  free(h);
  free(PA);
  free(PB);
}
