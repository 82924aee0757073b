/*
 * orc_scalar.h — CPU ORACLE (test infrastructure; see the header of orc_track.c): the scalar section of the tracker in
 * the product's canonical operation order (orc_scalar.c).
 */
#ifndef ORC_SCALAR_H_
#define ORC_SCALAR_H_
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_kpre {
  double fx, fy, cx, cy, ifx, ify;
} orc_kpre;

orc_kpre orc_kpre_of(float fx, float fy, float cx, float cy, int level);
void orc_scalar_rodrigues(const double* src, double* R);
void orc_scalar_ldlt_pivoted_d(int n, const double* A, const double* b, double* x, double tiny);
void orc_scalar_ldlt_pivoted_f(int n, const float* A, const float* b, float* x, float tiny);
int orc_scalar_ldlt_spd6(const double* A, const double* b, double* x);
void orc_scalar_so3_params(const double* resultR, const orc_kpre* k, float* imageBasis, float* kinv, float* krlr);
void orc_scalar_gn_params(const double* resultRt, const orc_kpre* k, float* krkinv, float* kt);
/* one SO3 update: R_lr (float 3x3) and resultR (double 3x3) from the sums' jtj / jtr */
void orc_scalar_so3_update(const float* jtj, const float* jtr, float* R_lr, double* resultR);
/* one Gauss-Newton update: A, b (out, combined fp64 system), resultRt (in/out), Rcurr / tcurr (out) */
void orc_scalar_gn_update(const float* A_icp, const float* b_icp, const float* A_rgb, const float* b_rgb, int icp, int rgb, float icpWeight,
                          const float* Rprev, const float* tprev, double* resultRt, double* A, double* b, float* Rcurr, float* tcurr);

#ifdef __cplusplus
}
#endif
#endif
