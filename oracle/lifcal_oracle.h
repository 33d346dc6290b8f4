/* oracle/lifcal_oracle.h — TEST INFRASTRUCTURE ONLY: C API of the CPU restatement (see lifcal_oracle.cpp).
 * PARITY UNPINNED (no reference fixtures exist; reference not buildable in this image). */
#ifndef LIFCAL_ORACLE_H
#define LIFCAL_ORACLE_H
#include <stdint.h>
#include "../include/lifcal_ba.h"
#ifdef __cplusplus
extern "C" {
#endif
int lo_project_point(const double pc[3], double spx, double spy, double fL, double bL0, double B,
                     const double c_raw[2], const double ml[2], const double* radial, int n_radial,
                     const double* tangential, int ml_center_adj, double out[2]);
int lo_rigid_transform(const double view[6], double RT[12]);
int lo_residual_block(uint32_t config, int arity, const double cam[17], const double view[6], const double point[3],
                      double u, double v, double mcx, double mcy, double spx, double spy, double scale,
                      double r[2], double J[52]);
int lo_constraint_block(const double p1[3], const double p2[3], double distance, double sigma, double* r, double J[6]);
int lo_cost(const lifcal_ba_problem* p, double loss_scale, int threads, double* cost);
int lo_residuals(const lifcal_ba_problem* p, double* r2n);
int lo_reduced_size(const lifcal_ba_problem* p, uint32_t* n_reduced, uint32_t* n_promoted);
int lo_sweep(const lifcal_ba_problem* p, const lifcal_ba_options* o, double radius, int threads,
             lifcal_ba_sweep_out* out, double* seconds_eval, double* seconds_schur);
int lo_solve(const lifcal_ba_problem* p, const lifcal_ba_options* o, int threads, lifcal_ba_summary* sum);
int lo_reproj_stats(const lifcal_ba_problem* p, double thr, lifcal_ba_stats* out, double* errors_2n);
int lo_hardware_threads(void);
/* reference src/CameraCalibration.cpp:456-499 (initPlenopticParameters), JacobiSVD restated as a one-sided Jacobi SVD */
int lo_init_plenoptic(const lifcal_init_problem* p, lifcal_init_result* out);
#ifdef __cplusplus
}
#endif
#endif
