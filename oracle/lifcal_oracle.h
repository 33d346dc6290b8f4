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
/* frames whose pose is held constant in every following lo_sweep / lo_solve (NULL or n_frames = 0 clears): mirrors lifcal_ba_set_fixed_frames */
int lo_set_fixed_frames(const uint8_t* fixed, uint32_t n_frames);
int lo_cost(const lifcal_ba_problem* p, double loss_scale, int threads, double* cost);
int lo_residuals(const lifcal_ba_problem* p, double* r2n);
int lo_reduced_size(const lifcal_ba_problem* p, uint32_t* n_reduced, uint32_t* n_promoted);
int lo_sweep(const lifcal_ba_problem* p, const lifcal_ba_options* o, double radius, int threads,
             lifcal_ba_sweep_out* out, double* seconds_eval, double* seconds_schur);
/* the same sweep, residual blocks evaluated with the analytic Jacobian arm (oracle/analytic.hpp) instead of dual numbers */
int lo_sweep_analytic(const lifcal_ba_problem* p, const lifcal_ba_options* o, double radius, int threads,
                      lifcal_ba_sweep_out* out, double* seconds_eval, double* seconds_schur);
int lo_solve(const lifcal_ba_problem* p, const lifcal_ba_options* o, int threads, lifcal_ba_summary* sum);
int lo_reproj_stats(const lifcal_ba_problem* p, double thr, lifcal_ba_stats* out, double* errors_2n);
int lo_hardware_threads(void);
/* reference src/CameraCalibration.cpp:456-499 (initPlenopticParameters), JacobiSVD restated as a one-sided Jacobi SVD */
int lo_init_plenoptic(const lifcal_init_problem* p, lifcal_init_result* out);

/* ---- oracle/lifcal_mla.cpp: micro-lens grid, lens maps, epipolar web, projection of virtual-image points into the micro
 * images (reference src/MicroLensGrid/MicroLensGrid.cpp:186-270, :338-421; src/CameraCalibration.cpp:521-632, :637-769) */
typedef struct lo_mla_params {
  int32_t width, height;             /* raw image size */
  float lens_diameter, lens_base_y[2], rotation, offset[2];   /* <diam>, <lens_base_y>, <rotation>, <offset> of the MLA file */
  int32_t rotation_on_grid;          /* readInGrid(..., doRotationOnGrid) */
} lo_mla_params;
void* lo_mla_create(const lo_mla_params* p);
void lo_mla_destroy(void* h);
int lo_mla_counts(void* h, int32_t* n_lenses, int32_t* n_web_groups, int32_t* n_web_lines);
int lo_mla_lenses(void* h, float* cx, float* cy, int32_t* type);
int lo_mla_maps(void* h, int32_t* map_ml, int32_t* map_next);          /* lens index or -1 per raw pixel */
int lo_mla_web(void* h, double* dist, double* ex, double* ey, int32_t* group);
/* one frame of projectPointsToRawImage; returns the observation count, or -(needed) when capacity is too small */
int64_t lo_mla_project_frame(void* h, int32_t depth_to_raw_im_scale, int64_t n, const double* px, const double* py, const double* vd,
                             int64_t capacity, double* xR, double* yR, double* cX, double* cY, int64_t* point);
#ifdef __cplusplus
}
#endif
#endif
