/* lifcal_mla.h — C ABI of the step that turns virtual-image points into micro-image observations:
 * the micro-lens grid with its per-pixel lens maps, the web of epipolar base lines, and the projection itself.  Same shared
 * library as include/lifcal_ba.h (liblifcal_ba.so), same error codes and lifcal_ba_last_error().
 *
 * Replaces, in the reference:
 *   MicroLensGrid::readInGrid (derived values)   src/MicroLensGrid/MicroLensGrid.cpp:56-170   (the XML stays with the caller)
 *   MicroLensGrid::createGrid                    src/MicroLensGrid/MicroLensGrid.cpp:186-270
 *   MicroLensGrid::defineMlMaps                  src/MicroLensGrid/MicroLensGrid.cpp:338-421
 *   CameraCalibration::defineEpiPolarLines       src/CameraCalibration.cpp:521-632
 *   CameraCalibration::projectPointsToRawImage   src/CameraCalibration.cpp:637-769
 * Results are bit-identical to the reference's arithmetic (float / double / int exactly where the reference uses them, no
 * fused multiply-add), in the reference's order: frame by frame, point by point, nearest lens first, then the web.
 * The maps and the projection run on the GPU; there is no CPU fallback (LIFCAL_BA_ERR_NO_DEVICE without a gfx950 device).
 */
#ifndef LIFCAL_MLA_H
#define LIFCAL_MLA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the values MicroLensGrid::readInGrid takes from the MLA calibration file and its arguments (MicroLensGrid.cpp:60-166) */
typedef struct lifcal_mla_params {
  int32_t width, height;       /* raw image size (rawImWidth, rawImHeight) */
  float lens_diameter;         /* <diam> */
  float lens_base_y[2];        /* <lens_base_y><x>, <y> */
  float rotation;              /* <rotation> */
  float offset[2];             /* <offset><x>, <y> */
  int32_t rotation_on_grid;    /* doRotationOnGrid */
} lifcal_mla_params;

typedef struct lifcal_mla_handle lifcal_mla_handle;

/* readInGrid + createGrid + defineMlMaps + defineEpiPolarLines.  The lens list and the web are built on the host (a few
 * thousand entries, order-dependent), both maps on the device, where they stay. */
int lifcal_mla_create(const lifcal_mla_params* p, int32_t device, lifcal_mla_handle** out);
void lifcal_mla_destroy(lifcal_mla_handle* h);

/* sizes of the arrays the getters below fill */
int lifcal_mla_info(const lifcal_mla_handle* h, int32_t* n_lenses, int32_t* n_web_groups, int32_t* n_web_lines);
/* MicroLensGrid::mlLensList: centre and lens type (x mod 3) per lens, in list order */
int lifcal_mla_get_lenses(const lifcal_mla_handle* h, float* center_x, float* center_y, int32_t* lens_type);
/* mapMlPointer / mapNextMl as lens indices ([height][width], -1 = no lens), copied from the device */
int lifcal_mla_get_maps(lifcal_mla_handle* h, int32_t* map_ml, int32_t* map_next);
/* epiLineWeb flattened in its order: base-line length, unit direction, index of the equal-length group */
int lifcal_mla_get_web(const lifcal_mla_handle* h, double* base_line_dist, double* epi_x, double* epi_y, int32_t* group);

/* Image points of the virtual image, all frames concatenated in frame order (frames[i].imageCoordinates, virtualDepthValues[i]).
 * Host arrays.  fr / pt (frame index, object-point index per image point) are optional: when given they are gathered into
 * the observation list, which is then exactly what lifcal_ba_problem takes. */
typedef struct lifcal_mla_points {
  uint64_t n;
  const double* x;
  const double* y;
  const double* vdepth;
  const uint32_t* fr;   /* or NULL */
  const uint32_t* pt;   /* or NULL */
} lifcal_mla_points;

/* Observation list (rawImageCoordinates, microLensCenter, objectCoordinatesByRawID of all frames, concatenated).  Host arrays
 * of `capacity` entries supplied by the caller; src = index of the image point an observation came from. */
typedef struct lifcal_mla_observations {
  uint64_t capacity;
  uint64_t n_obs;       /* out: observations found (also when capacity was too small) */
  double* u;
  double* v;
  double* mcx;
  double* mcy;
  uint32_t* src;        /* or NULL */
  uint32_t* fr;         /* or NULL; needs points.fr */
  uint32_t* pt;         /* or NULL; needs points.pt */
} lifcal_mla_observations;

#define LIFCAL_MLA_MORE 1   /* capacity < n_obs: nothing was written, call again with n_obs entries */

/* projectPointsToRawImage for all frames in one launch: a count pass, a prefix sum and a fill pass, one lane per image
 * point.  Returns 0, LIFCAL_MLA_MORE, or a negative lifcal_ba error code.  capacity = 0 (arrays may be NULL) just counts. */
int lifcal_mla_project(lifcal_mla_handle* h, int32_t depth_to_raw_im_scale, const lifcal_mla_points* points,
                       lifcal_mla_observations* obs);

#ifdef __cplusplus
}
#endif
#endif
