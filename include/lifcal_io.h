/* lifcal_io.h — C ABI of LiFCal's result files, so that consumers of the reference's output read the
 * new solver's results unchanged.  Host code only (no GPU needed), same shared library as include/lifcal_ba.h.
 *
 * Replaces, in the reference (src/CameraCalibration.cpp):
 *   storeCameraModel               :1296-1383   CameraModel.xml            (pugixml default formatting, boost::lexical_cast numbers)
 *   storeExtrinsicOrientations     :1385-1438   extrinsicOrientations.xml
 *   storeExtrinsicOrientationsTxt  :1440-1481   ExtrinsicOrientations.txt  ("%05d" + 16 x " %16.10f", frames sorted by id)
 *   storeRawImagePointsCsv         :1483-1543   rawImagePoints.csv         ("%d,%d,%f,%f,%f,%f,%d")
 *   storeProtocol                  :1545-1617   calibrationProtocol.txt
 * All functions return 0, LIFCAL_BA_ERR_INVALID_ARG (-1: null argument / file cannot be opened) or LIFCAL_BA_ERR_OUT_OF_RANGE (-4).
 */
#ifndef LIFCAL_IO_H
#define LIFCAL_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* What the reference keeps as members after performBundleAdjustment copied camera[] back (:965-988). */
typedef struct lifcal_camera_model {
  int32_t image_width, image_height;   /* imageSize */
  double pixel_size;                   /* pixelSize [mm] */
  double fL, bL0, B, cx, cy;
  int32_t n_radial;                    /* radialDist.rows() */
  double radial[8];
  int32_t tangential;                  /* tangentialDistParam */
  double tangential_dist[2];
  int32_t ml_center_adjustment;
} lifcal_camera_model;

int lifcal_write_camera_model(const char* path, const lifcal_camera_model* m);
/* views: [6 n_frames] (three angles, three translations per frame); frame_ids: frame.id per frame */
int lifcal_write_extrinsic_orientations_xml(const char* path, uint32_t n_frames, const int32_t* frame_ids, const double* views);
int lifcal_write_extrinsic_orientations_txt(const char* path, uint32_t n_frames, const int32_t* frame_ids, const double* views);
/* One line per observation, frames in order, observations in their order inside the frame (fr must be non-decreasing, as
 * projectPointsToRawImage produces it, and below n_frames): frame id, index inside the frame, u, v, x_proj, y_proj (lifcal_ba_project_observations),
 * object-point index. */
int lifcal_write_raw_image_points_csv(const char* path, uint64_t n_obs, uint32_t n_frames, const int32_t* frame_ids, const uint32_t* fr, const double* u,
                                      const double* v, const double* x_proj, const double* y_proj, const uint32_t* pt);

typedef struct lifcal_protocol {
  lifcal_camera_model model;
  int32_t refine_poses, refine_points, robust_cost;
  double std_x, std_y, mae_x, mae_y;   /* lifcal_ba_stats */
} lifcal_protocol;
int lifcal_write_protocol(const char* path, const lifcal_protocol* p);

#ifdef __cplusplus
}
#endif
#endif
