/* lifcal_colmap.h — C ABI of the COLMAP model ingestion: a sparse COLMAP model on disk -> the flattened
 * arrays the rest of this library takes (lifcal_mla_points, lifcal_init_problem, lifcal_ba_problem).  Host code only (no GPU
 * needed), same shared library as include/lifcal_ba.h.  COLMAP itself (reference: commit 1f69517d, not in this image) is NOT
 * used: the three model files are read directly in their published layouts.
 *
 * Replaces, in the reference:
 *   CalibrationData::readDataFromFirstCalibration   src/CalibrationData/CalibrationData.cpp:56-127
 *       colmap::Reconstruction::Read(folder): cameras.bin + images.bin + points3D.bin if all three exist, else the .txt triple
 *   IntrinsicOrientation::LoadIntrinsicOrientation  src/CalibrationData/IntrinsicOrientation/IntrinsicOrientation.cpp:51-71
 *       camera id 1, parameter count must match its model; params 0..7 read as fx fy cx cy k1 k2 p1 p2 (OPENCV model);
 *       getIntrinsicParam :40-47: f = (fx + fy) / 2
 *   Images::LoadImageCoordinates                    src/CalibrationData/ImagePoints/Images.cpp:29-101
 *       per image the 2D points with a valid point3D id, the FIRST occurrence of every 3D point id, in file order
 *   ExtrinsicOrientations / ExtrinsicOrientation    .../ExtrinsicOrientation/ExtrinsicOrientation.cpp:16-29
 *       cam_from_world = (unit quaternion, translation), worldToCameraMatrix = [R t; 0 0 0 1]
 *   ObjectPoints::LoadObjectPoints                  .../ObjectPoints/ObjectPoints.cpp:23-50
 *   CalibrationData::getCalibDataCV                 src/CalibrationData/CalibrationData.cpp:492-538
 *       dense point ids (pointIdMap), per frame: image coordinates, object-point index, worldToCam, transVector and
 *       rotationAngles = rotQuat.toRotationMatrix().eulerAngles(0, 1, 2) (:531, Eigen: first angle in [0, pi])
 *
 * Ordering: the reference walks std::unordered_map<image_t, Image> / <point3D_t, Point3D>, whose order is an implementation
 * detail of libstdc++; here frames are in ascending image id and points in ascending point3D id (any order gives the same
 * calibration problem).  A 2D point that references a 3D point id missing from points3D is an error here
 * (LIFCAL_BA_ERR_OUT_OF_RANGE); the reference silently maps it to point 0 through std::map::operator[].
 * Not part of this row: ReduceNumberPoints, ArUco markers, scaling by the first constraint (:199-487).
 */
#ifndef LIFCAL_COLMAP_H
#define LIFCAL_COLMAP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct lifcal_colmap_model lifcal_colmap_model;

typedef struct lifcal_colmap_info {
  uint32_t n_frames;            /* registered images                                                               */
  uint32_t n_points;            /* 3D points                                                                       */
  uint64_t n_image_points;      /* inlier image points over all frames (sum of frames[i].imageCoordinates.size()) */
  int32_t  binary;              /* 1: read from the .bin triple, 0: from the .txt triple                          */
  int32_t  camera_model_id;     /* COLMAP model id of camera 1 (4 = OPENCV, what LiFCal's reconstruction uses)    */
  int32_t  width, height;       /* IntrinsicOrientation::width / height                                            */
  int32_t  reserved;
  double   fx, fy, cx, cy, k1, k2, p1, p2;   /* params 0..7 of camera 1 (IntrinsicOrientation.cpp:58-67)          */
  double   f;                   /* (fx + fy) / 2 (getIntrinsicParam)                                               */
} lifcal_colmap_info;

/* Reads <folder>/cameras|images|points3D (.bin preferred, else .txt).  Returns 0, LIFCAL_BA_ERR_INVALID_ARG (-1: null argument,
 * files missing, malformed file, camera 1 absent or its parameter count wrong) or LIFCAL_BA_ERR_OUT_OF_RANGE (-4: a 2D point
 * references an unknown 3D point).  lifcal_ba_last_error() has the detail. */
int lifcal_colmap_read(const char* folder, lifcal_colmap_model** out);
int lifcal_colmap_get_info(const lifcal_colmap_model* m, lifcal_colmap_info* info);

/* Frames (ascending image id).  Any pointer may be NULL.
 *   frame_ids[F]        frame::id (COLMAP image id)
 *   views[6F]           rotationAngles (Euler XYZ, Eigen eulerAngles(0,1,2)) then transVector: the layout of lifcal_ba_problem.views
 *   world_to_cam[16F]   frame::worldToCam, COLUMN-major as Eigen::Matrix4d stores it: the layout of lifcal_init_problem.world_to_cam
 *   quat_wxyz[4F]       the normalised quaternion as read (w, x, y, z)                                                       */
int lifcal_colmap_get_frames(const lifcal_colmap_model* m, int32_t* frame_ids, double* views, double* world_to_cam, double* quat_wxyz);
/* 3D points (ascending COLMAP id = dense index i): colmap_ids[P] is pointIdMapFromNewToColmap, pts[3P] the layout of lifcal_ba_problem.pts */
int lifcal_colmap_get_points(const lifcal_colmap_model* m, uint64_t* colmap_ids, double* pts);
/* Inlier image points, frame-major, file order inside a frame (frames[i].imageCoordinates): x, y [pixels of the total-focus image],
 * fr = frame index, pt = dense object-point index (objectCoordinatesByID) — the arrays lifcal_mla_points and lifcal_init_problem take
 * (the virtual depth of each point comes from the depth maps, reference readDepthData :350-451, which stay with the caller). */
int lifcal_colmap_get_image_points(const lifcal_colmap_model* m, double* x, double* y, uint32_t* fr, uint32_t* pt);
void lifcal_colmap_free(lifcal_colmap_model* m);

#ifdef __cplusplus
}
#endif
#endif
