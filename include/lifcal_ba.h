/*
 * lifcal_ba.h — C ABI of the MI355X-native plenoptic bundle-adjustment solver.
 *
 * Drop-in boundary: the body of LiFCal's CameraCalibration::performBundleAdjustment()
 * (reference src/CameraCalibration.cpp:774-992), i.e. everything between
 * "ceres::Problem minimizationProblem" (:858) and "ceres::Solve" (:965), plus
 * calcReprojectionError() (:1026-1103).  The reference has no FFI of its own; these entry
 * points are what a binding for that seam would call (see INTEGRATION.md for the adapter).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - All parameter arrays are owned by the caller; lifcal_ba_solve() updates them IN PLACE on
 *     return (Ceres semantics: reference :892-911 registers the caller's own storage).
 *   - Every function returns 0 on success or a negative lifcal_ba_status; nothing throws.
 *   - A handle is not re-entrant; one host thread drives it.
 *   - There is NO CPU fallback: without a usable gfx950 device create() fails with
 *     LIFCAL_BA_ERR_NO_DEVICE.
 */
#ifndef LIFCAL_BA_H
#define LIFCAL_BA_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: #define MAX_NUMBER_OF_CAMERA_PARAMETERS 17  (src/CalibrationData/CalibrationData.h:19) */
#define LIFCAL_BA_MAX_CAMERA_PARAMETERS 17

/* config bitmask — identical bit layout to reference src/CameraCalibration.cpp:778-814 and the
 * decoder in src/BundleAdjustment/BundleAdjustment.h:28-79 */
#define LIFCAL_BA_CFG_NRADIAL_MASK   0x000003u  /* number of radial coefficients (0..2)           */
#define LIFCAL_BA_CFG_TANGENTIAL     0x000004u  /* two tangential coefficients                    */
#define LIFCAL_BA_CFG_REFINE_POSES   0x000100u  /* Model.refineExtrinsicOrientations              */
#define LIFCAL_BA_CFG_ROBUST         0x000200u  /* Model.robustCostFunction -> CauchyLoss(0.5)    */
#define LIFCAL_BA_CFG_REFINE_POINTS  0x000400u  /* Model.refineCoordinatesPoints                  */
#define LIFCAL_BA_CFG_ML_CENTER_ADJ  0x000800u  /* Model.adjustMicroLensCenters                   */

typedef enum {
  LIFCAL_BA_OK               =  0,
  LIFCAL_BA_ERR_INVALID_ARG  = -1,
  LIFCAL_BA_ERR_NO_DEVICE    = -2,  /* no gfx950 device / HIP runtime failure at create        */
  LIFCAL_BA_ERR_HIP          = -3,  /* a HIP call failed later (see lifcal_ba_last_error)      */
  LIFCAL_BA_ERR_OUT_OF_RANGE = -4,  /* an index in pt/fr/c_i/c_j exceeds n_points/n_frames     */
  LIFCAL_BA_ERR_NOMEM        = -5,
  LIFCAL_BA_ERR_COMM         = -6,  /* all-reduce hook / RCCL failure                           */
  LIFCAL_BA_ERR_NUMERIC      = -7   /* non-finite cost at the initial point                     */
} lifcal_ba_status;

/* termination reasons of lifcal_ba_solve (Ceres 2.1 TrustRegionMinimizer vocabulary) */
typedef enum {
  LIFCAL_BA_TERM_NONE               = 0,
  LIFCAL_BA_TERM_FUNCTION_TOLERANCE = 1,  /* |dcost| <= f_tol * cost                           */
  LIFCAL_BA_TERM_PARAMETER_TOLERANCE= 2,  /* |step| <= p_tol (|x| + p_tol)                     */
  LIFCAL_BA_TERM_GRADIENT_TOLERANCE = 3,  /* max|g| <= g_tol                                    */
  LIFCAL_BA_TERM_MAX_ITERATIONS     = 4,
  LIFCAL_BA_TERM_MIN_RADIUS         = 5,  /* trust-region radius below 1e-32                    */
  LIFCAL_BA_TERM_INVALID_STEPS      = 6   /* 5 consecutive invalid steps (Cholesky failed ...)  */
} lifcal_ba_termination;

/*
 * The problem, flattened from the reference's host structures:
 *   cam    <- double camera[17] = {fL, bL0, B, cx, cy, k_1..k_nRad, p_1, p_2, 0...}   (:821-853)
 *   views  <- double views[6F], per frame {ax, ay, az, tx, ty, tz}, Euler XYZ          (:864-868)
 *   pts    <- std::vector<Eigen::Vector3d> p3d_w, contiguous 24-byte records           (CameraCalibration.h:102)
 *   u,v    <- frame::rawImageCoordinates[k]           (CalibrationData.h:40)
 *   mcx,mcy<- frame::microLensCenter[k]               (CalibrationData.h:41)
 *   pt     <- frame::objectCoordinatesByRawID[k] - p3d_w.data()   (CalibrationData.h:42)
 *   fr     <- index of the frame the observation belongs to
 *   spx,spy<- pixelSize_totFoc (both), scale <- (double)depth_to_raw_im_scale          (:882)
 *   c_*    <- std::vector<Constraint>{pointID_1, pointID_2, distance, sigma}; they are used only
 *             when REFINE_POINTS is set and use_constraints != 0 (reference :916: not in recalib)
 *   fixed_mask / lower / upper <- recalib: SubsetManifold(17,{0,2}) and box bounds (:927-953).
 *             lower/upper may be NULL (unbounded); otherwise 17 entries, +-INFINITY = unbounded.
 * Observations may be given in any order (the reference's is frame-major); the library re-sorts
 * internally and never returns per-observation data in its own order.
 */
typedef struct lifcal_ba_problem {
  uint32_t n_obs, n_frames, n_points, n_constraints;
  const double*   u;
  const double*   v;
  const double*   mcx;
  const double*   mcy;
  const uint32_t* pt;
  const uint32_t* fr;
  double* cam;    /* [17]  in/out */
  double* views;  /* [6F]  in/out */
  double* pts;    /* [3P]  in/out */
  double spx, spy, scale;
  uint32_t config;
  uint32_t fixed_mask;       /* bit i set -> camera[i] held constant */
  const double* lower;       /* [17] or NULL */
  const double* upper;       /* [17] or NULL */
  const uint32_t* c_i;       /* [M] or NULL */
  const uint32_t* c_j;
  const double*   c_dist;
  const double*   c_sigma;
  uint32_t use_constraints;  /* 0: ignore c_* (recalib), 1: add them when REFINE_POINTS */
  uint32_t reserved;
} lifcal_ba_problem;

/* Solver options; defaults = what the reference hard-codes (:955-961) + Ceres 2.1 defaults. */
typedef struct lifcal_ba_options {
  double function_tolerance;    /* 1e-6  (:958) */
  double parameter_tolerance;   /* 1e-8  (:959) */
  double gradient_tolerance;    /* 1e-10 (Ceres default) */
  double initial_radius;        /* 1e4   (Ceres default initial_trust_region_radius) */
  double max_radius;            /* 1e16 */
  double min_radius;            /* 1e-32 */
  double min_relative_decrease; /* 1e-3 */
  double min_lm_diagonal;       /* 1e-6 */
  double max_lm_diagonal;       /* 1e32 */
  double loss_scale;            /* 0.5: CauchyLoss(0.5) (:892) */
  int32_t max_iterations;       /* 200 (:960) */
  int32_t jacobi_scaling;       /* 1 */
  int32_t precision;            /* 0: fp64 everywhere; 1: residual and Jacobian of an observation evaluated in fp32 (observation stored relative to its
                                   micro-lens centre, fp32 lens table: 12 B per observation in the sweep's stream), every accumulation, the
                                   elimination and the solve in fp64; accept/reject decisions use fp64 costs (BASELINE configs[4])              */
  int32_t device;               /* HIP device ordinal */
  int32_t rank;                 /* this process' rank in the point-sharded job (0 if single GPU) */
  int32_t world_size;           /* number of ranks (1 if single GPU) */
  int32_t verbose;              /* 1: print Ceres-style per-iteration table to stdout (:957) */
  int32_t deterministic;        /* 1: bitwise reproducible results — LDS accumulation in wave order, per-block window slabs summed in block order
                                   instead of the cross-block f64 atomics, value kernels summed per workgroup in order; points with distance
                                   constraints, tracks longer than the LDS window and the camera-only / pose-only arities emit their global
                                   atomics one wave after the other (slow when every point takes that path).  Every problem structure,
                                   box bounds included, is supported.                                                                     */
} lifcal_ba_options;

typedef struct lifcal_ba_summary {
  double initial_cost, final_cost;
  double final_radius;
  double final_gradient_max_norm;
  int32_t iterations;             /* LM iterations taken (Ceres counts iteration 0 separately) */
  int32_t successful_steps, unsuccessful_steps;
  int32_t termination;            /* lifcal_ba_termination */
  double seconds_total, seconds_sweep, seconds_linear_solve;
} lifcal_ba_summary;

/* One Jacobian+Schur sweep (the benchmarked unit).  Host pointers may be NULL to
 * skip the copy-back.  Canonical reduced ordering of S/rhs: [camera 0..16 | views 6F | promoted
 * points 3 each, in ascending point id]; fixed / structurally-dead camera slots are returned as
 * identity rows with rhs 0.  S is row-major n_red x n_red, symmetric, both triangles filled. */
typedef struct lifcal_ba_sweep_out {
  double cost;                   /* 1/2 sum rho(|r|^2) (+ constraints) at the current point      */
  double gradient_max_norm;      /* max |J^T r| over all free parameters                          */
  uint32_t n_reduced;            /* 17 + 6F + 3*n_promoted                                        */
  uint32_t n_promoted;
  double* S;                     /* [n_reduced^2] or NULL                                         */
  double* rhs;                   /* [n_reduced]   or NULL:  S * delta_reduced = rhs               */
  double* gradient_reduced;      /* [n_reduced]   or NULL:  J_B^T r before elimination            */
  double* point_gradient;        /* [3P]          or NULL:  J_P^T r                               */
  double* point_hessian_inv;     /* [9P]          or NULL:  (U_p + D_p)^-1 row-major              */
  double seconds;                /* device time of the sweep kernels (HIP events)                 */
} lifcal_ba_sweep_out;

/* reference calcReprojectionError (:1026-1103) */
typedef struct lifcal_ba_stats {
  double std_x, std_y;   /* sqrt(sum e^2 / N): RMS, not mean-removed (:1097-1098) */
  double mae_x, mae_y;   /* MAX abs error despite the name (:1083-1084)           */
  uint32_t num_points, num_inliers;   /* |e|^2 <= thr^2 (:1088)                   */
} lifcal_ba_stats;

typedef struct lifcal_ba_handle lifcal_ba_handle;

/* Multi-GPU: observations are sharded by 3D point across ranks (one process per GPU).  The only
 * data-path exchange is a sum all-reduce of f64 device buffers.  Either install a hook (the host
 * language supplies the collective, e.g. torch.distributed -> RCCL) or let the library own an
 * RCCL communicator (lifcal_ba_comm_init_rccl).  `stream` is the hipStream_t the buffer was
 * produced on; the hook must leave the reduced result in place, ordered on that stream. */
typedef int (*lifcal_ba_allreduce_fn)(void* ctx, void* device_buf, size_t count_f64, void* stream);
/* Optional second collective.  Points are owned in first-frame order, so the partial reduced system of a rank is
 * non-zero only in ONE contiguous range of frames (+ the small camera block): instead of summing the whole block the
 * library packs that range into a slab, ALL-GATHERS the equally sized slabs (count_f64 doubles per rank; recv holds
 * world_size * count_f64, rank-major) and adds them up locally - about half the bytes of the all-reduce, and it is the
 * path the library-owned RCCL communicator takes by itself.  With hooks it is used when BOTH hooks are installed (the
 * all-reduce is still needed for a few small buffers); LIFCAL_DENSE_ALLREDUCE=1 forces the plain all-reduce. */
typedef int (*lifcal_ba_allgather_fn)(void* ctx, const void* device_send, void* device_recv, size_t count_f64, void* stream);

void lifcal_ba_default_options(lifcal_ba_options* o);

/* replaces reference :858-953 (problem construction): validates, sorts observations point-major,
 * builds the tile layout and uploads everything to HBM. */
int lifcal_ba_create(const lifcal_ba_problem* p, const lifcal_ba_options* o, lifcal_ba_handle** out);

/* replaces reference :965 (ceres::Solve) + :967-988 is the caller's unpack of cam[] */
int lifcal_ba_solve(lifcal_ba_handle* h, lifcal_ba_summary* s);

/* one Jacobian+Schur sweep at trust-region radius `radius` on the CURRENT device-resident point */
int lifcal_ba_sweep(lifcal_ba_handle* h, double radius, lifcal_ba_sweep_out* out);

/* the same sweep without the host round trip: kernels are only enqueued on the handle's stream (read the
 * result later with lifcal_ba_sweep, or synchronise the stream).  Used for back-to-back timing. */
int lifcal_ba_sweep_enqueue(lifcal_ba_handle* h, double radius);

/* HIP-event instrumentation of the sweeps on the stream the kernels run on.  The dominant kernel carries its own start / stop
 * events (written by its dispatch packet: no extra barrier packets in the queue); one record opens the span before the first
 * profiled sweep, one closes it behind the last.  begin() reserves events for up to max_sweeps sweeps; end() synchronises and averages. */
typedef struct lifcal_ba_profile {
  uint32_t n_sweeps;
  double special_points;  /* points worked by the special-point kernels (k_sweep + k_schur: constraints, oversized groups,
                             camera-only / pose-only arities); their time is part of ms_schur, NOT of ms_accumulate        */
  double ms_accumulate;   /* the dominant kernel of the regular points by its own dispatch time stamps: k_sweep3 (fused residual +
                             Jacobian + accumulation + point elimination), or k_front4 start to k_back4 end                */
  double ms_schur;        /* ms_total - ms_accumulate: tables, special points, constraints, exchange, k_finalize, gaps  */
  double ms_total;        /* first kernel of the first sweep to the end of the last one, divided by the sweep count    */
  double ms_exchange;     /* world_size > 1: pack + collective + unpack of the partial reduced blocks (part of ms_schur)  */
  uint32_t n_sampled;     /* sweeps of the span whose dominant kernel carried the time stamps (ms_accumulate / ms_exchange average THESE) */
} lifcal_ba_profile;
/* The next max_sweeps sweeps form a span (ms_total: two event records, before the first and behind the last).  A SAMPLED sweep's
 * dominant kernel carries its own start / stop events, which costs ~5 us of queue time per sampled sweep: _begin samples every
 * sweep, _begin_sampled every stride-th (0, stride, 2 stride, ...) so that a timing loop is measured nearly undisturbed.            */
int lifcal_ba_profile_begin(lifcal_ba_handle* h, uint32_t max_sweeps);
int lifcal_ba_profile_begin_sampled(lifcal_ba_handle* h, uint32_t max_sweeps, uint32_t stride);
int lifcal_ba_profile_end(lifcal_ba_handle* h, lifcal_ba_profile* out);

/* replaces reference :1026-1103 (evaluated on the device-resident parameters) */
int lifcal_ba_reproj_stats(lifcal_ba_handle* h, double inlier_threshold, lifcal_ba_stats* out);
/* The x_proj / y_proj columns of reference storeRawImagePointsCsv (:1504-1538): the model's projection of every observation at
 * the stored parameters, [n_obs] host arrays in the caller's observation order, evaluated as reproj_stats evaluates it.
 * With world_size > 1 a rank fills the observations of the points it owns and leaves NaN elsewhere. */
int lifcal_ba_project_observations(lifcal_ba_handle* h, double* x_proj, double* y_proj);

/* Poses held constant: fixed[f] != 0 keeps views[6f..6f+5] at their stored values in every following sweep / solve (ceres
 * SetParameterBlockConstant on that pose block: the frame's observations still constrain camera and points, its six columns
 * leave the reduced system).  fixed == NULL frees all poses again.  No reference counterpart; it is what the frame-windowed
 * driver below needs to carry finished frames into the next window (BASELINE configs[4] "streaming"). */
int lifcal_ba_set_fixed_frames(lifcal_ba_handle* h, const uint8_t* fixed /* [n_frames] or NULL */);

/* Frame-windowed ("streaming") bundle adjustment for long sequences (BASELINE configs[4]: recalib, 2000 frames): the frame axis is
 * cut into windows of `window_frames` frames that advance by window_frames - overlap_frames; each window is ONE ordinary problem
 * (its frames, the points they observe, the camera block with the caller's fixed_mask / bounds) created, solved and destroyed
 * in turn, so only one window is resident on the device.  From the second window on, the first overlap_frames poses — already
 * refined by the previous window — are held constant, and points whose observations all lie in earlier windows keep their
 * values.  Parameters are updated in place like lifcal_ba_solve; `per_window`, if not NULL, receives one summary per window
 * (capacity *n_windows on entry, count on return).  With world_size > 1 every window is sharded by 3D point as usual; the
 * collective hooks / communicator of `comm_template` (a handle created on the SAME options, may be NULL at world_size 1) are reused. */
typedef struct lifcal_ba_window_report {
  uint32_t first_frame, n_frames, n_fixed_frames, n_points, n_obs;
  uint32_t n_dropped_constraints;   /* distance constraints touching the window whose second point lies outside it (not applied) */
  lifcal_ba_summary summary;
} lifcal_ba_window_report;
int lifcal_ba_solve_windowed(const lifcal_ba_problem* p, const lifcal_ba_options* o, uint32_t window_frames, uint32_t overlap_frames,
                             lifcal_ba_handle* comm_template, lifcal_ba_window_report* per_window, uint32_t* n_windows);

/* re-upload cam/views/pts from the caller's arrays (e.g. to re-run from a new initial point) */
int lifcal_ba_upload_parameters(lifcal_ba_handle* h);
/* copy the device-resident cam/views/pts back into the caller's arrays */
int lifcal_ba_download_parameters(lifcal_ba_handle* h);

int lifcal_ba_set_allreduce(lifcal_ba_handle* h, lifcal_ba_allreduce_fn fn, void* ctx);
int lifcal_ba_set_allgather(lifcal_ba_handle* h, lifcal_ba_allgather_fn fn, void* ctx);
/* RCCL path: rank 0 calls unique_id (128 bytes), distributes it, every rank calls init */
int lifcal_ba_comm_unique_id(void* out128);
int lifcal_ba_comm_init_rccl(lifcal_ba_handle* h, const void* unique_id128);

/* introspection used by bench.py / tests */
typedef struct lifcal_ba_info {
  uint32_t n_obs_local, n_points_local, n_groups, n_tiles, n_lenses, n_reduced, n_promoted;
  uint32_t n_chunks, max_window_frames;
  uint64_t device_bytes;
  void* stream;          /* hipStream_t all kernels are launched on */
} lifcal_ba_info;
int lifcal_ba_get_info(lifcal_ba_handle* h, lifcal_ba_info* out);

void lifcal_ba_destroy(lifcal_ba_handle* h);
const char* lifcal_ba_strerror(int code);
const char* lifcal_ba_last_error(void);
const char* lifcal_ba_version(void);

/* Host-only planning (no device needed): the observation re-ordering create() performs.
 * Exposed so the CPU test-suite can check the layout invariants without a GPU. */
typedef struct lifcal_ba_plan_info {
  uint32_t n_groups, n_tiles, n_lenses, n_promoted, n_reduced, max_group_obs, n_chunks, max_window_frames;
} lifcal_ba_plan_info;
int lifcal_ba_plan(const lifcal_ba_problem* p, int32_t rank, int32_t world_size,
                   lifcal_ba_plan_info* info,
                   uint32_t* obs_order /* [n_obs] or NULL: sorted position -> input index, UINT32_MAX-padded */,
                   uint32_t* point_owner /* [n_points] or NULL: rank owning each point */);

/* ---- shard-local problems (multi-GPU without handing every rank the whole observation list) ----
 * lifcal_ba_create takes the WHOLE problem on every rank and keeps the observations of the points the rank owns.  For long
 * sequences (BASELINE configs[3], configs[4]) a launcher can instead partition ONCE from the index arrays alone (pt, fr: 8 bytes
 * per observation) with lifcal_ba_partition, hand each rank only the observations of its points, and create the rank's handle
 * with lifcal_ba_create_shard.  The resulting layout, kernels and collectives are those of lifcal_ba_create.
 * Host-only, no device needed for the partition.  Distance constraints are not supported on shards (they couple points across
 * ranks: use lifcal_ba_create). */
typedef struct lifcal_ba_partition {
  uint32_t world_size, n_frames, n_points, band_width;   /* band_width: max over points of (last frame - first frame)          */
  uint64_t n_obs;                                        /* observations of the whole problem                                   */
  int32_t*  point_owner;    /* [n_points] caller-allocated: rank owning the point, -1 = not observed                            */
  uint32_t* rank_first;     /* [world_size] caller-allocated: first frame observed by the rank's points                         */
  uint32_t* rank_frames;    /* [world_size] caller-allocated: number of frames from rank_first to the last frame they observe   */
  uint64_t* rank_obs;       /* [world_size] caller-allocated: observations of the rank's points                                 */
  uint8_t*  frame_used;     /* [n_frames] caller-allocated: frame observed by any point                                         */
} lifcal_ba_partition;
/* fills the caller-allocated arrays of `part` (world_size must be set) from the index arrays of the whole problem: only p->n_obs,
 * n_frames, n_points, pt and fr are read.  Same ownership rule as lifcal_ba_create (points in first-frame order, contiguous
 * ranges balanced by observation count). */
int lifcal_ba_partition_points(const lifcal_ba_problem* index_only, lifcal_ba_partition* part);
/* `local` holds ONLY the observations of the points `rank` owns (any order), but the full cam / views / pts arrays (pts of other
 * ranks' points are neither read nor written until the final gather). */
int lifcal_ba_create_shard(const lifcal_ba_problem* local, const lifcal_ba_partition* part, const lifcal_ba_options* o, lifcal_ba_handle** out);
/* host-only planning of a shard (what lifcal_ba_plan is for whole problems): the CPU test-suite compares the two */
int lifcal_ba_plan_shard(const lifcal_ba_problem* local, const lifcal_ba_partition* part, int32_t rank, lifcal_ba_plan_info* info);

/* ---- start values of the plenoptic parameters ----
 * Replaces reference src/CameraCalibration.cpp:456-499 (CameraCalibration::initPlenopticParameters): with
 * bL = fL z / (z - fL), z the camera-frame depth of the object point of an image point, the linear model
 * bL = v B + bL0 (v = virtual depth) is fitted over all image points of all frames in the least-squares sense; a row is
 * dropped (zeroed, as the reference does) when v < 2 or bL < 0.  The reference solves with Eigen::JacobiSVD, i.e. the
 * minimum-norm solution with singular values below 2 eps sigma_max treated as zero; so does this.
 * Inputs are plain host arrays, one entry per image point in the reference's frame-major order (any order works).  The
 * sums run on the device (one reduction kernel), the 2x2 solve on the host. */
typedef struct lifcal_init_problem {
  uint64_t n;                 /* image points over all frames (reference: sum of frames[i].imageCoordinates.size())      */
  const double* vdepth;       /* [n] virtual depth of the point (reference virtualDepthValues[i][p])                      */
  const uint32_t* fr;         /* [n] frame of the point                                                                    */
  const uint32_t* pt;         /* [n] index of its object point (reference objectCoordinatesByID[p] - p3d_w.data())        */
  uint32_t n_frames, n_points;
  const double* world_to_cam; /* [n_frames][16] COLUMN-major, as Eigen::Matrix4d frame::worldToCam stores it              */
  const double* pts;          /* [n_points][3]                                                                             */
  double fL_init;             /* fPH_init * pixelSize_totFoc (reference :460)                                              */
} lifcal_init_problem;
typedef struct lifcal_init_result {
  double B_init, bL0_init;    /* x[0], x[1] of the reference (:492-493)                                                    */
  uint64_t n_used;            /* rows that entered the fit                                                                 */
  int32_t rank;               /* numerical rank of the n x 2 system (2 unless degenerate)                                  */
  int32_t reserved;
} lifcal_init_result;
int lifcal_init_plenoptic(const lifcal_init_problem* p, int32_t device, lifcal_init_result* out);
/* Replaces reference src/CameraCalibration.cpp:503-512 (CameraCalibration::initPlenopticParametersRecalibration): recalib mode takes fL
 * and B from the previous calibration (calibData->getFixedParameters) and starts bL0 at fL - 2 B.  Host arithmetic, no device. */
int lifcal_init_plenoptic_recalibration(double fL_fixed, double B_fixed, lifcal_init_result* out);

#ifdef __cplusplus
}
#endif
#endif /* LIFCAL_BA_H */
