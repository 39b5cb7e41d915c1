/* lvf.h — C-ABI of the MI355X-native lvio_fusion hot path (local-BA factor evaluation,
 * normal-equation / Schur reduction, LiDAR scan-to-map 3-NN association + point-to-plane ICP).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each group cites
 * the reference interface it replaces (paths under src/lvio_fusion/ of jypjypjypjyp/lvio_fusion):
 *
 *   lvf_pose_only_*    <- PoseOnlyReprojectionError::Create / AutoDiffCostFunction<..,2,7>::Evaluate
 *                         include/lvio_fusion/ceres/visual_error.hpp:48-76
 *   lvf_two_frame_*    <- TwoFrameReprojectionError  <2,1,7,7>   visual_error.hpp:78-107
 *   lvf_two_camera_*   <- TwoCameraReprojectionError <2,1>       visual_error.hpp:109-137
 *   lvf_lidar_plane_*  <- LidarPlaneErrorRPZ / LidarPlaneErrorYXY <1,1,1,1>
 *                         include/lvio_fusion/ceres/lidar_error.hpp:42-110 (+ ctor normal :13-18)
 *   lvf_imu_*          <- ImuError : SizedCostFunction<15,7,3,3,3,7,3,3,3>
 *                         include/lvio_fusion/ceres/imu_error.hpp:12-122, src/preintegration.cpp:144-165
 *   lvf_pose_prior_*   <- PoseGraphError <6,7,7> / PoseError <6,7>   include/lvio_fusion/ceres/pose_error.hpp:10-86
 *   lvf_preintegrate   <- imu::Preintegration::{Append,Propagate,MidPointIntegration}
 *                         src/preintegration.cpp:30-127, include/lvio_fusion/imu/preintegration.h:29-41
 *   lvf_map_* / lvf_scan_* / lvf_knn3_*
 *                      <- pcl::KdTreeFLANN::setInputCloud + nearestKSearch(point,3,..) + the 3x(d2<thr) gate
 *                         src/association.cpp:278-301 (ground), :336-359 (surf)
 *   lvf_icp_*          <- FeatureAssociation::ScanToMapWith{Ground,Segmented} + the DENSE_QR solve
 *                         src/association.cpp:270-384, src/mapping.cpp:154-178
 *   lvf_cloud_*        <- Mapping::MergeScan/ToWorld/BuildMapFrame, pcl::VoxelGrid / RadiusOutlierRemoval / SACSegmentation
 *                         src/mapping.cpp:78-137,193-220, src/association.cpp:210-268
 *   lvf_scan_match     <- Mapping::Optimize's per-frame body / Mapping::Relocate   src/mapping.cpp:147-178, :251-300
 *   lvf_window_*       <- Backend::BuildProblem's assembly kept incrementally          src/backend.cpp:96-183
 *   lvf_problem_*      <- adapt::Problem::{AddParameterBlock,AddResidualBlock,SetParameterBlockConstant}
 *                         + adapt::Solve (include/lvio_fusion/adapt/problem.h:34-88) as driven by
 *                         Backend::BuildProblem / Backend::Optimize (src/backend.cpp:96-183, :192-246)
 *
 * Conventions
 *   - pose block = Sophus SE3d::data() = [qx,qy,qz,qw,tx,ty,tz] (7 doubles), Twc maps body->world.
 *   - Jacobians are row-major num_residuals x block_size, w.r.t. AMBIENT parameters
 *     (the ceres::CostFunction::Evaluate contract).
 *   - every function returns 0 (LVF_OK) on success; on failure outputs are untouched and
 *     lvf_last_error() (thread-local) describes the problem.  No exceptions cross the boundary.
 *   - "create" calls copy their inputs to HBM; evaluate/solve calls are asynchronous on the
 *     context's HIP stream and leave their outputs resident in HBM; *_download calls synchronise.
 *   - a context is bound to one device and one stream; use one context per host thread
 *     (Backend::Optimize and Relocator can be in flight concurrently: src/relocator.cpp:188).
 *   - there is NO CPU fallback: without a usable gfx950 device lvf_ctx_create fails.
 */
#ifndef LVF_H_
#define LVF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVF_OK 0
#define LVF_ERR_INVALID 1   /* bad argument */
#define LVF_ERR_HIP 2       /* HIP runtime error (see lvf_last_error) */
#define LVF_ERR_NO_DEVICE 3 /* no usable GPU: the product path refuses to run */
#define LVF_ERR_STATE 4     /* call sequence error (e.g. download before evaluate) */

typedef struct lvf_ctx lvf_ctx;
typedef struct lvf_state lvf_state;
typedef struct lvf_batch lvf_batch;
typedef struct lvf_map lvf_map;
typedef struct lvf_scan lvf_scan;
typedef struct lvf_icp lvf_icp;
typedef struct lvf_cloud lvf_cloud;
typedef struct lvf_window lvf_window;
typedef struct lvf_problem lvf_problem;
typedef struct lvf_comm lvf_comm;
typedef struct lvf_problem_batch lvf_problem_batch;

/* Camera = intrinsics + sensor->robot extrinsic (include/lvio_fusion/sensor.h:41-44, visual/camera.h:74). */
typedef struct lvf_camera {
  double fx, fy, cx, cy;
  double extrinsic[7];
} lvf_camera;

/* Snapshot of imu::Preintegration after the last Append (include/lvio_fusion/imu/preintegration.h:66-86). */
typedef struct lvf_preint {
  double sum_dt;
  double lin_ba[3];
  double lin_bg[3];
  double dp[3];
  double dq[4]; /* x,y,z,w */
  double dv[3];
  double jac[225]; /* 15x15 row-major */
  double cov[225]; /* 15x15 row-major */
} lvf_preint;

const char* lvf_last_error(void);
const char* lvf_version(void);

/* Host memory the device can copy from without the runtime pinning and unpinning the pages around every copy (page-locked, recycled
 * process-wide in size buckets).  For arrays that are filled on the host and then handed to lvf_*_create / lvf_state_set: the adapter
 * (include/lvf_ceres_adapter.hpp) records the window's block lists straight into it while Backend::BuildProblem adds blocks
 * (src/lvio_fusion/src/backend.cpp:96-183), so gpu::Solve's uploads are plain DMA.  Falls back to ordinary memory when no device is
 * usable (host-only tools).  lvf_host_free takes the SAME byte count the block was allocated with.  Thread-safe. */
void* lvf_host_alloc(size_t bytes);
void lvf_host_free(void* p, size_t bytes);

/* ---- context ------------------------------------------------------------------------------ */
/* stream: an existing hipStream_t to run on (e.g. the caller's), or NULL to create a private one. */
int lvf_ctx_create(int device, void* hip_stream, lvf_ctx** out);
int lvf_ctx_destroy(lvf_ctx* ctx);
int lvf_ctx_synchronize(lvf_ctx* ctx);
void* lvf_ctx_stream(lvf_ctx* ctx);
/* HIP-event timing of the context's stream, for bench.py: begin/end record events, elapsed synchronises. */
int lvf_timer_begin(lvf_ctx* ctx);
int lvf_timer_end(lvf_ctx* ctx);
int lvf_timer_elapsed_ms(lvf_ctx* ctx, float* ms);
/* Box calibration for the latency-bound legs (bench.py `box_calibration`): out8 = {ns per dependent fp64 FMA of one wave, shader clocks per
 * such FMA, effective shader clock MHz during the chain, us per empty launch back to back, us for launch + stream wait of one empty
 * kernel, rated shader clock MHz, memory clock MHz, compute units}.  Not part of the reference surface. */
int lvf_box_calibration(lvf_ctx* ctx, double* out8);

/* ---- parameter state (the caller-owned double* blocks of the Ceres problem, mirrored in HBM) - */
enum lvf_field {
  LVF_POSES = 0,     /* n_kf x 7   frame->pose.data()                      backend.cpp:110 */
  LVF_VEL = 1,       /* n_kf x 3   frame->Vw.data()                        backend.cpp:147 */
  LVF_BA = 2,        /* n_kf x 3   frame->bias.linearized_ba.data()        backend.cpp:149 */
  LVF_BG = 3,        /* n_kf x 3   frame->bias.linearized_bg.data()        backend.cpp:148 */
  LVF_INV_DEPTH = 4, /* n_lm       &landmark->inv_depth                    backend.cpp:121 */
  LVF_W_VISUAL = 5   /* n_kf       frame->weights.visual                   frame.cpp:13    */
};
int lvf_state_create(lvf_ctx* ctx, int n_kf, int n_lm, lvf_state** out);
int lvf_state_destroy(lvf_state* st);
int lvf_state_set(lvf_state* st, int field, const double* host);
int lvf_state_get(lvf_state* st, int field, double* host);
/* lvf_state_create + the fields in one staged upload and one wait; a NULL field takes its default (identity poses, zeros, unit weights). */
int lvf_state_create_from(lvf_ctx* ctx, int n_kf, int n_lm, const double* poses, const double* vel, const double* ba, const double* bg,
                          const double* inv_depth, const double* w_visual, lvf_state** out);
/* lvf_state_set for every field at once: ONE staged upload and one wait; a NULL pointer leaves that field as it is. */
int lvf_state_set_all(lvf_state* st, const double* poses, const double* vel, const double* ba, const double* bg, const double* inv_depth,
                      const double* w_visual);
/* dst <- src (all fields; same n_kf / n_lm), device to device, asynchronous on dst's context: restoring a saved estimate without a host
 * round trip. */
int lvf_state_copy(lvf_state* dst, const lvf_state* src);

/* ---- factor batches (one batch = all residual blocks of one functor type) ------------------ */
/* n blocks; ob[n][2]; kf_idx[n] selects the pose block and the per-frame weight; pw_idx[n] indexes
 * pw[n_pw][3] (landmark->ToWorld(), constant during the solve: backend.cpp:129). */
int lvf_pose_only_create(lvf_ctx* ctx, const lvf_camera* cam0, int n, const double* ob, const int32_t* kf_idx,
                         const int32_t* pw_idx, int n_pw, const double* pw, lvf_batch** out);
/* first_ob[n][2] = right-image observation in the birth frame, ob[n][2] = current left observation;
 * parameter blocks (inv_depth[lm_idx], pose[kf1_idx], pose[kf2_idx]); weight = w_visual[kf2]. backend.cpp:134-139 */
int lvf_two_frame_create(lvf_ctx* ctx, const lvf_camera* left, const lvf_camera* right, int n,
                         const double* first_ob, const double* ob, const int32_t* lm_idx, const int32_t* kf1_idx,
                         const int32_t* kf2_idx, lvf_batch** out);
/* parameter block inv_depth[lm_idx]; weight = 5 * w_visual[kf_idx]   backend.cpp:119-124 */
int lvf_two_camera_create(lvf_ctx* ctx, const lvf_camera* left, const lvf_camera* right, int n,
                          const double* left_ob, const double* right_ob, const int32_t* lm_idx,
                          const int32_t* kf_idx, lvf_batch** out);
/* parameter blocks (pose,v,ba,bg)[kf_i], (pose,v,ba,bg)[kf_j]; sqrt_info = LLT(cov^-1).L^T is factorised
 * once here (the reference re-inverts on every Evaluate: imu_error.hpp:32). */
/* The caller VOUCHES that a TwoFrame batch has the shape Backend::BuildProblem always produces (backend.cpp:105-140) — sorted by current
 * keyframe, first keyframe < current keyframe, at most one block per (landmark, current keyframe), one first keyframe per landmark — and hands
 * blocks_per_kf2[n_kf]: lvf_problem_create then skips its host pass over the blocks.  A false claim gives wrong results; callers that cannot
 * prove it do not call this (include/lvf_ceres_adapter.hpp proves it block by block while adapt::Problem is being filled). */
int lvf_two_frame_set_shape(lvf_batch* two_frame, int n_kf, const int32_t* blocks_per_kf2);
/* Optional per-block weights of a TwoCamera batch = each functor's `weight` constructor argument (visual_error.hpp:112; the reference
 * passes 5 * frame->weights.visual, backend.cpp:123).  weight[n] replaces 5 * w_visual[kf_idx[i]]; NULL restores the per-keyframe rule. */
int lvf_two_camera_set_block_weights(lvf_batch* b, const double* weight);
int lvf_imu_create(lvf_ctx* ctx, int n, const lvf_preint* pre, const int32_t* kf_i, const int32_t* kf_j,
                   lvf_batch** out);
/* mode 0 = RPZ (pitch,roll,z), 1 = YXY (yaw,x,y).  p/pa/pb/pc [n][3]: the scan point and its three
 * map neighbours (association.cpp:303-314); the unit normal (pa-pb)x(pa-pc) is computed on device. */
int lvf_lidar_plane_create(lvf_ctx* ctx, int mode, int n, const double* p, const double* pa, const double* pb,
                           const double* pc, const double* Twc1, double weight, lvf_batch** out);
/* Weak-constraint pose priors of a window (backend.cpp:164-178).  Block i with kf_a[i] >= 0 is
 * PoseGraphError<6,7,7>(Twc1 = pose[kf_a], Twc2 = pose[kf_b]) with target[i][0..6) = rpyxyz_ (pose_error.hpp:13-17,
 * see lvf_relative_rpyxyz); with kf_a[i] < 0 it is PoseError<6,7>(pose[kf_b]) with target[i][0..7) = the origin pose
 * (pose_error.hpp:55-76); with kf_a[i] == -2 it is RError<4,7>(pose[kf_b]) with target[i][0..4) = the stored quaternion (x,y,z,w)
 * (pose_error.hpp:88-110; residual rows 4,5 and their Jacobian rows are zero padding).
 * target is [n][7]; weight[n], v[n] are the functor's (weight, v) ctor arguments.
 * Parameter blocks: 0 = pose[kf_a] (all-zero Jacobian for PoseError blocks), 1 = pose[kf_b]; 6 residuals. */
int lvf_pose_prior_create(lvf_ctx* ctx, int n, const int32_t* kf_a, const int32_t* kf_b, const double* target,
                          const double* weight, const double* v, lvf_batch** out);
/* rpyxyz6 = SE3ToRpyxyz(last_pose^-1 * pose): the constant PoseGraphError's constructor stores (host arithmetic). */
int lvf_relative_rpyxyz(const double* last_pose, const double* pose, double* rpyxyz6);
int lvf_batch_destroy(lvf_batch* b);
int lvf_batch_size(const lvf_batch* b);
int lvf_batch_num_param_blocks(const lvf_batch* b);

/* Batched CostFunction::Evaluate: residuals (+ Jacobians if want_jacobians) in Ceres layout, left in HBM.
 * st supplies the parameter blocks.  For lidar batches st may be NULL and rpyxyz (host, 6 doubles — the
 * caller's LIVE array, lidar_error.hpp:52,87) is read at call time. */
int lvf_batch_evaluate(lvf_batch* b, const lvf_state* st, const double* rpyxyz, int want_jacobians);
/* Copies of the last evaluate's outputs.  block = parameter-block index in the functor's template order.
 * Sizes: residuals n*R doubles; jacobian(block) n*R*size_block doubles (row-major R x size per block). */
int lvf_batch_download_residuals(lvf_batch* b, double* host);
int lvf_batch_download_jacobian(lvf_batch* b, int block, double* host);
/* lidar batches only: the device-computed unit normals [n][3] */
int lvf_batch_download_normals(lvf_batch* b, double* host);
/* device pointers of the same buffers (valid until the batch is destroyed) */
void* lvf_batch_residuals_dev(lvf_batch* b);
void* lvf_batch_jacobian_dev(lvf_batch* b, int block);

/* Batched Problem::Evaluate surface (upstream ceres::Problem::Evaluate with apply_loss_function = true): residuals [n][R] with the loss
 * function's Corrector applied (visual batches: rows scaled by sqrt(rho'(|r|^2)) of HuberLoss(huber_a); huber_a <= 0 or non-visual
 * batches: unchanged) and Jacobians [n][R][L] in LOCAL coordinates, the block's parameter blocks concatenated in functor order with every
 * 7-sized pose block reduced to its 6 tangent columns (ProductParameterization(EigenQuaternion, Identity3), backend.cpp:99-101); absent
 * pose blocks (PoseError / RError priors) are zero.  L = lvf_batch_local_columns(b).  jacobians_local may be NULL.  Host outputs. */
int lvf_batch_local_columns(const lvf_batch* b);
int lvf_batch_evaluate_local(lvf_batch* b, const lvf_state* st, double huber_a, double* residuals, double* jacobians_local);

/* ---- IMU pre-integration on device ---------------------------------------------------------- */
/* n independent keyframe pairs; pair k owns samples[offset[k] .. offset[k+1]) rows of (dt, acc[3], gyr[3]);
 * acc0/gyr0 [n][3] are the measurements latched by the first Append; ba/bg [n][3] the linearisation biases;
 * noise4 = (ACC_N, GYR_N, ACC_W, GYR_W).  out[n] receives the propagated state. */
int lvf_preintegrate(lvf_ctx* ctx, int n, const int32_t* offset, const double* samples, const double* acc0,
                     const double* gyr0, const double* ba, const double* bg, const double* noise4, lvf_preint* out);

/* ---- scan-to-map association (float32, bit-exact 3-NN) --------------------------------------- */
/* map cloud: M points, xyz at the start of each `stride_floats`-float record (pcl::PointXYZI: 8).
 * Builds a uniform-grid index on device.  max_radius2 = the largest squared gate that will be queried
 * (e.g. resolution^2*100); it only sizes the cells, any thr may be queried later. */
int lvf_map_create(lvf_ctx* ctx, const float* map_xyz, int M, int stride_floats, float max_radius2, lvf_map** out);
/* n indices at once: out[i] is what lvf_map_create(ctx, map_xyz[i], M[i], stride_floats, max_radius2[i]) builds (the same pyramid, the same
 * association results), with the host waits shared between the maps — the old-frame maps of a set of loop-closure candidates
 * (relocator.cpp:196-206 -> Mapping::Relocate, mapping.cpp:251-262: one BuildOldMapFrame + kd-tree build per candidate and cloud).
 * On error no map is returned (out[i] = NULL for all i). */
int lvf_map_create_batch(lvf_ctx* ctx, int n, const float* const* map_xyz, const int* M, int stride_floats, const float* max_radius2,
                         lvf_map** out);
int lvf_map_destroy(lvf_map* m);
int lvf_scan_create(lvf_ctx* ctx, const float* scan_xyz, int Q, int stride_floats, lvf_scan** out);
int lvf_scan_destroy(lvf_scan* s);
/* For every scan point: point = SE3TransformPoint<float>(pose.cast<float>(), p); its 3 nearest map points by
 * squared L2 in float ((dx*dx+dy*dy)+dz*dz, no FMA), ties by ascending map index; valid iff all three
 * d2 < thr.  Outputs stay in HBM (owned by the scan).  idx/d2 are exact for valid points; for invalid
 * points they hold the best candidates found inside the gate radius (or -1/inf).  thr = +inf asks for the
 * exact 3-NN of every point. */
int lvf_knn3(lvf_map* m, lvf_scan* s, const double* pose, float thr);
int lvf_scan_download(lvf_scan* s, int32_t* idx3, float* d2_3, uint8_t* valid);
/* Diagnostic only: per-point search statistics and the grid pyramid levels4[L][4] = {cell, nx, ny, nz}; the caller provides room for 8
 * levels (the pyramid halves the cell per level, at most 8 levels).
 *   lvf_knn3_debug_stats : stats4[Q][4] = {candidates, range lookups, last grid level, shells}.
 *   lvf_knn3_debug_stats2: stats[Q][stride], 4 <= stride <= 6, the same four + {start, end} (the 100 MHz wall clock when the point's wave
 *                          began and finished, low 31 bits).  The record width is an ARGUMENT: it grew once under a fixed name. */
int lvf_knn3_debug_stats(lvf_map* m, lvf_scan* s, const double* pose, float thr, int32_t* stats4, float* levels4,
                         int* n_levels);
int lvf_knn3_debug_stats2(lvf_map* m, lvf_scan* s, const double* pose, float thr, int32_t* stats, int stride, float* levels4,
                          int* n_levels);

/* Test hook: the stable (key, value) pair sort under lvf_cloud_voxel_filter on host arrays; only the low key_bits bits of a key order the
 * pairs, equal keys keep their input order.  Not part of the reference surface. */
int lvf_debug_sort_pairs_u32(lvf_ctx* ctx, const uint32_t* keys, const int32_t* vals, int n, int key_bits, uint32_t* keys_out,
                             int32_t* vals_out);

/* ---- map-cloud maintenance on device (the steps immediately before the association) ------------- */
/* A cloud is n x (x, y, z, intensity) float32 in HBM (pcl::PointXYZI payload).  points: strided host records, xyz at the
 * start, intensity at float offset `intensity_offset` (4 for pcl::PointXYZI; -1 = none -> 0). */
int lvf_cloud_create(lvf_ctx* ctx, const float* points, int n, int stride_floats, int intensity_offset, lvf_cloud** out);
int lvf_cloud_destroy(lvf_cloud* c);
int lvf_cloud_size(const lvf_cloud* c);
int lvf_cloud_download(const lvf_cloud* c, float* xyzi /* [n][4] */);
/* Mapping::MergeScan / ToWorld / Sensor2Robot (mapping.cpp:193-220, association.cpp:236-247): out_i = SE3TransformPoint<float>
 * (pose.cast<float>(), in_i), intensity copied.  Bit-exact float arithmetic (normalise-then-rotate polynomial, no FMA). */
int lvf_cloud_transform(const lvf_cloud* in, const double* pose, lvf_cloud** out);
/* `merged += pointclouds[t]` of BuildMapFrame / BuildOldMapFrame (mapping.cpp:78-137): device-side concatenation. */
int lvf_cloud_concat(lvf_ctx* ctx, const lvf_cloud* const* parts, int n_parts, lvf_cloud** out);
/* FeatureAssociation::AlignScan (association.cpp:39-64): the keyframe's sweep [time - cycle/2, time + cycle/2] cut out of two consecutive raw
 * revolutions pc1 (stamped stamp1) + pc2 (stamp2 > stamp1) with the reference's index arithmetic (double, truncated); output points carry
 * intensity 0 (copyPointCloud from PointXYZ).  *aligned = 0 and an empty cloud where the reference returns false (sweep not covered). */
int lvf_cloud_align_scan(const lvf_cloud* pc1, double stamp1, const lvf_cloud* pc2, double stamp2, double cycle_time, double time, lvf_cloud** out,
                         int* aligned);
/* pcl::VoxelGrid<PointXYZI> with a cubic leaf (association.cpp:210-215, 222-224): per-voxel centroid of all four fields (float accumulation
 * over the voxel's points in ascending input index — reproducible bit for bit), voxels in ascending (i + j*dx + k*dx*dy) order. */
int lvf_cloud_voxel_filter(const lvf_cloud* in, float leaf, lvf_cloud** out);
/* pcl::RadiusOutlierRemoval (association.cpp:217-221): keeps a point iff more than min_neighbors points (itself
 * included) lie at squared distance < radius^2; input order preserved. */
int lvf_cloud_radius_outlier_filter(const lvf_cloud* in, float radius, int min_neighbors, lvf_cloud** out);
/* FeatureAssociation::SegmentGround (association.cpp:249-268): RANSAC plane (all hypotheses scored in one launch, PCL's
 * adaptive stopping rule applied in hypothesis order), least-squares refit, inliers extracted in input order.
 * coefficients4 (may be NULL) = (nx, ny, nz, d) with nz >= 0; iterations_used (may be NULL) = hypotheses consumed.
 * Sampling is splitmix64(seed, hypothesis, draw) — PCL's boost::mt19937 stream is not reproducible without PCL. */
int lvf_cloud_segment_plane(const lvf_cloud* in, float distance_threshold, int max_iterations, uint64_t seed, lvf_cloud** out,
                            double* coefficients4, int* iterations_used);
/* ---- LiDAR feature extraction of one keyframe scan: FeatureAssociation::Process, association.cpp:86-235 - */
typedef struct lvf_lidar_params {
  int num_scans, horizon_scan;       /* 64, 1800 (config/kitti.yaml:35-36); num_scans <= 64 */
  float ang_res_y, ang_bottom;       /* 0.427, 24.9 */
  int ground_rows;                   /* 60 */
  double cycle_time;                 /* 0.1036 (a double in FeatureAssociation) */
  float min_range, max_range, resolution;   /* 5, 30, 0.2 */
  uint64_t ransac_seed;              /* SegmentGround's sampler (see lvf_cloud_segment_plane) */
} lvf_lidar_params;
void lvf_lidar_params_default(lvf_lidar_params* p);
/* optional taps for parity tests: host arrays (may each be NULL) sized num_scans*horizon_scan (label/ground/range) or
 * num_scans*horizon_scan*4 floats (ground_raw / surf_raw = ExtractFeatures' picks before the PCL filters) */
typedef struct lvf_lidar_extract_debug {
  int n_filtered, n_segmented, n_ground_raw, n_surf_raw;
  int32_t* label_mat; int8_t* ground_mat; float* range_mat; float* ground_raw; float* surf_raw;
} lvf_lidar_extract_debug;
/* points: the raw sensor-frame scan (xyz at the start of each stride_floats record, scan order = acquisition order);
 * extrinsic7 = Lidar::Get()->extrinsic (sensor -> robot).  ground_out / surf_out = frame->feature_lidar->points_ground /
 * points_surf as NEW device clouds (robot frame; intensity = ring + relative time as the reference writes it). */
int lvf_lidar_extract(lvf_ctx* ctx, const float* points, int n, int stride_floats, const lvf_lidar_params* prm, const double* extrinsic7,
                      lvf_cloud** ground_out, lvf_cloud** surf_out, lvf_lidar_extract_debug* dbg);
/* Test / A-B hook, process-wide: 1 = lvf_lidar_extract takes the host-counted path of rounds 2-4 (every stage's output count is read back
 * before the next stage is sized: 13 stream waits per scan) instead of the device-counted one (one wait); returns the previous setting.  The
 * environment variable LVF_EXTRACT_HOST_COUNTS=1 sets the initial value.  Not part of the reference surface. */
int lvf_debug_extract_host_counts(int on);
/* kNN index / query scan straight from device-resident clouds (no host round trip) */
int lvf_map_create_from_cloud(const lvf_cloud* c, float max_radius2, lvf_map** out);
/* ... of n device-resident clouds in one call (lvf_map_create_batch's shared waits and launches, no upload): the old-frame maps of a set of
 * loop-closure candidates whose per-keyframe clouds are kept as lvf_cloud (relocator.cpp:196-206 -> mapping.cpp:251-262) */
int lvf_map_create_batch_from_clouds(lvf_ctx* ctx, int n, const lvf_cloud* const* clouds, const float* max_radius2, lvf_map** out);
int lvf_scan_create_from_cloud(const lvf_cloud* c, lvf_scan** out);

/* ---- one scan-to-map sub-problem (3-DoF LM on device) ---------------------------------------- */
typedef struct lvf_icp_options {
  int mode;                /* 0 = ground/RPZ (TrivialLoss), 1 = surf/YXY (HuberLoss(0.1)) */
  float thr;               /* squared distance gate */
  double weight;           /* frame->weights.lidar_ground / lidar_surf */
  double huber_a;          /* <=0: TrivialLoss (association.cpp:272), 0.1 for surf (:330) */
  double prior_weight;     /* PoseErrorRPZ/YXY weight = |features_left| * weights.visual; 0 = relocate mode */
  int max_num_iterations;  /* 4 (mapping.cpp:161) */
} lvf_icp_options;
typedef struct lvf_icp_summary {
  double initial_cost, final_cost;
  int num_residual_blocks; /* valid correspondences (+1 if prior) — Summary::num_residual_blocks_reduced */
  int num_iterations;
  int num_successful_steps;
} lvf_icp_summary;
/* Runs association at frame_pose (the frame's current pose, cast to float as association.cpp:287), builds the
 * point-to-plane problem against map_pose (Twc1) and solves it on device; rpyxyz (host, 6 doubles =
 * se32rpyxyz(map_pose^-1 * frame_pose), mapping.cpp:154) is updated IN PLACE like the reference's stack array
 * (mapping.cpp:153-165).  The caller then sets frame->pose = map_pose * rpyxyz2se3(rpyxyz) (mapping.cpp:164). */
int lvf_icp_solve(lvf_map* m, lvf_scan* s, const double* map_pose, const double* frame_pose, double* rpyxyz,
                  const lvf_icp_options* opt, lvf_icp_summary* summary);

/* ---- one frame's scan-to-map update: Mapping::Optimize's body (mapping.cpp:147-178) / Mapping::Relocate (:251-300) ---- */
typedef struct lvf_scan_match_options {
  float thr_ground, thr_surf;      /* resolution^2 * 100 / * 25 (association.cpp:285,343) */
  double weight_ground, weight_surf; /* frame->weights.lidar_ground / lidar_surf — FLOAT fields in the reference (adapt/weights.h:10-12): pass the
                                      * float's value, e.g. (double)0.01f, which is what lvf_scan_match_options_default sets */
  double huber_surf;               /* 0.1 (association.cpp:330); ground uses TrivialLoss */
  double prior_weight;             /* |features_left| * weights.visual; 0 = relocate mode (association.cpp:321,379) */
  int outer_iterations;            /* 1 = Mapping::Optimize, 4 = Mapping::Relocate (mapping.cpp:264) */
  int max_num_iterations;          /* 4 */
} lvf_scan_match_options;
typedef struct lvf_scan_match_result {
  double pose[7];                  /* the frame's pose after the update */
  double relative_o_c[7];          /* last_pose^-1 * pose (mapping.cpp:298); = pose when last_pose is NULL */
  double score_ground, score_surf; /* mapping.cpp:279-280, :293-294 (last outer iteration) */
  int score;                       /* (int)(score_ground + score_surf): Mapping::Relocate's return value */
  lvf_icp_summary ground, surf;    /* summaries of the last ground / surf sub-problem */
} lvf_scan_match_result;
void lvf_scan_match_options_default(lvf_scan_match_options* o, double lidar_resolution);
/* Either (map, scan) pair may be NULL (empty cloud: the reference skips that sub-problem).  All four handles must
 * belong to one context. */
int lvf_scan_match(lvf_map* map_ground, lvf_scan* scan_ground, lvf_map* map_surf, lvf_scan* scan_surf, const double* map_pose,
                   const double* frame_pose, const double* last_pose, const lvf_scan_match_options* opt,
                   lvf_scan_match_result* result);

/* Many scan-to-map updates of ONE context advanced side by side: the loop-closure candidates Relocator::Relocate evaluates one after the
 * other (relocator.cpp:196-206 -> Mapping::Relocate, mapping.cpp:251-300: no shared mutable state between candidates).  Every launch of
 * the chain covers all candidates (the candidate is a grid dimension), each candidate's pose / rpyxyz / scores live in a device record
 * between its sub-problems, and the n results are read back once.  results[i] equals lvf_scan_match on candidate i (which IS this call
 * with n = 1).  A candidate's pairs may be NULL like lvf_scan_match's; scan handles must not repeat across candidates (a scan owns its
 * association outputs).  best (may be NULL) = the reference's choice: the LAST candidate whose score - relocate_base_score is > 0 and
 * >= every earlier one (relocator.cpp:181,198-204; 20 in the reference), or -1. */
typedef struct lvf_scan_match_job {
  lvf_map* map_ground; lvf_scan* scan_ground; lvf_map* map_surf; lvf_scan* scan_surf;
  double map_pose[7], frame_pose[7], last_pose[7];
  int has_last_pose;               /* 0: relative_o_c = pose (lvf_scan_match's last_pose == NULL) */
} lvf_scan_match_job;
int lvf_scan_match_batch(lvf_ctx* ctx, const lvf_scan_match_job* jobs, int n, const lvf_scan_match_options* opt, int relocate_base_score,
                         lvf_scan_match_result* results, int* best);

/* The same device-resident 3-DoF solve over a caller-built lidar batch (lvf_lidar_plane_create): what adapt::Solve
 * does for the problem ScanToMapWithGround/Segmented assembled (mapping.cpp:157-163, :270-296).  mode, weight and Twc1
 * are the batch's; opt supplies huber_a, prior_weight and max_num_iterations (opt->mode/weight/thr are ignored). */
int lvf_lidar_solve(lvf_batch* lidar_batch, double* rpyxyz, const lvf_icp_options* opt, lvf_icp_summary* summary);

/* PoseErrorRPZ / PoseErrorYXY <3,1,1,1> (pose_error.hpp:135-190) evaluated stand-alone.  target3 and x3 are in PARAMETER
 * order — (pitch, roll, z) for mode 0, (yaw, x, y) for mode 1; residuals3 in the functor's order — (roll, pitch, z) /
 * (yaw, x, y); jacobians9 (may be NULL) = three 3x1 blocks, one per parameter block, concatenated. */
int lvf_prior3_evaluate(lvf_ctx* ctx, int mode, const double* target3, double weight, const double* x3, double* residuals3,
                        double* jacobians9);

/* ---- sliding-window BA problem (adapt::Problem + adapt::Solve) --------------------------------- */
typedef struct lvf_solver_options {
  int max_num_iterations;          /* Ceres default 50; UpdateFrontend uses 1 (backend.cpp:264) */
  double max_solver_time_in_seconds; /* backend.cpp:208 ; <=0 = unlimited */
  double huber_a;                  /* shared HuberLoss(1.0) on visual blocks (backend.cpp:98) */
  double initial_trust_region_radius; /* 1e4 */
  double function_tolerance;       /* 1e-6 */
  double gradient_tolerance;       /* 1e-10 */
  double parameter_tolerance;      /* 1e-8 */
  double min_relative_decrease;    /* 1e-3 */
} lvf_solver_options;
typedef struct lvf_solver_summary {
  double initial_cost, final_cost;
  int num_iterations, num_successful_steps;
  int num_residual_blocks;
  int termination; /* 0 convergence, 1 no_convergence (iteration/time cap), 2 failure */
  int num_unsuccessful_steps; /* rejected + invalid steps (Summary::num_unsuccessful_steps) */
  int termination_reason;     /* LVF_WHY_*: which test of ceres::Solve's TrustRegionMinimizer ended the loop */
  int hand_over_retries;      /* iterations this problem has re-run (over its lifetime) because an in-launch hand-over between chained sparse levels
                               * timed out on a busy GPU; each was repeated with un-chained launches — results are unaffected, only slower */
} lvf_solver_summary;
/* ceres::Solve's termination tests in the order the loop applies them (upstream trust_region_minimizer.cc; the test oracle restates the same loop):
 * before a step — gradient_max_norm <= gradient_tolerance, radius < 1e-32 (both CONVERGENCE, the pass does not count as an iteration);
 * after the linear solve — 5 consecutive invalid steps (solver failure or model_cost_change <= 0; FAILURE, else radius *= 0.5);
 * with the candidate — step_norm <= parameter_tolerance (x_norm + parameter_tolerance), then |cost - candidate_cost| <= function_tolerance cost
 * (both CONVERGENCE, the candidate is NOT taken); after the step — num_iterations >= max_num_iterations (NO_CONVERGENCE), radius < 1e-32. */
enum { LVF_WHY_NONE = 0, LVF_WHY_GRADIENT = 1, LVF_WHY_PARAMETER = 2, LVF_WHY_FUNCTION = 3, LVF_WHY_MIN_RADIUS = 4, LVF_WHY_MAX_ITERATIONS = 5,
       LVF_WHY_INVALID_STEPS = 6, LVF_WHY_TIME = 7,
       LVF_WHY_HANDOVER = 8 /* internal: the device loop stopped for a hand-over retry; only reported if the retry could not run either */ };
void lvf_solver_options_default(lvf_solver_options* o);
/* The problem borrows the state and the batches (they must outlive it).  Any of the batch pointers may
 * be NULL.  Pose blocks use ProductParameterization(EigenQuaternion, Identity3) (backend.cpp:99-101). */
int lvf_problem_create(lvf_ctx* ctx, lvf_state* st, lvf_batch* two_camera, lvf_batch* two_frame,
                       lvf_batch* pose_only, lvf_batch* imu, lvf_problem** out);
int lvf_problem_destroy(lvf_problem* p);
/* Adds (or with NULL removes) the window's pose-prior batch (ProblemType::Other blocks with no loss function). */
int lvf_problem_set_pose_priors(lvf_problem* p, lvf_batch* pose_priors);
int lvf_problem_set_pose_constant(lvf_problem* p, int kf, int is_constant);
/* SetParameterBlockConstant on keyframe kf's velocity / accelerometer-bias / gyroscope-bias blocks (Environment::Optimize holds all of
 * them, and the previous frame's, constant: environment.cpp:62-68): the ImuError factors keep their residuals, the blocks get no
 * Jacobian columns and are left out of the step norm's x_norm like Ceres' reduced program does. */
int lvf_problem_set_vbb_constant(lvf_problem* p, int kf, int v_constant, int ba_constant, int bg_constant);
/* Problem::Evaluate-style cost at the current state: 0.5 * sum rho(|r|^2). */
int lvf_problem_cost(lvf_problem* p, const lvf_solver_options* o, double* cost);
/* One Levenberg-Marquardt iteration from the current state with trust-region radius *radius (updated);
 * accepted != 0 if the step was taken.  Deterministic per-iteration parity point. */
int lvf_problem_lm_iteration(lvf_problem* p, const lvf_solver_options* o, double* radius, double* decrease_factor,
                             double* cost_before, double* cost_after, int* accepted);
int lvf_problem_solve(lvf_problem* p, const lvf_solver_options* o, lvf_solver_summary* summary);
/* Debug/parity taps of the last linearisation: reduced (Schur) system S [d x d row-major], rhs [d],
 * d = 15 * n_kf (pose tangent 6 | v 3 | ba 3 | bg 3 per keyframe).  Available after lvf_problem_lm_iteration (the per-call API keeps the
 * normal equations of its iteration); lvf_problem_solve / lvf_problem_batch_solve clear them at the end of every iteration for the
 * next one, so after those the tap returns LVF_ERR_STATE ("no linearisation yet"). */
int lvf_problem_reduced_dim(lvf_problem* p);
int lvf_problem_download_reduced(lvf_problem* p, double* S, double* rhs);

/* ---- a batch of independent windows on ONE GPU -------------------------------------------------------------------------------------
 * W windows (each its own lvf_state / batches / lvf_problem, all of one context) advanced by ONE chain of launches per LM iteration: every
 * kernel of the iteration takes the window as blockIdx.y and reads that window's argument block from a device table.  A single window's
 * iteration is a chain of small launches that leaves most of the 256 CUs idle; a batch fills the same launches W times over — the shape of
 * the reference's independent-window clients (RL environments: src/environment.cpp:18-115; loop-closure candidates: relocator.cpp:196-206)
 * and of what one GPU of the 8-GPU sharding works on.  The LM loop of every window runs on device (accept / reject, trust region and
 * termination per window; a finished window's launches return immediately).  Per-window results equal lvf_problem_solve /
 * lvf_problem_lm_iteration on that window alone up to summation order (a batch of more than one window sums its Schur complement over
 * wider landmark slices: same accept / reject decisions and iteration counts, states within ~1e-12 relative).  Windows that cannot use the table launches (no sorted TwoFrame work list, pose priors,
 * no IMU blocks) make the batch fall back to running the windows' own chains back to back on the context's stream.
 * All result arrays have one entry per window, in the order of `problems`. */
int lvf_problem_batch_create(lvf_ctx* ctx, lvf_problem* const* problems, int n, lvf_problem_batch** out);
/* The batch BORROWS its problems and changes nothing of theirs (its wider Schur slices and their work lists are the batch's own): a window
 * solved alone, in a batch, and alone again runs the same arithmetic the first and the third time.  Destroy order is free: a batch whose
 * member was destroyed first is marked orphaned (later calls on it return LVF_ERR_STATE) and may still be destroyed. */
int lvf_problem_batch_destroy(lvf_problem_batch* b);
int lvf_problem_batch_size(const lvf_problem_batch* b);
/* 1: table launches (one chain for all windows), 0: fallback, -1: error */
int lvf_problem_batch_uses_tables(lvf_problem_batch* b, const lvf_solver_options* o);
int lvf_problem_batch_lm_iteration(lvf_problem_batch* b, const lvf_solver_options* o, double* radius, double* decrease_factor, double* cost_before,
                                   double* cost_after, int* accepted);
int lvf_problem_batch_solve(lvf_problem_batch* b, const lvf_solver_options* o, lvf_solver_summary* summaries);

/* Measurement tap (bench.py's roofline lines): `reps` LM iterations from the current state, timed on the library's stream.  us[k] = average
 * duration of stage k in microseconds (the sum of its kernels' own durations: see lvf_problem_stage_times2), launches[k] (may be NULL) =
 * kernel launches it consists of; k < lvf_problem_stage_count(), names from lvf_problem_stage_name. */
int lvf_problem_stage_count(void);
const char* lvf_problem_stage_name(int stage);
int lvf_problem_stage_times(lvf_problem* p, const lvf_solver_options* o, double radius, int reps, double* us, int* launches);
/* The same with both clocks: us[k] = the sum of the stage's KERNEL durations (a start / stop event pair recorded with every launch — the
 * dispatch's own timestamps, what rocprofv3 --kernel-trace reports); spans_us[k] (may be NULL) = the time between the events that bracket
 * the stage on the stream (kernels + gaps + the markers' own cost). */
int lvf_problem_stage_times2(lvf_problem* p, const lvf_solver_options* o, double radius, int reps, double* us, double* spans_us, int* launches);

/* Problem::Evaluate's gradient at the current state: J^T r with the Corrector applied and pose blocks in tangent coordinates.
 * gc[15 n_kf] = (6 x n_kf pose tangents | 9 x n_kf (v, ba, bg)); gl[n_lm] (may be NULL) = the inverse-depth entries.  Constant poses: 0. */
int lvf_problem_gradient(lvf_problem* p, const lvf_solver_options* o, double* gc, double* gl);
/* Test hook (tests/test_gpu_solver.py): the next n chained hand-overs of this problem wait for a producer that never arrives and time out
 * after 20 us, so the retry path (LVF_WHY_HANDOVER -> un-chained re-run, lvf_solver_summary::hand_over_retries) can be exercised on an
 * idle GPU.  Also re-enables chaining for the problem.  Not part of the reference surface. */
int lvf_problem_debug_force_handover_timeout(lvf_problem* p, int n);
/* Diagnostic (environment LVF_LM_HISTORY=1 when the problem's launch chain is built): out512 receives eight doubles per closed pass of the
 * last device-loop solve, slot = iteration & 63: {iteration, cost at the point, candidate cost, model cost change, accepted, failure flag,
 * trust-region radius used, gradient max norm}.  LVF_ERR_STATE without the environment variable. */
int lvf_problem_debug_history(lvf_problem* p, double* out512);

/* ---- loop-correction tail: Relocator::UpdateNewSubmap / PoseGraph::ForwardUpdate ---------------------------------- */
/* RelocateRError <7,4> (pose_error.hpp:192-222) batched: block i = RelocateRError(relocated[i], unrelocated[i]) evaluated at the shared
 * quaternion q4 (x,y,z,w, NOT normalised by the functor).  residuals [n][7]; jacobians [n][7][4] row-major ambient, or NULL. */
int lvf_relocate_r_evaluate(lvf_ctx* ctx, int n, const double* relocated, const double* unrelocated, const double* q4, double* residuals,
                            double* jacobians);
/* The rotation solve of Relocator::UpdateNewSubmap (relocator.cpp:251-267): one quaternion parameter under EigenQuaternionParameterization,
 * n RelocateRError blocks, no loss, LM (the DENSE_QR solve of 3 tangent unknowns) — the whole loop is one device launch.  q4 (x,y,z,w) is
 * updated IN PLACE like `para` (identity in the reference); on failure (summary->termination == 2) it is left untouched. */
int lvf_relocate_rotation_solve(lvf_ctx* ctx, int n, const double* relocated, const double* unrelocated, double* q4, const lvf_solver_options* o,
                                lvf_solver_summary* summary);
/* PoseGraph::ForwardUpdate (pose_graph.cpp:245-252; also Backend::UpdateFrontend, backend.cpp:256): pose <- transform * pose (Sophus SE3
 * product: Hamilton product re-normalised, t_T + R(q_T) t) and Vw <- R(q_T) Vw for n keyframes; host arrays updated in place, vw may be NULL. */
int lvf_forward_update(lvf_ctx* ctx, const double* transform7, int n, double* poses, double* vw);
/* the same on a device-resident state: keyframes [first_kf, n_kf) of st (poses and velocities), nothing crosses PCIe but the transform */
int lvf_state_forward_update(lvf_state* st, const double* transform7, int first_kf);

/* ---- multi-GPU: the path's only exchange, for a C / C++ host ----------------------------------------------------------- */
/* Independent windows / loop-closure candidates shard one per GPU: one process per GPU, one lvf_ctx each, no data-path collective.  The
 * single exchange is an all-gather of fixed-size records (score, relative_o_c[7], candidate id: relocator.cpp:196-206) over RCCL / xGMI.
 * Rank 0 calls lvf_comm_get_unique_id and hands the 128 bytes to the other ranks out of band (file, socket, MPI_Bcast); every rank then
 * calls lvf_comm_create with the same id.  id128 == NULL with world_size == 1 gives a communicator that needs no RCCL.
 * lvf_comm_allgather: every rank contributes n_doubles (equal on all ranks); recv[world_size][n_doubles] in rank order (host buffers). */
#define LVF_COMM_ID_BYTES 128
int lvf_comm_get_unique_id(void* id128);
int lvf_comm_create(lvf_ctx* ctx, int world_size, int rank, const void* id128, lvf_comm** out);
int lvf_comm_destroy(lvf_comm* c);
int lvf_comm_world_size(const lvf_comm* c);
int lvf_comm_rank(const lvf_comm* c);
int lvf_comm_allgather(lvf_comm* c, const double* send, int n_doubles, double* recv);

/* ---- persistent sliding window (Backend::BuildProblem's assembly, kept incrementally) -------------- */
/* The window mirrors Map's active keyframes, their features_left and the landmarks behind them as flat arrays that the
 * front-end updates as it goes; lvf_window_solve assembles the block lists with BuildProblem's rules (backend.cpp:96-183) in
 * frame order / ascending landmark id, re-fills persistent device batches (no per-tick allocation) and runs adapt::Solve's
 * device loop.  Keyframe and landmark ids are the caller's (frame->id, landmark->id); keyframe ids must increase. */
typedef struct lvf_window_options {
  double baseline;              /* Camera::baseline: a landmark deeper than 50 baselines is "Far" (camera.h:38-41) */
  int weak_visual_threshold;    /* 20 (backend.cpp:166) */
  double prior_weight, prior_v; /* PoseGraphError / PoseError (.., 100, 0)  (backend.cpp:170,175) */
  int device_assembly;          /* 1 (default): the per-tick block lists are assembled ON THE DEVICE from resident feature / landmark tables (only
                                 * what changed since the last tick is uploaded); 0: the host walks every feature and uploads the lists */
} lvf_window_options;
void lvf_window_options_default(lvf_window_options* o);
int lvf_window_create(lvf_ctx* ctx, const lvf_camera* left, const lvf_camera* right, const lvf_window_options* opt, lvf_window** out);
int lvf_window_destroy(lvf_window* w);
/* w_visual = frame->weights.visual.  The reference keeps its weights as FLOAT (adapt/weights.h:10): pass the float's value.  PoseOnly / TwoFrame
 * blocks of the keyframe take it as it is (backend.cpp:129, :138); its TwoCamera blocks take `5 * frame->weights.visual` formed as the reference
 * forms it, a float product widened to double: (double)(5.0f * (float)w_visual) (backend.cpp:123). */
int lvf_window_add_keyframe(lvf_window* w, int64_t kf_id, const double* pose7, double w_visual);
/* marks the frame good_imu with its Vw / linearised biases; pre (may be NULL) = frame->preintegration from the previous keyframe */
int lvf_window_set_imu(lvf_window* w, int64_t kf_id, const double* vel3, const double* ba3, const double* bg3, const lvf_preint* pre);
/* a landmark triangulated in keyframe birth_kf_id: its left feature there and first_observation (right image) */
int lvf_window_add_landmark(lvf_window* w, int64_t lm_id, int64_t birth_kf_id, const double* left_ob2, const double* right_ob2, double inv_depth);
int lvf_window_add_observation(lvf_window* w, int64_t lm_id, int64_t kf_id, const double* ob2);
int lvf_window_remove_observation(lvf_window* w, int64_t lm_id, int64_t kf_id);   /* outlier rejection, backend.cpp:232-243 */
int lvf_window_slide(lvf_window* w, int64_t first_active_kf_id);                  /* Map::GetKeyFrames(finished); also forgets landmarks nobody observes */
/* Backend::Optimize's outlier gate (backend.cpp:185-190, :229-245): every feature that is not its landmark's first observation is
 * re-projected on device with weight 1 (PoseOnly residual pass at the window's current poses / inverse depths) and removed from the window
 * when the pixel error exceeds max_px (10 in the reference).  Up to `capacity` removed (landmark id, keyframe id) pairs are reported. */
int lvf_window_reject_outliers(lvf_window* w, double max_px, int64_t* removed_lm, int64_t* removed_kf, int capacity, int* n_removed);
int lvf_window_solve(lvf_window* w, const lvf_solver_options* o, lvf_solver_summary* summary);
int lvf_window_set_pose(lvf_window* w, int64_t kf_id, const double* pose7);       /* front-end / ForwardUpdate writes */
int lvf_window_get_pose(const lvf_window* w, int64_t kf_id, double* pose7);
int lvf_window_get_imu(const lvf_window* w, int64_t kf_id, double* vel3, double* ba3, double* bg3);
int lvf_window_get_inv_depth(const lvf_window* w, int64_t lm_id, double* inv_depth);
/* counts8 = {keyframes, landmarks in the problem, TwoCamera, TwoFrame, PoseOnly, ImuError, prior blocks, landmarks known} of the last solve */
int lvf_window_counts(const lvf_window* w, int32_t* counts8);
/* Test hook: the block lists of the last lvf_window_solve, kind 0 TwoCamera / 1 PoseOnly / 2 TwoFrame / 3 ImuError / 4 weak-constraint priors,
 * in the device batches' order with keyframe and landmark IDS: ids[capacity][3] = {landmark, first / previous keyframe, keyframe} (-1 where a
 * kind has none), vals[capacity][8] = {weight handed to the reference's Create, ob x, y, first (right) ob x, y, pw x, y, z}; *n = blocks of the
 * kind.  What tests compare with Backend::BuildProblem's own output (backend.cpp:96-183).  Not part of the reference surface. */
int lvf_window_debug_blocks(lvf_window* w, int kind, int capacity, int64_t* ids, double* vals, int* n);

#ifdef __cplusplus
}
#endif
#endif /* LVF_H_ */
