/*
 * vofod.h — C-ABI drop-in boundary for VoFOD's per-scan point-cloud hot path.
 *
 * The reference (ctu-mrs/vofod) exposes no C/FFI interface: its only plugin
 * surface is the pluginlib nodelet vofod::VoFOD (nodelets.xml:1-5,
 * src/vofod_nodelet.cpp:141-145).  This header declares the seam a maintainer
 * would cut inside that nodelet: every entry point replaces the *body* of one
 * member function of vofod::VoFOD (cited per function as file:line, relative
 * to the reference tree).  INTEGRATION.md shows the ~60-line shim that calls
 * these from processMsg()/raycast_cloud()/updateSeparatedBGClusters().
 *
 * Two libraries implement exactly this interface:
 *   - libvofod_hip.so    (vofod_amd/csrc, symbols vofod_*)         the product:
 *                         hand-written gfx950 HIP kernels + C++ host driver.
 *   - libvofod_oracle.so (oracle/,        symbols vofod_oracle_*)  the checker:
 *                         CPU restatement of the reference algorithm; test
 *                         infrastructure only, never linked by the product.
 *
 * Conventions
 *   - plain pointers + sizes only; caller owns every in/out buffer, the handle
 *     owns device memory; nothing allocated on one side is freed on the other.
 *   - every function returns a vofod_status (0 = ok) and never throws.
 *   - a handle is thread-safe: calls are serialised on an internal mutex in the
 *     order m_voxels_mtx would order them (vofod_nodelet.cpp:712,943,1146,1210,1530).
 *   - transforms are float[12], row-major 3x4 [R|t] (Eigen::Affine3f s2w_tf,
 *     vofod_nodelet.cpp:913-922).
 */
#ifndef VOFOD_H
#define VOFOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */

typedef enum vofod_status {
  VOFOD_OK = 0,
  VOFOD_ERR_INVALID_ARG = 1,
  VOFOD_ERR_SIZE_MISMATCH = 2,        /* cloud size != LUT size: vofod_nodelet.cpp:895-899, 1407-1411 */
  VOFOD_ERR_SENSOR_OUTSIDE_MAP = 3,   /* vofod_nodelet.cpp:1432, 1523-1526 (map left untouched by the DDA) */
  VOFOD_ERR_INDEX_OVERFLOW = 4,       /* voxel_grid_weighted.cpp:61-69 (output left empty) */
  VOFOD_ERR_CAPACITY = 5,             /* caller buffer too small; n_out holds the required size */
  VOFOD_ERR_DEVICE = 6,               /* HIP runtime error; see vofod_last_error_string */
  VOFOD_ERR_RAYCAST_NO_DETECTION = 7, /* no detection iteration since begin: vofod_nodelet.cpp:1531-1537 */
  VOFOD_ERR_RAYCAST_EMPTY = 8,        /* max raycast value is zero: vofod_nodelet.cpp:1542-1548 */
  VOFOD_ERR_PAUSED = 9,               /* raycast__pause / sepclusters__pause: :1400-1404, :1128-1132 */
  VOFOD_ERR_NOT_PENDING = 10,         /* *_finish without a matching *_begin */
  VOFOD_ERR_EMPTY = 11,               /* sepclusters: thresholded map cloud empty: :1155-1159 */
  VOFOD_ERR_MAP_RANGE = 12,           /* a weighted point fell outside the voxel map (vector::at would throw: voxel_map.cpp:116-117) */
  VOFOD_ERR_BUSY = 13                 /* a submitted batch (vofod_batch_submit) still reads the state this call would overwrite: collect it first */
} vofod_status;

typedef struct vofod_handle vofod_handle;

/* ------------------------------------------------------------ parameters */

/* Static parameters: the block loaded once in onInit (vofod_nodelet.cpp:168-230)
 * plus the sensor description (initialize_sensor_rosparam :422-438).
 * Offsets' z components are the *bottom* of the box exactly as in the yaml
 * files; the library adds size_z/2 as :204 and :212 do. */
typedef struct vofod_static_params {
  float voxel_size;                         /* voxel_map/voxel_size            (detection_params.yaml:17) */
  float score_init;                         /* voxel_map/scores/init           (:21) */
  float background_sufficient_points_ratio; /* (:9), used as vofod_nodelet.cpp:228-230 */
  float oparea_offset[3];                   /* operation_area/offset (sim.yaml:8-11) */
  float oparea_size[3];                     /* operation_area/size   (sim.yaml:12-15) */
  float exclude_offset[3];                  /* exclude_box/offset    (detection_params.yaml:76-79) */
  float exclude_size[3];                    /* exclude_box/size      (:80-83) */
  int32_t sensor_hrays;                     /* sensor/horizontal_rays = cloud width  */
  int32_t sensor_vrays;                     /* sensor/vertical_rays   = cloud height */
  float sensor_vfov;                        /* sensor/vertical_fov_angle, radians */
  const float* lut_directions;              /* 3*w*h floats, xyz interleaved, index row*w+col (xyz_lut_t :77-81, :1447);
                                               NULL -> simulated LUT of initialize_sensor_lut_simulation :374-420 */
  const float* lut_offsets;                 /* same layout; NULL -> zeros (:416) */
  const uint8_t* mask;                      /* w*h bytes, non-zero = ray may be cast when range==0 (load_mask :506-560);
                                               NULL -> all ones (:558) */
  int32_t device;                           /* HIP device ordinal (ignored by the oracle) */
  int32_t max_batch_frames;                 /* frames a single vofod_process_batch call may carry (>=1) */
} vofod_static_params;

/* Dynamic parameters: one field per key of DetectionParams.cfg:16-44 (the
 * "__" spelling is dynamic_reconfigure's for "/").  May be changed between any
 * two calls, as m_drmgr_ptr->config may. */
typedef struct vofod_dyn_params {
  double ground_points_max_distance;
  double output__position_sigma;
  double voxel_map__scores__point;
  double voxel_map__scores__unknown;
  double voxel_map__scores__ray;
  double voxel_map__thresholds__apriori_map;
  double voxel_map__thresholds__new_obstacles;
  double voxel_map__thresholds__sure_obstacles;
  double voxel_map__thresholds__frontiers;
  int32_t classification__min_points;
  double classification__max_size;
  double classification__max_distance;
  double classification__max_explore_distance;
  int32_t raycast__pause;
  int32_t raycast__new_update_rule;
  double raycast__max_distance;
  double raycast__min_intensity;
  double raycast__weight_coefficient;
  int32_t sepclusters__pause;
  double sepclusters__max_bg_distance;
  int32_t sepclusters__min_sure_points;
} vofod_dyn_params;

/* Fills both structs with config/detection_params.yaml + config/apriori_maps/sim.yaml
 * + config/sensors/os1-128.yaml (the values the reference's simulation demo runs with). */
void vofod_default_params(vofod_static_params* sp, vofod_dyn_params* dp);

/* ------------------------------------------------------------------ types */

enum { VOFOD_MEM_HOST = 0, VOFOD_MEM_DEVICE = 1 };

/* One organised LiDAR scan: pcl::PointCloud<ouster_ros::Point> (types.h:7-8),
 * height = vrays, width = hrays (vofod_nodelet.cpp:1407).  Columns are given as
 * base pointer + common byte stride, so the 48-byte ouster_ros::Point AoS
 * (x at +0, y +4, z +8, intensity +16, range +36 in ouster_ros >= 0.10) and a
 * packed SoA (stride 4) both work without a copy on the caller's side. */
typedef struct vofod_scan {
  const void* x;          /* float */
  const void* y;          /* float */
  const void* z;          /* float */
  const void* intensity;  /* float    (raycast only: vofod_nodelet.cpp:1446); may be NULL for process_scan */
  const void* range;      /* uint32 mm (raycast only: :1449, :1455-1456);     may be NULL for process_scan */
  size_t stride_bytes;
  int32_t width, height;
  int32_t memspace;       /* VOFOD_MEM_HOST or VOFOD_MEM_DEVICE (device pointers on the handle's device) */
  double stamp;           /* seconds; carried through, not interpreted */
} vofod_scan;

/* payload of vofod::PointXYZR (point_types.h:51-56): voxel centre + weight */
typedef struct vofod_point_xyzr {
  float x, y, z;
  uint32_t range;
} vofod_point_xyzr;

/* pcl::PointXYZI payload of the debug clouds (voxel_map.h pt_t) */
typedef struct vofod_point_xyzi {
  float x, y, z;
  float intensity;
} vofod_point_xyzi;

/* vofod/Detection (msgs/Detection.msg:1-12), filled as vofod_nodelet.cpp:972-985 */
typedef struct vofod_detection {
  uint32_t id;
  uint32_t frame;                /* index of the scan inside a batch (0 for process_scan) */
  uint64_t n_points;
  double confidence;
  double detection_probability;
  double position[3];
  double covariance[9];
} vofod_detection;

enum { VOFOD_CLASS_MAV = 0, VOFOD_CLASS_UNKNOWN = 1, VOFOD_CLASS_INVALID = 2, VOFOD_CLASS_NONE = -1 };

/* One Euclidean cluster of the weighted cloud.  Canonical order:
 * size descending (extract_clusters' reverse sort), ties by smallest member. */
typedef struct vofod_cluster_info {
  uint32_t first_member;   /* smallest member index into the weighted cloud == label */
  uint32_t n_points;
  int32_t is_close;        /* findCloseFarClusters: vofod_nodelet.cpp:727-748 */
  int32_t cclass;          /* classify_cluster :1648-1730; VOFOD_CLASS_NONE for close clusters */
  float aabb_min[3];
  float aabb_max[3];
  float obb_center[3];     /* NaN when not evaluated */
  float obb_size;
} vofod_cluster_info;

/* Optional per-scan debug/parity outputs; every array is caller-allocated. */
typedef struct vofod_scan_debug {
  vofod_point_xyzr* weighted;   /* cloud_weighted of filterAndTransform :659-668 */
  uint32_t* labels;             /* per weighted point: label (= first_member of its cluster) */
  size_t weighted_cap;
  size_t n_weighted;
  vofod_cluster_info* clusters;
  size_t clusters_cap;
  size_t n_clusters;
  uint64_t n_input_after_crop;  /* points entering VoxelGridWeighted (:657) */
  uint64_t n_bg_voxels;         /* nVoxelsOver(new_obstacles) :715 */
  int32_t background_pts_sufficient;
  int32_t sure_background_sufficient;
  /* host wall-clock per stage, ms, reference ScopeTimer names (:924-964):
   * [0] filtering [1] clusterization [2] close X far [3] vmap update [4] classification */
  double stage_ms[8];
  /* INPUT, batches only (read from dbg[0]): non-zero asks for the view of the production path of a read-only batch, which
   * clusters close first - only far clusters are ever used (:727-748, :946-963): `clusters` lists the far clusters only
   * (is_close = 0, canonical order) and `labels` is VOFOD_LABEL_NONE for every voxel outside them.
   * The struct carries inputs (the buffers, their capacities and this field): ZERO-INITIALISE it (`vofod_scan_debug d = {0}`)
   * before filling in what is wanted - garbage here silently switches the view. */
  int32_t far_only;
  int32_t reserved_;
} vofod_scan_debug;
#define VOFOD_LABEL_NONE 0xffffffffu

typedef struct vofod_status_info {
  int32_t detection_its;               /* m_detection_its */
  uint32_t last_detection_id;          /* m_last_detection_id */
  int32_t background_pts_sufficient;   /* m_background_pts_sufficient */
  int32_t sure_background_sufficient;  /* m_sure_background_sufficient */
  int32_t raycast_pending;             /* m_raycast_running */
  int32_t map_size[3];                 /* VoxelMap::sizesIdx */
  float map_offset[3];                 /* VoxelMap::origin */
} vofod_status_info;

enum { VOFOD_MAP_VOXELS = 0, VOFOD_MAP_FLAGS = 1, VOFOD_MAP_RAYCAST = 2 };

/* process_scan flags */
enum {
  VOFOD_SCAN_DEFAULT = 0,
  VOFOD_SCAN_NO_MAP_UPDATE = 1,  /* read-only map: skip updateVMaps/++its and keep exploreToGround's
                                    frontier writes in a per-scan overlay (batched mode) */
  VOFOD_SCAN_AUTO_RAYCAST = 2    /* emulate the detached raycast thread :951-957 deterministically: after ++its EITHER
                                    finish the pending raycast OR (none pending) begin one for this scan - as in the
                                    reference, where a new thread starts only when none runs, a pass covers every other scan */
};

/* --------------------------------------------------------------- lifecycle */

/* onInit parameter block :168-230 + reset() :1610-1632.  Allocates the three
 * voxel maps (m_voxel_map := score_init, m_voxel_flags := 0, m_voxel_raycast := 0). */
int vofod_create(const vofod_static_params* sp, const vofod_dyn_params* dp, vofod_handle** out);
void vofod_destroy(vofod_handle* h);
/* reset() :1610-1632 (also re-zeroes the latches like onInit :283-284 and the detection id :296) */
int vofod_reset(vofod_handle* h);
int vofod_set_dynamic_params(vofod_handle* h, const vofod_dyn_params* dp);
const char* vofod_last_error_string(vofod_handle* h);
int vofod_get_status(vofod_handle* h, vofod_status_info* out);

/* initialize_apriori_map :339-345: every point inside the map limits sets its
 * voxel to +inf; both background latches are set.  xyz interleaved, world frame,
 * already transformed/downsampled by the caller (rows N2 of SURVEY 8f). */
int vofod_load_apriori(vofod_handle* h, const float* xyz, size_t n);

/* The whole of initialize_apriori_map (:214-226, :306-345) from a point-cloud file (row N2 of SURVEY 8f): load_cloud
 * -> rigid transform (apriori_map/tf/{x,y,z,yaw} + sim_correction) -> stock pcl::VoxelGrid centroid filter at the map's
 * voxel size -> vofod_load_apriori.  Init-time host work in the reference and here. */
int vofod_ingest_apriori(vofod_handle* h, const char* filename, const float tf_xyz[3], double yaw_deg, const float sim_correction[3],
                         size_t* n_loaded, size_t* n_voxels);

/* VoxelMap::voxelsAsPC (voxel_map.cpp:157-183): the voxels with ((value > threshold) == greater_than) of map `which` as
 * world-frame centres + value, in the reference's order (x outer, y, z inner).  The nodelet's debug clouds
 * (vofod_nodelet.cpp:999-1013): background = (VOFOD_MAP_VOXELS, new_obstacles, 1), sure air = (VOFOD_MAP_VOXELS, frontiers, 0).
 * VOFOD_ERR_CAPACITY: *n_out holds the required number of points. */
int vofod_voxels_as_pc(vofod_handle* h, int which, float threshold, int greater_than, vofod_point_xyzi* out, size_t cap, size_t* n_out);

/* processMsg(sensor_msgs::Range) :581-613 (row N4 of SURVEY 8f): the height range-finder marks the voxel it hits as
 * background-ish: p = tf * (range, 0, 0); if inLimits(p): map(p) = (map(p) + voxel_map/scores/point) / 2.0.
 * The reference's validity test `range <= min_range && range >= max_range` is kept as written.
 * Returns VOFOD_ERR_MAP_RANGE when the point is outside the operation area (map untouched). */
int vofod_update_ground(vofod_handle* h, float range, float min_range, float max_range, const float tf[12]);

/* test/visualisation access to the three maps (x-fastest, idx = ix + iy*sx + iz*sx*sy: voxel_map.cpp:81) */
int vofod_read_map(vofod_handle* h, int which, float* dst, size_t n);
int vofod_write_map(vofod_handle* h, int which, const float* src, size_t n);

/* ---------------------------------------------------------------- hot path */

/* Body of processMsg(pc_t::ConstPtr,int) between :926 and :965:
 * filterAndTransform :621-684 -> clusterCloud :689-698 -> findCloseFarClusters :703-750
 * -> updateVMaps :943-950 -> classifyClusters :819-830 -> extractDetections :834-879. */
int vofod_process_scan(vofod_handle* h, const vofod_scan* scan, const float tf[12], int flags,
                       vofod_detection* out, size_t cap, size_t* n_out, vofod_scan_debug* dbg);

/* Batched mode (new): n independent scans against the handle's current map,
 * each with VOFOD_SCAN_NO_MAP_UPDATE semantics.  Detections of all frames are appended to
 * `out` in frame order, `n_out_per_frame[f]` counts them.  dbg: NULL or an array of n. */
int vofod_process_batch(vofod_handle* h, const vofod_scan* scans, const float* tfs, size_t n,
                        vofod_detection* out, size_t cap, uint32_t* n_out_per_frame, size_t* n_out,
                        vofod_scan_debug* dbg);

/* The same, pipelined: submit enqueues the kernels of a batch and returns a ticket (0..7; at most eight batches in flight:
 * streaming kernels, frame kernels and classification tails of consecutive batches run as a three-stage pipeline on the
 * device; batches of fewer than 128 frames run side by side on streams of their own, tails included - see INTEGRATION.md on GPU_MAX_HW_QUEUES), collect waits for it and returns the detections.  Submitting batch k+1 (and k+2) before collecting batch k keeps
 * the pipeline full.  Read-only map only (VOFOD_SCAN_NO_MAP_UPDATE semantics); collect in submit order for deterministic
 * detection ids.  The scans' host buffers need not outlive submit.  collect with an `out` too small for the batch returns
 * VOFOD_ERR_CAPACITY with *n_out = the detections to make room for; the ticket then stays pending and no ids are handed out
 * (batches of >= 4 frames; a batch that had to take the host tail is consumed by the failing call). */
int vofod_batch_submit(vofod_handle* h, const vofod_scan* scans, const float* tfs, size_t n, int* ticket);
/* Allocates now what the first `tickets` (1..8) batches in flight would otherwise allocate inside their first
 * vofod_batch_submit (workspaces of max_batch_frames slots, flood-fill buffers of the device tail: hundreds of ms of
 * allocation in a real-time caller's first calls).  Optional; the reference has no counterpart (its buffers are
 * std::vectors grown on use). */
int vofod_reserve(vofod_handle* h, int tickets);
int vofod_batch_collect(vofod_handle* h, int ticket, vofod_detection* out, size_t cap, uint32_t* n_out_per_frame, size_t* n_out);

/* raycast_cloud :1397-1605 split where the reference thread blocks on m_detection_cv (:1530-1537):
 *   begin  = guards :1400-1423, start_detection_its :1425, clear + DDA accumulation :1430-1492
 *   finish = detection_its_diff :1539, max :1542, update sweep :1550-1601, flags clear :1602 */
int vofod_raycast_begin(vofod_handle* h, const vofod_scan* scan, const float tf[12]);
int vofod_raycast_finish(vofod_handle* h);

/* updateSeparatedBGClusters :1126-1277 split at the second lock (:1210):
 *   begin  = snapshot :1146-1150, voxelsAsVoxelPC :1153, VoxelGridCounted :1162-1167,
 *            clusterCloud :1171, sure counts :1175-1183, latch :1188-1206
 *   finish = detection_its_diff :1212, stencil :1219-1237, erase :1239-1272
 * `sure_background_sufficient` (nullable) receives the latch. */
int vofod_sepclusters_begin(vofod_handle* h, int* sure_background_sufficient);
int vofod_sepclusters_finish(vofod_handle* h);

/* --------------------------------------------- stateless L4 entry points */

typedef struct vofod_cloud_view {
  const void* x; const void* y; const void* z;
  const void* intensity;   /* VoxelGridCounted only */
  size_t stride_bytes;
  size_t n;
  int32_t memspace;
} vofod_cloud_view;

/* Lattice of a voxel-grid output: centre = (ijk + 0.5) * leaf + offset,
 * key = i + j*div[0] + k*div[0]*div[1] (voxel_grid_weighted.cpp:109-113,136,178-180) */
typedef struct vofod_grid_desc {
  float leaf[3];
  float offset[3];
  int32_t min_b[3];
  int32_t div_b[3];
} vofod_grid_desc;

/* VoxelGridWeighted::filter (voxel_grid_weighted.cpp:28-190).  align != 0 reproduces
 * setVoxelAlign(align_center) :22-26.  keys (nullable) receives each output voxel's key. */
int vofod_voxel_grid_weighted(vofod_handle* h, const vofod_cloud_view* in, float leaf,
                              int align, const float align_center[3],
                              vofod_point_xyzr* out, uint32_t* keys, size_t cap, size_t* n_out,
                              vofod_grid_desc* grid);

/* VoxelGridCounted::filter (voxel_grid_counted.cpp:36-196), including the positional
 * count range of :185-187. */
int vofod_voxel_grid_counted(vofod_handle* h, const vofod_cloud_view* in, float leaf, float threshold,
                             vofod_point_xyzr* out, uint32_t* keys, size_t cap, size_t* n_out,
                             vofod_grid_desc* grid);

/* clusterCloud (vofod_nodelet.cpp:689-698) on a voxel-grid output.  labels[i] = smallest
 * member index of i's cluster.  keys/grid describe the lattice the points sit on (the HIP
 * implementation clusters on the lattice; the oracle ignores them and uses the coordinates). */
int vofod_cluster(vofod_handle* h, const vofod_point_xyzr* pts, const uint32_t* keys,
                  const vofod_grid_desc* grid, size_t n, float tolerance,
                  uint32_t* labels, size_t* n_clusters);

/* load_cloud (pc_loader.cpp:17-90): whitespace-separated "x y z" text (".pts": count on the
 * first line).  xyz interleaved; returns VOFOD_ERR_CAPACITY with *n_out = required points. */
int vofod_load_cloud(const char* filename, float* xyz, size_t cap, size_t* n_out);

/* simulated sensor LUT of initialize_sensor_lut_simulation (vofod_nodelet.cpp:374-420) */
int vofod_sim_lut(int32_t w, int32_t h, float vfov, float* directions /* 3*w*h */);

/* initialize_sensor_lut (vofod_nodelet.cpp:358-372) from the Ouster metadata (row N1 of SURVEY 8f): [3P] ouster::make_xyz_lut
 * restated, then cast to float and directions normalised.  tf16 = lidar_to_sensor_transform, row-major 4x4, NULL = identity;
 * azimuth / altitude: beam angles in degrees, one per row.  Output layout as vofod_static_params::lut_*. */
int vofod_ouster_lut(int32_t w, int32_t h, double range_unit, double lidar_origin_to_beam_origin_mm, const double* tf16,
                     const double* azimuth_deg, const double* altitude_deg, float* directions /* 3*w*h */, float* offsets /* 3*w*h */);

/* load_mask (:506-560) after the image is decoded: plain copy or "mangling" into the packets' staggered column-major order
 * (:527-541, pixel_shift_by_row from the metadata, NULL = zeros); image NULL = no usable file -> all ones (:558). */
int vofod_mask_layout(const uint8_t* image /* w*h, row-major */, int32_t w, int32_t h, const int32_t* pixel_shift_by_row /* h */,
                      int32_t mangle, uint8_t* mask /* w*h */);

/* check_sensor_params (vofod_nodelet.cpp:1869-1917): the first valid pixel (mask set, range != 0; rows outer) of an organised
 * host-resident cloud against the sensor model: direction of (point - beam offset) vs the LUT direction, its length vs
 * range * 0.001 m, unit length of the LUT direction, each within 1e-3.  *checked (nullable) = a valid pixel was found
 * (m_sensor_params_checked).  VOFOD_OK: the parameters fit or nothing could be checked; VOFOD_ERR_SIZE_MISMATCH: they do not
 * (m_sensor_params_ok = false: the nodelet then refuses to raycast, :1413-1418).  lut_offsets and mask may be NULL. */
int vofod_check_sensor_params(const vofod_scan* scan, const float* lut_directions /* 3*w*h */, const float* lut_offsets /* 3*w*h */, const uint8_t* mask /* w*h */,
                              int32_t* checked);

/* ------------------------------------------------------------ diagnostics */

/* Per-kernel device time, measured with HIP events on the handle's own stream (the reference analogue is
 * mrs_lib::ScopeTimer, vofod_nodelet.cpp:887,1135,1426).  While enabled every kernel launch is bracketed by
 * two events; vofod_profile_read drains them: names[64*i..] = kernel name, ms[i] = summed time, calls[i] =
 * launches.  Returns the number of distinct kernels.  No-ops in the oracle. */
int vofod_profile_enable(vofod_handle* h, int on);
size_t vofod_profile_read(vofod_handle* h, char* names, double* ms, uint64_t* calls, size_t cap);

/* ------------------------------------------------- outgoing messages (product library only)
 *
 * The nodelet's publications in the ROS 1 wire format (little endian; strings / arrays carry a uint32 length; a
 * std_msgs/Header is seq, stamp.sec, stamp.nsec, frame_id), for hosts without ROS: vofod/Detections
 * (msgs/Detections.msg + Detection.msg:1-12, vofod_nodelet.cpp:968-988), vofod/Status (msgs/Status.msg, :1379-1385),
 * vofod/ProfilingInfo (msgs/ProfilingInfo.msg, :2178-2201; event_type 1 = start, 2 = end).  buf may be NULL to ask for the
 * size; VOFOD_ERR_CAPACITY when cap is too small (*n_bytes = bytes needed). */
typedef struct vofod_msg_header {
  uint32_t seq;
  uint32_t stamp_sec, stamp_nsec;
  const char* frame_id; /* world frame (m_world_frame_id) */
} vofod_msg_header;
int vofod_serialize_detections(const vofod_msg_header* header, const vofod_detection* dets, size_t n, uint8_t* buf, size_t cap, size_t* n_bytes);
int vofod_serialize_status(const vofod_msg_header* header, int detection_enabled, int detection_active, uint8_t* buf, size_t cap, size_t* n_bytes);
int vofod_serialize_profiling_info(uint32_t stamp_sec, uint32_t stamp_nsec, uint32_t routine_id, uint64_t event_sequence, uint8_t event_type, uint8_t* buf, size_t cap,
                                   size_t* n_bytes);

/* ------------------------------------------------- batched mode: the collective (product library only)
 *
 * one process per GPU runs vofod_process_batch / vofod_batch_submit+collect on its own block of frames; the only
 * exchange is one all-gather of fixed-size slots - d_max 128-byte vofod_detection records (msgs/Detection.msg:1-12 + frame)
 * and a count word per frame - with RCCL over xGMI.  The communicator is RCCL's: one rank asks for an id
 * (ncclGetUniqueId), the host program ships its 128 bytes to the other ranks by whatever means it has (MPI, a socket,
 * torch.distributed), every rank creates its vofod_comm from it (ncclCommInitRank).  RCCL is loaded at run time.
 * The CPU oracle does not export these (it has no device to gather on). */
#define VOFOD_COMM_ID_BYTES 128
typedef struct vofod_comm vofod_comm;
int vofod_comm_unique_id(uint8_t id[VOFOD_COMM_ID_BYTES]);
int vofod_comm_create(const uint8_t id[VOFOD_COMM_ID_BYTES], int32_t rank, int32_t n_ranks, int32_t device, vofod_comm** out);
void vofod_comm_destroy(vofod_comm* comm);
const char* vofod_comm_last_error(vofod_comm* comm);
/* The exchange's wire format as plain host functions (no device, no communicator): a caller with a transport of its own (MPI,
 * gloo) packs, all-gathers `frames * vofod_detection_slot_bytes(d_max)` bytes per rank and unpacks.  Slot of a frame: d_max
 * records of 128 bytes (the frame's first detections in order, zero padded) followed by the frame's true count (8 bytes). */
size_t vofod_detection_slot_bytes(size_t d_max);
int vofod_pack_detection_slots(const vofod_detection* local, const uint32_t* n_per_frame, size_t frames, size_t d_max, void* slots);
int vofod_unpack_detection_slots(const void* slots, size_t frames_total, size_t d_max, vofod_detection* all, uint32_t* all_counts);
/* local: this rank's detections in frame order (what vofod_process_batch / vofod_batch_collect returned), n_per_frame: their
 * count per frame.  all: n_ranks * frames_per_rank * d_max records, slot (rank, frame) holds min(count, d_max) records;
 * all_counts: n_ranks * frames_per_rank counts (a count above d_max tells that the slot was truncated). */
int vofod_allgather_detections(vofod_comm* comm, const vofod_detection* local, const uint32_t* n_per_frame, size_t frames_per_rank, size_t d_max, vofod_detection* all,
                               uint32_t* all_counts);

#ifdef __cplusplus
}
#endif
#endif /* VOFOD_H */
