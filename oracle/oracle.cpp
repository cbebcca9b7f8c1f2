// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of VoFOD's per-scan hot path
// behind the same C-ABI as the product (include/vofod.h), exported as vofod_oracle_*.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (vofod_amd/) never does.
//
// PARITY UNPINNED: the reference has no tests or golden vectors and its
// sources need PCL/Eigen/ROS, none of which exist in this image, so it can be neither
// built (oracle/_ref) nor run.  The restatement is pinned by the hand-derived
// known-answer tests of tests/test_oracle_kat.py only.
//
// Reference lines followed are cited per function (paths relative to the reference tree).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>

#include "algorithms.hpp"
#include "voxel_map.hpp"

#define ORACLE_API(name) vofod_oracle_##name

namespace
{

using clk = std::chrono::steady_clock;
inline double ms_since(const clk::time_point& t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

inline float ld_f32(const void* base, size_t stride, size_t i) { float v; std::memcpy(&v, static_cast<const char*>(base) + i * stride, 4); return v; }
inline uint32_t ld_u32(const void* base, size_t stride, size_t i) { uint32_t v; std::memcpy(&v, static_cast<const char*>(base) + i * stride, 4); return v; }

}  // namespace

struct vofod_handle
{
  std::mutex mtx;
  vofod_static_params sp{};
  vofod_dyn_params dp{};
  std::string err;

  // derived in onInit (vofod_nodelet.cpp:198-212, 228-230)
  float exclude_center[3], oparea_center[3];
  uint64_t background_min_sufficient_pts = 0;

  std::vector<float> lut_dirs, lut_offs;
  std::vector<uint8_t> mask;

  vo::VoxelMap vmap, vflags, vraycast;
  bool background_pts_sufficient = false, sure_background_sufficient = false;
  int detection_its = 0;
  uint32_t last_detection_id = 0;

  // raycast_cloud split state
  bool raycast_pending = false;
  int raycast_start_its = 0;

  // updateSeparatedBGClusters split state
  bool sep_pending = false;
  int sep_start_its = 0;
  std::vector<vofod_point_xyzr> sep_ds;
  std::vector<vo::Cluster> sep_clusters;
  std::vector<size_t> sep_n_sure;

  // vofod_batch_submit / vofod_batch_collect: the oracle simply computes at submit time
  struct Ticket
  {
    bool pending = false;
    std::vector<vofod_detection> dets;
    std::vector<uint32_t> per_frame;
    int status = VOFOD_OK;
  } tickets[4];
};

namespace
{

constexpr float VFLAGS_UNMARKED = 0.0f, VFLAGS_POINT = 2.0f, VFLAGS_UNKNOWN = 3.0f;  // vofod_nodelet.cpp:2335-2337

// reset() vofod_nodelet.cpp:1610-1632
void do_reset(vofod_handle* h)
{
  const vofod_static_params& sp = h->sp;
  h->vmap.resize_center(h->oparea_center, sp.oparea_size, sp.voxel_size);
  h->vmap.setTo(sp.score_init);
  const float* o = h->vmap.off;
  const int s[3] = {h->vmap.sx, h->vmap.sy, h->vmap.sz};
  h->vflags.resize(o, s, sp.voxel_size);
  h->vflags.setTo(0);
  h->vraycast.resize(o, s, sp.voxel_size);
  h->vraycast.setTo(0);
  h->detection_its = 0;
  h->raycast_pending = false;
  h->sep_pending = false;
}

// updateVoxel vofod_nodelet.cpp:777-797
bool update_voxel(vofod_handle* h, const vofod_point_xyzr& pt, const float vmap_score, const float vflags)
{
  const auto c = h->vmap.coordToIdx(pt.x, pt.y, pt.z);
  if (!h->vmap.inLimitsIdx(c[0], c[1], c[2]))
    return false;  // the reference's vector::at would throw / hit a wrong cell
  float& mapval = h->vmap.at(c[0], c[1], c[2]);
  const float w = 1.0f / static_cast<float>(1lu << std::clamp(pt.range, 0u, 63u));
  mapval = w * mapval + (1.0f - w) * vmap_score;
  h->vflags.at(c[0], c[1], c[2]) = vflags;
  return true;
}

struct ClassifiedCluster
{
  int cclass = VOFOD_CLASS_INVALID;
  vo::Boxes boxes;
  float obb_size = std::numeric_limits<float>::quiet_NaN();
};

// classify_cluster vofod_nodelet.cpp:1648-1730.  undo (nullable) records map writes so that a
// VOFOD_SCAN_NO_MAP_UPDATE scan can restore them.
ClassifiedCluster classify_cluster(vofod_handle* h, const std::vector<vofod_point_xyzr>& cloud, const vo::Cluster& cl, const float tf[12],
                                   std::vector<std::pair<size_t, float>>* undo)
{
  ClassifiedCluster ret;
  ret.boxes = vo::moie(cloud, cl.indices);
  const vofod_dyn_params& dp = h->dp;
  if (static_cast<int>(cl.indices.size()) < dp.classification__min_points)
    return ret;
  const float t[3] = {tf[3], tf[7], tf[11]};
  {
    const float d[3] = {t[0] - ret.boxes.obb_center[0], t[1] - ret.boxes.obb_center[1], t[2] - ret.boxes.obb_center[2]};
    const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);  // float norm(), widened
    if (dist > dp.classification__max_distance)
      return ret;
  }
  {
    const float d[3] = {ret.boxes.obb_max[0] - ret.boxes.obb_min[0], ret.boxes.obb_max[1] - ret.boxes.obb_min[1], ret.boxes.obb_max[2] - ret.boxes.obb_min[2]};
    ret.obb_size = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (ret.obb_size > dp.classification__max_size)
      return ret;
  }
  bool is_floating = true;
  if (h->background_pts_sufficient && h->sure_background_sufficient)
  {
    const int max_explore_voxel_size = static_cast<int>((ret.obb_size + dp.classification__max_explore_distance) / h->sp.voxel_size);
    const float thr_frontiers = static_cast<float>(dp.voxel_map__thresholds__frontiers);
    const float thr_new_obstacles = static_cast<float>(dp.voxel_map__thresholds__new_obstacles);
    for (const int idx : cl.indices)
    {
      const auto& pt = cloud[idx];
      const auto [is_connected, explored] = h->vmap.exploreToGround(pt.x, pt.y, pt.z, thr_frontiers, thr_new_obstacles, static_cast<float>(max_explore_voxel_size));
      if (is_connected)
      {
        is_floating = false;
        break;
      }
      for (const auto& e : explored)  // :1712-1715, classification mutates the map
      {
        const size_t li = h->vmap.lin(e[0], e[1], e[2]);
        if (undo)
          undo->emplace_back(li, h->vmap.data[li]);
        h->vmap.data[li] = thr_frontiers;
      }
    }
  }
  else
    is_floating = false;
  ret.cclass = is_floating ? VOFOD_CLASS_MAV : VOFOD_CLASS_UNKNOWN;
  return ret;
}

// extractDetections vofod_nodelet.cpp:834-879 for one mav cluster
vofod_detection make_detection(vofod_handle* h, const std::vector<vofod_point_xyzr>& cloud, const vo::Cluster& cl, const ClassifiedCluster& cc,
                               const float tf[12])
{
  const vofod_dyn_params& dp = h->dp;
  vofod_detection det{};
  const float d[3] = {tf[3] - cc.boxes.obb_center[0], tf[7] - cc.boxes.obb_center[1], tf[11] - cc.boxes.obb_center[2]};
  const double det_dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  det.id = h->last_detection_id++;
  det.n_points = cl.indices.size();
  const float cov = static_cast<float>(std::sqrt(det_dist) * dp.output__position_sigma);  // double scalar folded into a Matrix3f
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      det.covariance[3 * r + c] = r == c ? cov : 0.0f;

  vo::VoxelMap submap = h->vmap.getSubmapCopy(cc.boxes.aabb_min, cc.boxes.aabb_max, 2);
  const float ray = static_cast<float>(dp.voxel_map__scores__ray);
  for (const int idx : cl.indices)
  {
    const auto& pt = cloud[idx];
    const auto c = submap.coordToIdx(pt.x, pt.y, pt.z);
    submap.at(c[0], c[1], c[2]) = ray;
  }
  double uncertainty = 0.0;
  for (const float val : submap.data)
    uncertainty += 1.0 - val / dp.voxel_map__scores__ray;
  uncertainty /= cl.indices.size();
  det.confidence = static_cast<float>(1.0 / std::exp(uncertainty));

  const double vray_res = h->sp.sensor_vfov / static_cast<double>(h->sp.sensor_vrays);
  const double hray_res = 2 * M_PI / static_cast<double>(h->sp.sensor_hrays);
  const double pdet_vert = std::min(std::atan(1.0 / det_dist) / (vray_res * dp.classification__min_points), 1.0);
  const double pdet_hori = std::min(std::atan(1.0 / det_dist) / (hray_res), 1.0);
  det.detection_probability = pdet_vert * pdet_hori;
  for (int a = 0; a < 3; a++)
    det.position[a] = cc.boxes.obb_center[a];
  return det;
}

int raycast_begin_locked(vofod_handle* h, const vofod_scan* scan, const float tf[12]);
int raycast_finish_locked(vofod_handle* h);

// processMsg body vofod_nodelet.cpp:926-965
int process_scan_locked(vofod_handle* h, const vofod_scan* scan, const float tf[12], int flags, uint32_t frame, vofod_detection* out, size_t cap,
                        size_t* n_out, vofod_scan_debug* dbg)
{
  if (!scan || !tf || !n_out || !scan->x || !scan->y || !scan->z)
    return VOFOD_ERR_INVALID_ARG;
  *n_out = 0;
  if (scan->memspace != VOFOD_MEM_HOST)
    return VOFOD_ERR_INVALID_ARG;
  const size_t n = static_cast<size_t>(scan->width) * scan->height;
  if (n != static_cast<size_t>(h->sp.sensor_hrays) * h->sp.sensor_vrays)  // :895-899
    return VOFOD_ERR_SIZE_MISMATCH;
  const vofod_static_params& sp = h->sp;
  const vofod_dyn_params& dp = h->dp;
  const bool no_update = flags & VOFOD_SCAN_NO_MAP_UPDATE;
  int ret = VOFOD_OK;

  // ---- filterAndTransform :621-684
  auto t0 = clk::now();
  float ex_min[3], ex_max[3], op_min[3], op_max[3];
  for (int a = 0; a < 3; a++)
  {
    ex_max[a] = h->exclude_center[a] + sp.exclude_size[a] / 2;
    ex_min[a] = h->exclude_center[a] - sp.exclude_size[a] / 2;
    op_max[a] = h->oparea_center[a] + sp.oparea_size[a] / 2;
    op_min[a] = h->oparea_center[a] - sp.oparea_size[a] / 2;
  }
  vo::Cloud filtered;
  for (size_t i = 0; i < n; i++)
  {
    const float p[3] = {ld_f32(scan->x, scan->stride_bytes, i), ld_f32(scan->y, scan->stride_bytes, i), ld_f32(scan->z, scan->stride_bytes, i)};
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]))
      continue;
    if (vo::crop_inside(p, ex_min, ex_max))  // negative crop :625-636
      continue;
    float q[3];
    vo::transform_point(tf, p, q);  // :640
    if (!vo::crop_inside(q, op_min, op_max))  // :644-655
      continue;
    filtered.x.push_back(q[0]);
    filtered.y.push_back(q[1]);
    filtered.z.push_back(q[2]);
  }
  const auto ac = h->vmap.idxToCoord(0, 0, 0);  // :664
  const float align_center[3] = {ac[0], ac[1], ac[2]};
  const float leaf[3] = {sp.voxel_size, sp.voxel_size, sp.voxel_size};
  vo::GridOut g = vo::voxel_grid(filtered, leaf, true, align_center, false, 0.0f);
  if (g.status != VOFOD_OK)
    ret = g.status;  // reference logs and carries on with an empty cloud
  const std::vector<vofod_point_xyzr>& cloud = g.pts;
  if (dbg)
    dbg->stage_ms[0] = ms_since(t0);

  // ---- clusterCloud :932
  t0 = clk::now();
  const std::vector<uint32_t> labels = vo::euclidean_labels(cloud, static_cast<float>(dp.ground_points_max_distance));
  const std::vector<vo::Cluster> clusters = vo::clusters_from_labels(labels);
  if (dbg)
    dbg->stage_ms[1] = ms_since(t0);

  // ---- findCloseFarClusters :703-750
  t0 = clk::now();
  const float max_dist = static_cast<float>(dp.ground_points_max_distance);
  const float thr_new_obstacles = static_cast<float>(dp.voxel_map__thresholds__new_obstacles);
  const uint64_t n_bg_pts = h->vmap.nVoxelsOver(thr_new_obstacles);
  if (n_bg_pts > h->background_min_sufficient_pts)
    h->background_pts_sufficient = true;
  std::vector<char> is_close(clusters.size(), 0);
  for (size_t c = 0; c < clusters.size(); c++)
    for (const int idx : clusters[c].indices)
    {
      const auto& pt = cloud[idx];
      if (h->vmap.hasCloseTo(pt.x, pt.y, pt.z, max_dist, thr_new_obstacles))
      {
        is_close[c] = 1;
        break;
      }
    }
  if (dbg)
    dbg->stage_ms[2] = ms_since(t0);

  // ---- updateVMaps :943-950
  t0 = clk::now();
  if (!no_update)
  {
    for (int pass = 0; pass < 2; pass++)  // close clusters first, then far
      for (size_t c = 0; c < clusters.size(); c++)
      {
        if ((pass == 0) != (is_close[c] != 0))
          continue;
        const float score = static_cast<float>(pass == 0 ? dp.voxel_map__scores__point : dp.voxel_map__scores__unknown);
        const float flag = pass == 0 ? VFLAGS_POINT : VFLAGS_UNKNOWN;
        for (const int idx : clusters[c].indices)
          if (!update_voxel(h, cloud[idx], score, flag))
            ret = VOFOD_ERR_MAP_RANGE;
      }
    h->detection_its++;
    if (flags & VOFOD_SCAN_AUTO_RAYCAST)  // deterministic stand-in for :951-957
    {
      if (h->raycast_pending)
        raycast_finish_locked(h);
      else
        raycast_begin_locked(h, scan, tf);
    }
  }
  if (dbg)
    dbg->stage_ms[3] = ms_since(t0);

  // ---- classifyClusters :961 + extractDetections :963
  t0 = clk::now();
  std::vector<std::pair<size_t, float>> undo;
  std::vector<ClassifiedCluster> classified(clusters.size());
  size_t n_det = 0;
  for (size_t c = 0; c < clusters.size(); c++)
  {
    if (is_close[c])
      continue;
    classified[c] = classify_cluster(h, cloud, clusters[c], tf, no_update ? &undo : nullptr);
  }
  for (size_t c = 0; c < clusters.size(); c++)
  {
    if (is_close[c] || classified[c].cclass != VOFOD_CLASS_MAV)
      continue;
    vofod_detection det = make_detection(h, cloud, clusters[c], classified[c], tf);
    det.frame = frame;
    if (n_det < cap && out)
      out[n_det] = det;
    n_det++;
  }
  for (auto it = undo.rbegin(); it != undo.rend(); ++it)
    h->vmap.data[it->first] = it->second;
  *n_out = n_det;
  if (n_det > cap)
    ret = VOFOD_ERR_CAPACITY;
  if (dbg)
    dbg->stage_ms[4] = ms_since(t0);

  if (dbg)
  {
    dbg->n_input_after_crop = filtered.size();
    dbg->n_bg_voxels = n_bg_pts;
    dbg->background_pts_sufficient = h->background_pts_sufficient;
    dbg->sure_background_sufficient = h->sure_background_sufficient;
    dbg->n_weighted = cloud.size();
    if (dbg->weighted && dbg->weighted_cap >= cloud.size())
      std::copy(cloud.begin(), cloud.end(), dbg->weighted);
    if (dbg->labels && dbg->weighted_cap >= cloud.size())
      std::copy(labels.begin(), labels.end(), dbg->labels);
    if ((dbg->weighted || dbg->labels) && dbg->weighted_cap < cloud.size())
      ret = VOFOD_ERR_CAPACITY;
    dbg->n_clusters = clusters.size();
    if (dbg->clusters)
    {
      if (dbg->clusters_cap < clusters.size())
        ret = VOFOD_ERR_CAPACITY;
      else
        for (size_t c = 0; c < clusters.size(); c++)
        {
          vofod_cluster_info& ci = dbg->clusters[c];
          ci.first_member = clusters[c].indices.front();
          ci.n_points = clusters[c].indices.size();
          ci.is_close = is_close[c];
          const float nanv = std::numeric_limits<float>::quiet_NaN();
          if (is_close[c])
          {
            ci.cclass = VOFOD_CLASS_NONE;
            // AABB is still well defined; evaluate it for parity of the cluster table
            for (int a = 0; a < 3; a++)
            {
              ci.aabb_min[a] = FLT_MAX;
              ci.aabb_max[a] = -FLT_MAX;
            }
            for (const int idx : clusters[c].indices)
            {
              const float p[3] = {cloud[idx].x, cloud[idx].y, cloud[idx].z};
              for (int a = 0; a < 3; a++)
              {
                ci.aabb_min[a] = std::min(ci.aabb_min[a], p[a]);
                ci.aabb_max[a] = std::max(ci.aabb_max[a], p[a]);
              }
            }
            for (int a = 0; a < 3; a++)
              ci.obb_center[a] = nanv;
            ci.obb_size = nanv;
          }
          else
          {
            ci.cclass = classified[c].cclass;
            for (int a = 0; a < 3; a++)
            {
              ci.aabb_min[a] = classified[c].boxes.aabb_min[a];
              ci.aabb_max[a] = classified[c].boxes.aabb_max[a];
              ci.obb_center[a] = classified[c].boxes.obb_center[a];
            }
            ci.obb_size = classified[c].obb_size;
          }
        }
    }
  }
  return ret;
}

// raycast_cloud vofod_nodelet.cpp:1397-1492 (guards, clear, DDA accumulation)
int raycast_begin_locked(vofod_handle* h, const vofod_scan* scan, const float tf[12])
{
  const vofod_dyn_params& dp = h->dp;
  if (h->raycast_pending)
    return VOFOD_ERR_INVALID_ARG;  // at most one in flight (:952-957)
  if (dp.raycast__pause)
    return VOFOD_ERR_PAUSED;
  if (!scan || !scan->intensity || !scan->range || scan->memspace != VOFOD_MEM_HOST)
    return VOFOD_ERR_INVALID_ARG;
  if (scan->height != h->sp.sensor_vrays || scan->width != h->sp.sensor_hrays)  // :1407-1411
    return VOFOD_ERR_SIZE_MISMATCH;

  h->raycast_start_its = h->detection_its;  // :1425
  h->raycast_pending = true;
  // [3P] tf.rotation() on an Affine3f re-orthonormalises through a JacobiSVD; for the rigid
  // transforms tf2 delivers it equals the linear part up to rounding, which is what is used here.
  const float R[9] = {tf[0], tf[1], tf[2], tf[4], tf[5], tf[6], tf[8], tf[9], tf[10]};
  const float origin[3] = {tf[3], tf[7], tf[11]};
  h->vraycast.setTo(0);  // :1430
  int ret = VOFOD_OK;
  if (h->vraycast.inLimits(origin[0], origin[1], origin[2]))  // :1432
  {
    const float max_dist = static_cast<float>(dp.raycast__max_distance);
    const float min_intensity = static_cast<float>(dp.raycast__min_intensity);
    const int W = scan->width, H = scan->height;
    for (int row = 0; row < H; row++)
      for (int col = 0; col < W; col++)
      {
        const size_t idx = static_cast<size_t>(row) * W + col;
        const float intensity = ld_f32(scan->intensity, scan->stride_bytes, idx);
        const uint32_t range = ld_u32(scan->range, scan->stride_bytes, idx);
        if (intensity < min_intensity || (!h->mask[idx] && range == 0))  // :1449
          continue;
        const float* d1 = &h->lut_dirs[3 * idx];
        const float* o1 = &h->lut_offs[3 * idx];
        float dir[3], start[3];
        for (int r = 0; r < 3; r++)
        {
          dir[r] = (R[3 * r] * d1[0] + R[3 * r + 1] * d1[1]) + R[3 * r + 2] * d1[2];            // :1453
          start[r] = ((R[3 * r] * o1[0] + R[3 * r + 1] * o1[1]) + R[3 * r + 2] * o1[2]) + origin[r];  // :1477
        }
        const float ray_dist = 0.001f * static_cast<float>(range);  // :1455-1456
        const float dist = ray_dist == 0.0f ? max_dist : std::min(ray_dist - h->sp.voxel_size, max_dist);  // :1457
        if (h->vraycast.inLimits(start[0], start[1], start[2]))  // :1482
          h->vraycast.forEachRay(start, dir, dist, [h](float val, int x, int y, int z) { h->vraycast.at(x, y, z) += val; });  // :1484-1489
      }
  }
  else
    ret = VOFOD_ERR_SENSOR_OUTSIDE_MAP;  // :1523-1526; the thread still goes on to wait + update
  return ret;
}

// raycast_cloud vofod_nodelet.cpp:1529-1605 (after the wait on m_detection_cv)
int raycast_finish_locked(vofod_handle* h)
{
  if (!h->raycast_pending)
    return VOFOD_ERR_NOT_PENDING;
  h->raycast_pending = false;  // AtomicScopeFlag :1399
  const vofod_dyn_params& dp = h->dp;
  if (h->detection_its == h->raycast_start_its)  // :1531-1537 (the 0.8 s timeout)
    return VOFOD_ERR_RAYCAST_NO_DETECTION;
  const float detection_its_diff = static_cast<float>(h->detection_its - h->raycast_start_its);  // :1539
  const float max_val = *std::max_element(h->vraycast.data.begin(), h->vraycast.data.end());    // :1542
  if (max_val == 0.0f)
    return VOFOD_ERR_RAYCAST_EMPTY;  // :1544-1548 (flags are *not* cleared on this path)
  const float ray_update_score = static_cast<float>(dp.voxel_map__scores__ray);
  const float ray_update_weight = static_cast<float>(dp.raycast__weight_coefficient);
  const size_t M = h->vmap.size();
  if (dp.raycast__new_update_rule)  // :1550-1573
  {
    const float voxel_diag = static_cast<float>(std::sqrt(3) * h->sp.voxel_size);
    const float weighting_factor = ray_update_weight / voxel_diag;
    for (size_t i = 0; i < M; i++)
    {
      float raycastval;
      if (h->vflags.data[i] == VFLAGS_UNMARKED && (raycastval = h->vraycast.data[i]) > 0.0f)
      {
        float& mapval = h->vmap.data[i];
        const float n_int = weighting_factor * raycastval;
        const float w1 = static_cast<float>(std::pow(2, -detection_its_diff * n_int));
        const float w2 = 1.0f - w1;
        mapval = w1 * mapval + w2 * ray_update_score;
      }
    }
  }
  else  // :1574-1601
  {
    for (size_t i = 0; i < M; i++)
    {
      float raycastval;
      if (h->vflags.data[i] == VFLAGS_UNMARKED && (raycastval = h->vraycast.data[i]) > 0.0f)
      {
        float& mapval = h->vmap.data[i];
        const float norm_val = raycastval / max_val;
        const float w_update_single = ray_update_weight * std::sqrt(norm_val);
        const float w1 = std::clamp(std::pow(1.0f - w_update_single, detection_its_diff), 0.0f, 1.0f);
        const float w2 = 1.0f - w1;
        mapval = w1 * mapval + w2 * ray_update_score;
      }
    }
  }
  h->vflags.setTo(0);  // :1602
  return VOFOD_OK;
}

// updateSeparatedBGClusters vofod_nodelet.cpp:1126-1207
int sepclusters_begin_locked(vofod_handle* h, int* sure_out)
{
  const vofod_dyn_params& dp = h->dp;
  if (sure_out)
    *sure_out = h->sure_background_sufficient;
  if (dp.sepclusters__pause)
    return VOFOD_ERR_PAUSED;
  h->sep_pending = false;
  h->sep_start_its = h->detection_its;  // :1134
  const double max_dist = dp.sepclusters__max_bg_distance;
  const float thr_new = static_cast<float>(dp.voxel_map__thresholds__new_obstacles);
  const float thr_sure = static_cast<float>(dp.voxel_map__thresholds__sure_obstacles);
  const unsigned n_pts_sure_cluster = dp.sepclusters__min_sure_points;
  const float max_dist_idx = static_cast<float>(max_dist / h->sp.voxel_size);  // :1142
  const int max_voxel_dist = static_cast<int>(std::ceil(max_dist_idx));      // :1143

  vo::Cloud raw;  // local_vmap snapshot + voxelsAsVoxelPC :1146-1153
  h->vmap.voxelsAsVoxelPC(thr_new, raw.x, raw.y, raw.z, raw.intensity);
  if (raw.size() == 0)
    return VOFOD_ERR_EMPTY;  // :1155-1159

  const float lsz = static_cast<float>(std::max(max_voxel_dist - 1, 0));  // :1162
  if (!(lsz > 0.0f))
    return VOFOD_ERR_INVALID_ARG;  // leaf 0 -> inverse leaf inf in the reference (undefined output)
  const float leaf[3] = {lsz, lsz, lsz};
  const float zero[3] = {0, 0, 0};
  vo::GridOut g = vo::voxel_grid(raw, leaf, false, zero, true, thr_sure);  // :1163-1167
  if (g.status != VOFOD_OK)
    return g.status;
  h->sep_ds = g.pts;
  const std::vector<uint32_t> labels = vo::euclidean_labels(h->sep_ds, static_cast<float>(max_voxel_dist));  // :1171
  h->sep_clusters = vo::clusters_from_labels(labels);
  h->sep_n_sure.clear();
  for (const auto& cl : h->sep_clusters)  // :1175-1183 (accumulator is an int)
  {
    int acc = 0;
    for (const int idx : cl.indices)
      acc = static_cast<int>(acc + h->sep_ds[idx].range);
    h->sep_n_sure.push_back(static_cast<size_t>(acc));
  }
  const size_t n_sure_clusters = std::count_if(h->sep_n_sure.begin(), h->sep_n_sure.end(), [n_pts_sure_cluster](const size_t a) { return a >= n_pts_sure_cluster; });
  if (n_sure_clusters == 0)  // :1192-1199
  {
    h->sure_background_sufficient = false;
    if (sure_out)
      *sure_out = 0;
    return VOFOD_OK;
  }
  h->sure_background_sufficient = true;  // :1205
  if (sure_out)
    *sure_out = 1;
  h->sep_pending = true;
  return VOFOD_OK;
}

// updateSeparatedBGClusters vofod_nodelet.cpp:1209-1272
int sepclusters_finish_locked(vofod_handle* h)
{
  if (!h->sep_pending)
    return VOFOD_ERR_NOT_PENDING;
  h->sep_pending = false;
  const vofod_dyn_params& dp = h->dp;
  const unsigned n_pts_sure_cluster = dp.sepclusters__min_sure_points;
  const float max_dist_idx = static_cast<float>(dp.sepclusters__max_bg_distance / h->sp.voxel_size);
  const int max_voxel_dist = static_cast<int>(std::ceil(max_dist_idx));
  const float detection_its_diff = static_cast<float>(std::max(h->detection_its - h->sep_start_its, 1));  // :1212

  std::vector<std::array<int, 3>> index_offsets;  // :1219-1237
  for (int x = -max_voxel_dist; x <= max_voxel_dist; x++)
    for (int y = -max_voxel_dist; y <= max_voxel_dist; y++)
      for (int z = -max_voxel_dist; z <= max_voxel_dist; z++)
      {
        const int norm = static_cast<int>(std::sqrt(static_cast<double>(x * x + y * y + z * z)));
        if (static_cast<float>(norm) <= max_dist_idx)
          index_offsets.push_back({x, y, z});
      }
  const float update_val = static_cast<float>(dp.voxel_map__scores__ray);
  const float w_update_single = 0.5f;
  const float w1 = std::clamp(std::pow(1.0f - w_update_single, detection_its_diff), 0.0f, 1.0f);
  const float w2 = 1.0f - w1;
  for (size_t it = 0; it < h->sep_clusters.size(); it++)  // :1244-1272
  {
    const unsigned cur_n_sure = static_cast<unsigned>(h->sep_n_sure[it]);
    if (cur_n_sure >= n_pts_sure_cluster)
      continue;
    for (const int idx : h->sep_clusters[it].indices)
    {
      const auto& p = h->sep_ds[idx];
      const int pos[3] = {static_cast<int>(p.x), static_cast<int>(p.y), static_cast<int>(p.z)};  // cast<int>() truncates :1252
      for (const auto& o : index_offsets)
      {
        const int x = pos[0] + o[0], y = pos[1] + o[1], z = pos[2] + o[2];
        if (!h->vmap.inLimitsIdx(x, y, z))
          continue;
        float& mapval = h->vmap.at(x, y, z);
        mapval = w1 * mapval + w2 * update_val;
      }
    }
  }
  return VOFOD_OK;
}

vo::Cloud cloud_from_view(const vofod_cloud_view* in, bool want_intensity)
{
  vo::Cloud c;
  c.x.resize(in->n);
  c.y.resize(in->n);
  c.z.resize(in->n);
  if (want_intensity)
    c.intensity.resize(in->n);
  for (size_t i = 0; i < in->n; i++)
  {
    c.x[i] = ld_f32(in->x, in->stride_bytes, i);
    c.y[i] = ld_f32(in->y, in->stride_bytes, i);
    c.z[i] = ld_f32(in->z, in->stride_bytes, i);
    if (want_intensity)
      c.intensity[i] = ld_f32(in->intensity, in->stride_bytes, i);
  }
  return c;
}

int emit_grid(const vo::GridOut& g, vofod_point_xyzr* out, uint32_t* keys, size_t cap, size_t* n_out, vofod_grid_desc* grid)
{
  if (grid)
    *grid = g.grid;
  if (n_out)
    *n_out = g.pts.size();
  if (g.status != VOFOD_OK)
    return g.status;
  if (g.pts.size() > cap)
    return VOFOD_ERR_CAPACITY;
  if (out)
    std::copy(g.pts.begin(), g.pts.end(), out);
  if (keys)
    std::copy(g.keys.begin(), g.keys.end(), keys);
  return VOFOD_OK;
}

}  // namespace

extern "C" {

void ORACLE_API(default_params)(vofod_static_params* sp, vofod_dyn_params* dp)
{
  if (sp)
  {
    *sp = vofod_static_params{};
    sp->voxel_size = 0.5f;                              // detection_params.yaml:17
    sp->score_init = -740.0f;                           // :21
    sp->background_sufficient_points_ratio = 0.15f;     // :9
    const float oo[3] = {40.0f, 20.0f, -1.25f}, os[3] = {120.0f, 100.0f, 25.0f};  // sim.yaml:8-15
    const float eo[3] = {0.09f, 0.0f, -0.75f}, es[3] = {2.5f, 2.5f, 1.6f};         // detection_params.yaml:76-83
    for (int a = 0; a < 3; a++)
    {
      sp->oparea_offset[a] = oo[a];
      sp->oparea_size[a] = os[a];
      sp->exclude_offset[a] = eo[a];
      sp->exclude_size[a] = es[a];
    }
    sp->sensor_hrays = 1024;  // sensors/os1-128.yaml:3-5
    sp->sensor_vrays = 128;
    sp->sensor_vfov = static_cast<float>(45.0 / 180.0 * M_PI);
    sp->max_batch_frames = 1;
  }
  if (dp)
  {
    *dp = vofod_dyn_params{};
    dp->ground_points_max_distance = 1.5;
    dp->output__position_sigma = 0.1;
    dp->voxel_map__scores__point = 0.0;
    dp->voxel_map__scores__unknown = -740.0;
    dp->voxel_map__scores__ray = -1000.0;
    dp->voxel_map__thresholds__apriori_map = 0.0;
    dp->voxel_map__thresholds__new_obstacles = -300.0;
    dp->voxel_map__thresholds__sure_obstacles = -0.1;
    dp->voxel_map__thresholds__frontiers = -750.0;
    dp->classification__min_points = 2;
    dp->classification__max_size = 3.0;
    dp->classification__max_distance = 50.0;
    dp->classification__max_explore_distance = 3.0;
    dp->raycast__pause = 0;
    dp->raycast__new_update_rule = 1;
    dp->raycast__max_distance = 20.0;
    dp->raycast__min_intensity = 0.0;
    dp->raycast__weight_coefficient = 0.003;
    dp->sepclusters__pause = 0;
    dp->sepclusters__max_bg_distance = 0.8;
    dp->sepclusters__min_sure_points = 24;
  }
}

int ORACLE_API(create)(const vofod_static_params* sp, const vofod_dyn_params* dp, vofod_handle** out)
{
  if (!sp || !dp || !out || !(sp->voxel_size > 0) || sp->sensor_hrays < 2 || sp->sensor_vrays < 2)
    return VOFOD_ERR_INVALID_ARG;
  vofod_handle* h = new vofod_handle;
  h->sp = *sp;
  h->dp = *dp;
  const size_t n = static_cast<size_t>(sp->sensor_hrays) * sp->sensor_vrays;
  h->lut_dirs.resize(3 * n);
  if (sp->lut_directions)
    std::copy(sp->lut_directions, sp->lut_directions + 3 * n, h->lut_dirs.begin());
  else
    vo::sim_lut(sp->sensor_hrays, sp->sensor_vrays, sp->sensor_vfov, h->lut_dirs.data());
  h->lut_offs.assign(3 * n, 0.0f);
  if (sp->lut_offsets)
    std::copy(sp->lut_offsets, sp->lut_offsets + 3 * n, h->lut_offs.begin());
  h->mask.assign(n, 1);
  if (sp->mask)
    std::copy(sp->mask, sp->mask + n, h->mask.begin());
  h->sp.lut_directions = nullptr;
  h->sp.lut_offsets = nullptr;
  h->sp.mask = nullptr;
  for (int a = 0; a < 3; a++)
  {
    h->exclude_center[a] = sp->exclude_offset[a];
    h->oparea_center[a] = sp->oparea_offset[a];
  }
  h->exclude_center[2] = sp->exclude_offset[2] + sp->exclude_size[2] / 2.0f;  // vofod_nodelet.cpp:204
  h->oparea_center[2] = sp->oparea_offset[2] + sp->oparea_size[2] / 2.0f;     // :212
  const float n_voxels_xy = sp->oparea_size[0] / sp->voxel_size * sp->oparea_size[1] / sp->voxel_size;  // :229
  h->background_min_sufficient_pts = static_cast<uint64_t>(n_voxels_xy * sp->background_sufficient_points_ratio);  // :230
  do_reset(h);
  h->sure_background_sufficient = false;  // :283-284
  h->background_pts_sufficient = false;
  h->last_detection_id = 0;  // :296
  *out = h;
  return VOFOD_OK;
}

void ORACLE_API(destroy)(vofod_handle* h) { delete h; }

int ORACLE_API(reset)(vofod_handle* h)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  do_reset(h);
  h->sure_background_sufficient = false;
  h->background_pts_sufficient = false;
  h->last_detection_id = 0;
  return VOFOD_OK;
}

int ORACLE_API(set_dynamic_params)(vofod_handle* h, const vofod_dyn_params* dp)
{
  if (!h || !dp)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  h->dp = *dp;
  return VOFOD_OK;
}

const char* ORACLE_API(last_error_string)(vofod_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ORACLE_API(get_status)(vofod_handle* h, vofod_status_info* out)
{
  if (!h || !out)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  out->detection_its = h->detection_its;
  out->last_detection_id = h->last_detection_id;
  out->background_pts_sufficient = h->background_pts_sufficient;
  out->sure_background_sufficient = h->sure_background_sufficient;
  out->raycast_pending = h->raycast_pending;
  out->map_size[0] = h->vmap.sx;
  out->map_size[1] = h->vmap.sy;
  out->map_size[2] = h->vmap.sz;
  for (int a = 0; a < 3; a++)
    out->map_offset[a] = h->vmap.off[a];
  return VOFOD_OK;
}

// initialize_apriori_map vofod_nodelet.cpp:339-345
int ORACLE_API(load_apriori)(vofod_handle* h, const float* xyz, size_t n)
{
  if (!h || (!xyz && n))
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  for (size_t i = 0; i < n; i++)
  {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (h->vmap.inLimits(x, y, z))
    {
      const auto c = h->vmap.coordToIdx(x, y, z);
      h->vmap.at(c[0], c[1], c[2]) = std::numeric_limits<float>::infinity();
    }
  }
  h->sure_background_sufficient = true;
  h->background_pts_sufficient = true;
  return VOFOD_OK;
}

static vo::VoxelMap* pick_map(vofod_handle* h, int which)
{
  switch (which)
  {
    case VOFOD_MAP_VOXELS: return &h->vmap;
    case VOFOD_MAP_FLAGS: return &h->vflags;
    case VOFOD_MAP_RAYCAST: return &h->vraycast;
  }
  return nullptr;
}

int ORACLE_API(read_map)(vofod_handle* h, int which, float* dst, size_t n)
{
  if (!h || !dst)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  vo::VoxelMap* m = pick_map(h, which);
  if (!m || n != m->size())
    return VOFOD_ERR_SIZE_MISMATCH;
  std::copy(m->data.begin(), m->data.end(), dst);
  return VOFOD_OK;
}

int ORACLE_API(update_ground)(vofod_handle* h, float range, float min_range, float max_range, const float tf[12])
{
  if (!h || !tf)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  if (range <= min_range && range >= max_range)  // :585, as written
    return VOFOD_OK;
  // :597  Affine3f * (range, 0, 0): linear * v + translation, the zero terms add nothing
  const float x = tf[0] * range + tf[3], y = tf[4] * range + tf[7], z = tf[8] * range + tf[11];
  if (!h->vmap.inLimits(x, y, z))  // :601-605
    return VOFOD_ERR_MAP_RANGE;
  const auto ci = h->vmap.coordToIdx(x, y, z);  // VoxelMap::at(x, y, z) = at(coordToIdx(...)) voxel_map.cpp:116-117
  float& mapval = h->vmap.at(ci[0], ci[1], ci[2]);
  mapval = static_cast<float>((static_cast<double>(mapval) + h->dp.voxel_map__scores__point) / 2.0);  // :609 (float + double config value)
  return VOFOD_OK;
}

int ORACLE_API(write_map)(vofod_handle* h, int which, const float* src, size_t n)
{
  if (!h || !src)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  vo::VoxelMap* m = pick_map(h, which);
  if (!m || n != m->size())
    return VOFOD_ERR_SIZE_MISMATCH;
  std::copy(src, src + n, m->data.begin());
  return VOFOD_OK;
}

int ORACLE_API(process_scan)(vofod_handle* h, const vofod_scan* scan, const float tf[12], int flags, vofod_detection* out, size_t cap, size_t* n_out,
                             vofod_scan_debug* dbg)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  return process_scan_locked(h, scan, tf, flags, 0, out, cap, n_out, dbg);
}

int ORACLE_API(process_batch)(vofod_handle* h, const vofod_scan* scans, const float* tfs, size_t n, vofod_detection* out, size_t cap,
                              uint32_t* n_out_per_frame, size_t* n_out, vofod_scan_debug* dbg)
{
  if (!h || !scans || !tfs || !n_out)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  size_t total = 0;
  int ret = VOFOD_OK;
  for (size_t f = 0; f < n; f++)
  {
    size_t nf = 0;
    const size_t room = total < cap ? cap - total : 0;
    const int r = process_scan_locked(h, &scans[f], tfs + 12 * f, VOFOD_SCAN_NO_MAP_UPDATE, static_cast<uint32_t>(f), out ? out + total : nullptr, room, &nf,
                                      dbg ? &dbg[f] : nullptr);
    if (r != VOFOD_OK)
      ret = r;
    if (dbg && dbg[0].far_only && r == VOFOD_OK)
    {
      // the far-only view of include/vofod.h: the oracle clusters everything as the reference does (clusterCloud :932, then
      // findCloseFarClusters :727-748) and shows the far part - far clusters in their order, VOFOD_LABEL_NONE elsewhere
      vofod_scan_debug& d = dbg[f];
      std::vector<uint32_t> far_roots;
      size_t n_far = 0;
      if (d.clusters && d.clusters_cap >= d.n_clusters)
      {
        for (size_t c = 0; c < d.n_clusters; c++)
          if (!d.clusters[c].is_close)
          {
            far_roots.push_back(d.clusters[c].first_member);
            d.clusters[n_far++] = d.clusters[c];
          }
        d.n_clusters = n_far;
        std::sort(far_roots.begin(), far_roots.end());
        if (d.labels && d.weighted_cap >= d.n_weighted)
          for (size_t v = 0; v < d.n_weighted; v++)
            if (!std::binary_search(far_roots.begin(), far_roots.end(), d.labels[v]))
              d.labels[v] = VOFOD_LABEL_NONE;
      }
      else if (d.clusters)
        ret = VOFOD_ERR_CAPACITY;  // the full table does not fit: no far view can be cut from it (the HIP library reports the same)
    }
    if (n_out_per_frame)
      n_out_per_frame[f] = static_cast<uint32_t>(nf);
    total += nf;
  }
  *n_out = total;
  return ret;
}

int ORACLE_API(reserve)(vofod_handle* h, int tickets)
{
  return (h && tickets >= 1 && tickets <= 8) ? VOFOD_OK : VOFOD_ERR_INVALID_ARG;  // nothing to allocate on the CPU
}

int ORACLE_API(batch_submit)(vofod_handle* h, const vofod_scan* scans, const float* tfs, size_t n, int* ticket)
{
  if (!h || !scans || !tfs || !ticket)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  int t = -1;
  for (int i = 0; i < 4 && t < 0; i++)
    if (!h->tickets[i].pending)
      t = i;
  if (t < 0)
    return VOFOD_ERR_CAPACITY;
  auto& T = h->tickets[t];
  T.dets.clear();
  T.per_frame.assign(n, 0);
  T.status = VOFOD_OK;
  // ids are handed out at collect time (collect order), as in the product: compute with a scratch counter and renumber then
  const uint32_t id0 = h->last_detection_id;
  for (size_t f = 0; f < n; f++)
  {
    vofod_detection buf[256];
    size_t nf = 0;
    const int r = process_scan_locked(h, &scans[f], tfs + 12 * f, VOFOD_SCAN_NO_MAP_UPDATE, static_cast<uint32_t>(f), buf, 256, &nf, nullptr);
    if (r != VOFOD_OK)
      T.status = r;
    T.per_frame[f] = static_cast<uint32_t>(nf);
    T.dets.insert(T.dets.end(), buf, buf + std::min<size_t>(nf, 256));
  }
  h->last_detection_id = id0;
  T.pending = true;
  *ticket = t;
  return VOFOD_OK;
}

int ORACLE_API(batch_collect)(vofod_handle* h, int ticket, vofod_detection* out, size_t cap, uint32_t* n_out_per_frame, size_t* n_out)
{
  if (!h || !n_out || ticket < 0 || ticket > 3)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  auto& T = h->tickets[ticket];
  if (!T.pending)
    return VOFOD_ERR_NOT_PENDING;
  T.pending = false;
  for (auto& d : T.dets)
    d.id = h->last_detection_id++;
  *n_out = T.dets.size();
  if (n_out_per_frame)
    std::copy(T.per_frame.begin(), T.per_frame.end(), n_out_per_frame);
  if (T.dets.size() > cap)
    return VOFOD_ERR_CAPACITY;
  if (out)
    std::copy(T.dets.begin(), T.dets.end(), out);
  return T.status;
}

int ORACLE_API(raycast_begin)(vofod_handle* h, const vofod_scan* scan, const float tf[12])
{
  if (!h || !tf)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  return raycast_begin_locked(h, scan, tf);
}

int ORACLE_API(raycast_finish)(vofod_handle* h)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  return raycast_finish_locked(h);
}

int ORACLE_API(sepclusters_begin)(vofod_handle* h, int* sure)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  return sepclusters_begin_locked(h, sure);
}

int ORACLE_API(sepclusters_finish)(vofod_handle* h)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  return sepclusters_finish_locked(h);
}

int ORACLE_API(voxel_grid_weighted)(vofod_handle*, const vofod_cloud_view* in, float leaf, int align, const float align_center[3], vofod_point_xyzr* out,
                                    uint32_t* keys, size_t cap, size_t* n_out, vofod_grid_desc* grid)
{
  if (!in || in->memspace != VOFOD_MEM_HOST || !(leaf > 0) || (align && !align_center))
    return VOFOD_ERR_INVALID_ARG;
  const vo::Cloud c = cloud_from_view(in, false);
  const float l[3] = {leaf, leaf, leaf};
  const float zero[3] = {0, 0, 0};
  return emit_grid(vo::voxel_grid(c, l, align != 0, align ? align_center : zero, false, 0.0f), out, keys, cap, n_out, grid);
}

int ORACLE_API(voxel_grid_counted)(vofod_handle*, const vofod_cloud_view* in, float leaf, float threshold, vofod_point_xyzr* out, uint32_t* keys, size_t cap,
                                   size_t* n_out, vofod_grid_desc* grid)
{
  if (!in || in->memspace != VOFOD_MEM_HOST || !(leaf > 0) || !in->intensity)
    return VOFOD_ERR_INVALID_ARG;
  const vo::Cloud c = cloud_from_view(in, true);
  const float l[3] = {leaf, leaf, leaf};
  const float zero[3] = {0, 0, 0};
  return emit_grid(vo::voxel_grid(c, l, false, zero, true, threshold), out, keys, cap, n_out, grid);
}

int ORACLE_API(cluster)(vofod_handle*, const vofod_point_xyzr* pts, const uint32_t*, const vofod_grid_desc*, size_t n, float tolerance, uint32_t* labels,
                        size_t* n_clusters)
{
  if ((!pts && n) || !labels)
    return VOFOD_ERR_INVALID_ARG;
  const std::vector<vofod_point_xyzr> v(pts, pts + n);
  const std::vector<uint32_t> l = vo::euclidean_labels(v, tolerance);
  std::copy(l.begin(), l.end(), labels);
  if (n_clusters)
  {
    size_t c = 0;
    for (size_t i = 0; i < n; i++)
      c += l[i] == i;
    *n_clusters = c;
  }
  return VOFOD_OK;
}

// O(n^2) check of the above, tests only
int vofod_oracle_cluster_bruteforce(const vofod_point_xyzr* pts, size_t n, float tolerance, uint32_t* labels)
{
  const std::vector<vofod_point_xyzr> v(pts, pts + n);
  const std::vector<uint32_t> l = vo::euclidean_labels_bruteforce(v, tolerance);
  std::copy(l.begin(), l.end(), labels);
  return VOFOD_OK;
}

// load_cloud pc_loader.cpp:17-90
int ORACLE_API(load_cloud)(const char* filename, float* xyz, size_t cap, size_t* n_out)
{
  if (!filename || !n_out)
    return VOFOD_ERR_INVALID_ARG;
  std::ifstream fs(filename, std::ios::binary);
  if (!fs.is_open() || fs.fail())
    return VOFOD_ERR_INVALID_ARG;  // :22-27 returns nullptr
  const std::string fname(filename);
  const std::string ftype = fname.substr(fname.find_last_of(".") + 1);
  std::string line;
  if (ftype == "pts")  // :36-41: the first line is the point count (only used to reserve)
    std::getline(fs, line);
  auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  size_t n = 0;
  while (!fs.eof())  // :52-83
  {
    std::getline(fs, line);
    if (line.empty())
      continue;
    size_t b = 0, e = line.size();
    while (b < e && is_space(line[b]))
      b++;
    while (e > b && is_space(line[e - 1]))
      e--;
    std::vector<std::string> st;  // boost::split(is_any_of("\t\r "), token_compress_on) on the trimmed line
    std::string tok;
    bool in_delim = false;
    for (size_t i = b; i < e; i++)
    {
      const char c = line[i];
      if (c == '\t' || c == '\r' || c == ' ')
      {
        if (!in_delim)
        {
          st.push_back(tok);
          tok.clear();
        }
        in_delim = true;
      }
      else
      {
        tok.push_back(c);
        in_delim = false;
      }
    }
    st.push_back(tok);
    if (st.size() < 3)
      continue;  // :65-69 warns and skips
    if (xyz && n < cap)
    {
      xyz[3 * n + 0] = float(atof(st[0].c_str()));
      xyz[3 * n + 1] = float(atof(st[1].c_str()));
      xyz[3 * n + 2] = float(atof(st[2].c_str()));
    }
    n++;
  }
  *n_out = n;
  return n > cap ? VOFOD_ERR_CAPACITY : VOFOD_OK;
}

int ORACLE_API(ingest_apriori)(vofod_handle* h, const char* filename, const float tf_xyz[3], double yaw_deg, const float sim_correction[3], size_t* n_loaded,
                               size_t* n_voxels)
{
  if (!h || !filename || !tf_xyz || !sim_correction)
    return VOFOD_ERR_INVALID_ARG;
  size_t n = 0;
  int r = ORACLE_API(load_cloud)(filename, nullptr, 0, &n);
  if (r != VOFOD_OK && r != VOFOD_ERR_CAPACITY)
    return r;
  std::vector<float> xyz(3 * n), cent;
  if (n && (r = ORACLE_API(load_cloud)(filename, xyz.data(), n, &n)) != VOFOD_OK)
    return r;
  vo::apriori_points(xyz, tf_xyz, yaw_deg, sim_correction, h->sp.voxel_size, cent);
  if (n_loaded)
    *n_loaded = n;
  if (n_voxels)
    *n_voxels = cent.size() / 3;
  return ORACLE_API(load_apriori)(h, cent.data(), cent.size() / 3);
}


// initialize_sensor_lut (vofod_nodelet.cpp:358-372).  [3P] ouster::make_xyz_lut restated from the published ouster_client
// sources (ouster_example 2.x lidar_scan.cpp; the reference pins no version): per pixel i = u*w + v
//   encoder = 2 pi - v * 2 pi / w,  azimuth = -az[u] deg,  altitude = alt[u] deg
//   direction = (cos(enc + az) cos(alt), sin(enc + az) cos(alt), sin(alt))
//   offset    = ((cos(enc), sin(enc), 0) - direction) * lidar_origin_to_beam_origin_mm
//   both rotated by the lidar_to_sensor transform (offset also translated), both scaled by range_unit;
// then the nodelet casts to float and normalises the directions column by column (:368-369).
int ORACLE_API(ouster_lut)(int32_t w, int32_t hh, double range_unit, double lidar_origin_to_beam_origin_mm, const double* tf16, const double* azimuth_deg, const double* altitude_deg,
                       float* directions, float* offsets)
{
  if (w < 1 || hh < 1 || !azimuth_deg || !altitude_deg || !directions || !offsets)
    return VOFOD_ERR_INVALID_ARG;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0};
  if (tf16)
    for (int r = 0; r < 3; r++)
    {
      for (int c = 0; c < 3; c++)
        R[3 * r + c] = tf16[4 * r + c];
      T[r] = tf16[4 * r + 3];
    }
  const double azimuth_radians = M_PI * 2.0 / w;
  for (int u = 0; u < hh; u++)
    for (int v = 0; v < w; v++)
    {
      const size_t i = static_cast<size_t>(u) * w + v;
      const double encoder = 2.0 * M_PI - (v * azimuth_radians);
      const double azimuth = -azimuth_deg[u] * M_PI / 180.0;
      const double altitude = altitude_deg[u] * M_PI / 180.0;
      double d[3] = {std::cos(encoder + azimuth) * std::cos(altitude), std::sin(encoder + azimuth) * std::cos(altitude), std::sin(altitude)};
      double o[3] = {std::cos(encoder) - d[0], std::sin(encoder) - d[1], -d[2]};
      for (int c = 0; c < 3; c++)
        o[c] *= lidar_origin_to_beam_origin_mm;
      // row vector times rot = transform.topLeftCorner(3,3).transpose(): element j = (d0*R[j][0] + d1*R[j][1]) + d2*R[j][2]
      double dr[3], orr[3];
      for (int j = 0; j < 3; j++)
      {
        dr[j] = (d[0] * R[3 * j] + d[1] * R[3 * j + 1]) + d[2] * R[3 * j + 2];
        orr[j] = ((o[0] * R[3 * j] + o[1] * R[3 * j + 1]) + o[2] * R[3 * j + 2]) + T[j];
      }
      float df[3];
      for (int j = 0; j < 3; j++)
      {
        df[j] = static_cast<float>(dr[j] * range_unit);
        offsets[3 * i + j] = static_cast<float>(orr[j] * range_unit);
      }
      const float norm = std::sqrt((df[0] * df[0] + df[1] * df[1]) + df[2] * df[2]);  // colwise().normalize() :369
      for (int j = 0; j < 3; j++)
        directions[3 * i + j] = df[j] / norm;
    }
  return VOFOD_OK;
}

// load_mask (vofod_nodelet.cpp:506-560) after cv::imread: the image (row-major, w x h, or NULL when the file is missing or
// has the wrong size) is copied as is, or "mangled" (:527-541) into the staggered column-major order of the raw Ouster
// packets, index ((v + pixel_shift_by_row[u]) % w) * h + u; entries no image provides are 1 (:558).
int ORACLE_API(mask_layout)(const uint8_t* image, int32_t w, int32_t hh, const int32_t* pixel_shift_by_row, int32_t mangle, uint8_t* mask)
{
  if (w < 1 || hh < 1 || !mask)
    return VOFOD_ERR_INVALID_ARG;
  const size_t n = static_cast<size_t>(w) * hh;
  if (!image)
  {
    std::fill(mask, mask + n, static_cast<uint8_t>(1));
    return VOFOD_OK;
  }
  if (!mangle)
  {
    std::copy(image, image + n, mask);
    return VOFOD_OK;
  }
  std::fill(mask, mask + n, static_cast<uint8_t>(0));  // std::vector::resize value-initialises (:519); every slot is written below
  for (int u = 0; u < hh; u++)
    for (int v = 0; v < w; v++)
    {
      const int shift = pixel_shift_by_row ? pixel_shift_by_row[u] : 0;
      const size_t vv = static_cast<size_t>(v + shift) % static_cast<size_t>(w);
      mask[vv * hh + u] = image[static_cast<size_t>(u) * w + v];
    }
  return VOFOD_OK;
}

// VoxelMap::voxelsAsPC (voxel_map.cpp:157-183), loop for loop
int ORACLE_API(voxels_as_pc)(vofod_handle* h, int which, float threshold, int greater_than, vofod_point_xyzi* out, size_t cap, size_t* n_out)
{
  if (!h || !n_out || (cap && !out))
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  vo::VoxelMap* m = which == VOFOD_MAP_VOXELS ? &h->vmap : which == VOFOD_MAP_FLAGS ? &h->vflags : which == VOFOD_MAP_RAYCAST ? &h->vraycast : nullptr;
  if (!m)
    return VOFOD_ERR_INVALID_ARG;
  size_t n = 0;
  for (int x_it = 0; x_it < m->sx; x_it++)
    for (int y_it = 0; y_it < m->sy; y_it++)
      for (int z_it = 0; z_it < m->sz; z_it++)
      {
        const float mapval = m->data[static_cast<size_t>(x_it) + static_cast<size_t>(y_it) * m->sx + static_cast<size_t>(z_it) * m->sx * m->sy];
        if ((mapval > threshold) == (greater_than != 0))
        {
          if (n < cap)
          {
            const auto c = m->idxToCoord(x_it, y_it, z_it);
            out[n] = vofod_point_xyzi{c[0], c[1], c[2], mapval};
          }
          n++;
        }
      }
  *n_out = n;
  return n > cap ? VOFOD_ERR_CAPACITY : VOFOD_OK;
}

// check_sensor_params (vofod_nodelet.cpp:1869-1917), statement by statement: rows outer, columns inner, the first pixel with
// mask != 0 and range != 0 decides.  Eigen: (a - b).normalized() = v / sqrt(v.squaredNorm()), norm() = sqrt(x*x + y*y + z*z).
int ORACLE_API(check_sensor_params)(const vofod_scan* scan, const float* lut_directions, const float* lut_offsets, const uint8_t* mask, int32_t* checked)
{
  if (!scan || !scan->x || !scan->y || !scan->z || !scan->range || !lut_directions || scan->memspace != VOFOD_MEM_HOST)
    return VOFOD_ERR_INVALID_ARG;
  bool found_valid = false, params_ok = true;
  const float range_to_meters = 0.001f;
  auto col_f = [&](const void* base, unsigned idx) { return *reinterpret_cast<const float*>(static_cast<const char*>(base) + static_cast<size_t>(idx) * scan->stride_bytes); };
  auto col_u = [&](const void* base, unsigned idx) { return *reinterpret_cast<const uint32_t*>(static_cast<const char*>(base) + static_cast<size_t>(idx) * scan->stride_bytes); };
  auto norm3 = [](const float v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  for (int row = 0; row < static_cast<int>(scan->height) && !found_valid; row++)
    for (int col = 0; col < static_cast<int>(scan->width) && !found_valid; col++)
    {
      const unsigned idx = row * scan->width + col;
      const uint32_t range = col_u(scan->range, idx);
      if ((mask && !mask[idx]) || range == 0)  // :1882
        continue;
      const float* lut_dir = lut_directions + 3 * static_cast<size_t>(idx);
      const float lut_dist = range_to_meters * float(range);
      float v[3] = {col_f(scan->x, idx), col_f(scan->y, idx), col_f(scan->z, idx)};
      if (lut_offsets)
        for (int a = 0; a < 3; a++)
          v[a] = v[a] - lut_offsets[3 * static_cast<size_t>(idx) + a];
      const float pt_dist = norm3(v);                                                     // :1889
      const float pt_dir[3] = {v[0] / pt_dist, v[1] / pt_dist, v[2] / pt_dist};           // :1888
      const float diff[3] = {pt_dir[0] - lut_dir[0], pt_dir[1] - lut_dir[1], pt_dir[2] - lut_dir[2]};
      if (norm3(diff) > 1e-3f)  // :1891
        params_ok = false;
      if (std::abs(pt_dist - lut_dist) > 1e-3f)  // :1896
        params_ok = false;
      if (1.0f - norm3(lut_dir) > 1e-3f)  // :1901
        params_ok = false;
      found_valid = true;
    }
  if (checked)
    *checked = found_valid ? 1 : 0;
  return params_ok ? VOFOD_OK : VOFOD_ERR_SIZE_MISMATCH;
}

int ORACLE_API(sim_lut)(int32_t w, int32_t h, float vfov, float* directions)
{
  if (w < 2 || h < 2 || !directions)
    return VOFOD_ERR_INVALID_ARG;
  vo::sim_lut(w, h, vfov, directions);
  return VOFOD_OK;
}

int ORACLE_API(profile_enable)(vofod_handle*, int) { return VOFOD_OK; }
size_t ORACLE_API(profile_read)(vofod_handle*, char*, double*, uint64_t*, size_t) { return 0; }

// ---- direct VoxelMap access for the known-answer tests (tests/test_oracle_kat.py)
int vofod_oracle_map_has_close_to(vofod_handle* h, float x, float y, float z, float max_dist, float thr) { return h->vmap.hasCloseTo(x, y, z, max_dist, thr); }

int vofod_oracle_map_explore_to_ground(vofod_handle* h, float x, float y, float z, float unknown_thr, float ground_thr, float max_voxel_dist, int* explored_xyz,
                                       size_t cap, size_t* n_explored)
{
  const auto [connected, explored] = h->vmap.exploreToGround(x, y, z, unknown_thr, ground_thr, max_voxel_dist);
  *n_explored = explored.size();
  for (size_t i = 0; i < explored.size() && i < cap; i++)
    for (int a = 0; a < 3; a++)
      explored_xyz[3 * i + a] = explored[i][a];
  return connected;
}

// walks one ray through the *raycast* map geometry and returns the visited voxels and path lengths
int vofod_oracle_map_ray(vofod_handle* h, const float start[3], const float dir[3], float length, int* vox_xyz, float* ddist, size_t cap, size_t* n_steps)
{
  size_t n = 0;
  h->vraycast.forEachRay(start, dir, length, [&](float d, int x, int y, int z) {
    if (n < cap)
    {
      vox_xyz[3 * n] = x;
      vox_xyz[3 * n + 1] = y;
      vox_xyz[3 * n + 2] = z;
      ddist[n] = d;
    }
    n++;
  });
  *n_steps = n;
  return VOFOD_OK;
}

int vofod_oracle_map_coord_to_idx(vofod_handle* h, float x, float y, float z, int out[3])
{
  const auto c = h->vmap.coordToIdx(x, y, z);
  for (int a = 0; a < 3; a++)
    out[a] = c[a];
  return h->vmap.inLimitsIdx(c[0], c[1], c[2]);
}

// test hook: 0 = Eigen::EigenSolver restatement (default, what PCL calls), 1 = cyclic Jacobi in double (cross-check)
int vofod_oracle_set_obb_solver(int solver)
{
  vo::obb_solver() = solver == 1 ? 1 : 0;
  return VOFOD_OK;
}

int vofod_oracle_moie(const vofod_point_xyzr* pts, size_t n, float* aabb_min, float* aabb_max, float* obb_center, float* obb_size, float* eig)
{
  std::vector<vofod_point_xyzr> v(pts, pts + n);
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  const vo::Boxes b = vo::moie(v, idx);
  for (int a = 0; a < 3; a++)
  {
    aabb_min[a] = b.aabb_min[a];
    aabb_max[a] = b.aabb_max[a];
    obb_center[a] = b.obb_center[a];
    eig[a] = b.eig[a];
  }
  const float d[3] = {b.obb_max[0] - b.obb_min[0], b.obb_max[1] - b.obb_min[1], b.obb_max[2] - b.obb_min[2]};
  *obb_size = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  return VOFOD_OK;
}

}  // extern "C"
