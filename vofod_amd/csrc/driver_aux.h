// Second half of the driver (included by vofod_hip.hip): raycast, sepclusters, stateless L4 entry points
// and the extern "C" surface of include/vofod.h.
#pragma once

namespace
{

int gscan(vofod_handle* h, const uint32_t* d_in, uint32_t n, uint32_t* d_out, uint32_t* d_bsum, uint32_t* d_total)
{
  const uint32_t nblk = (n + vr::GS_EPB - 1) / vr::GS_EPB;
  if (n == 0)
  {
    HIPCHK(hipMemsetAsync(d_out, 0, sizeof(uint32_t), h->stream));
    if (d_total)
      HIPCHK(hipMemsetAsync(d_total, 0, sizeof(uint32_t), h->stream));
    return VOFOD_OK;
  }
  KLAUNCH(h, vr::k_gscan_a, dim3(nblk), dim3(256), d_in, n, d_bsum);
  KLAUNCH(h, vr::k_gscan_b, dim3(1), dim3(1024), d_bsum, nblk, d_total);
  KLAUNCH(h, vr::k_gscan_c, dim3(nblk), dim3(256), d_in, n, d_bsum, d_out);
  HIPCHK(hipGetLastError());
  return VOFOD_OK;
}

template <class T>
int regrow(vofod_handle* h, T*& p, size_t n)
{
  if (p)
    (void)hipFree(p);
  p = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)));
  return VOFOD_OK;
}

int sep_ensure_words(vofod_handle* h, size_t n_words)
{
  vr::SepState& s = h->sep;
  if (!s.d_small)
  {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&s.d_small), 16 * sizeof(uint32_t)));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s.h_small), 16 * sizeof(uint32_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&s.d_offsets), 3 * (2 * MAX_R + 1) * (2 * MAX_R + 1) * (2 * MAX_R + 1) * sizeof(int)));
  }
  if (n_words <= s.words_cap)
    return VOFOD_OK;
  int r;
  if ((r = regrow(h, s.d_tbits, n_words + 2)) || (r = regrow(h, s.d_tpop, n_words + 2)) || (r = regrow(h, s.d_tprefix, n_words + 2)))
    return r;
  s.words_cap = n_words;
  return VOFOD_OK;
}

int sep_ensure_pts(vofod_handle* h, size_t n)
{
  vr::SepState& s = h->sep;
  if (n <= s.pts_cap)
    return VOFOD_OK;
  const size_t cap = n + n / 4 + 1024;
  int r;
  if ((r = regrow(h, s.d_px, cap)) || (r = regrow(h, s.d_py, cap)) || (r = regrow(h, s.d_pz, cap)) || (r = regrow(h, s.d_pi, cap)) || (r = regrow(h, s.d_sure, cap)) ||
      (r = regrow(h, s.d_sure_pre, cap + 1)) || (r = regrow(h, s.d_vcnt, cap)) || (r = regrow(h, s.d_first, cap + 1)) || (r = regrow(h, s.d_nsure, cap)) ||
      (r = regrow(h, s.d_bsum, cap / vr::GS_EPB + 1024)))
    return r;
  s.pts_cap = cap;
  return VOFOD_OK;
}

// ------------------------------------------------------------------ raycast_cloud :1397-1605

int raycast_begin_locked(vofod_handle* h, const vofod_scan* scan, const float tf[12])
{
  const vofod_dyn_params& dp = h->dp;
  if (h->raycast_pending)
    return VOFOD_ERR_INVALID_ARG;
  if (dp.raycast__pause)
    return VOFOD_ERR_PAUSED;
  if (!scan || !scan->intensity || !scan->range)
    return VOFOD_ERR_INVALID_ARG;
  if (scan->height != h->sp.sensor_vrays || scan->width != h->sp.sensor_hrays)
    return VOFOD_ERR_SIZE_MISMATCH;
  h->raycast_start_its = h->detection_its;
  h->raycast_pending = true;
  const uint32_t n = static_cast<uint32_t>(scan->width) * scan->height;
  // stage intensity/range if they live on the host
  const char *d_int, *d_rng;
  uint64_t stride;
  if (scan->memspace == VOFOD_MEM_DEVICE)
  {
    d_int = static_cast<const char*>(scan->intensity);
    d_rng = static_cast<const char*>(scan->range);
    stride = scan->stride_bytes;
  }
  else
  {
    const int r = stage_cloud(h, h->ws, 0, scan->x ? scan->x : scan->intensity, scan->y ? scan->y : scan->intensity, scan->z ? scan->z : scan->intensity,
                              scan->intensity, scan->range, scan->stride_bytes, n, VOFOD_MEM_HOST, 0, nullptr);
    if (r != VOFOD_OK)
      return r;
    float* base = h->ws.d_stage;
    d_int = reinterpret_cast<const char*>(base + 3 * static_cast<size_t>(h->ws.pt_cap));
    d_rng = reinterpret_cast<const char*>(base + 4 * static_cast<size_t>(h->ws.pt_cap));
    stride = 4;
  }
  // m_voxel_raycast.clear() :1430 — the sweep leaves it zeroed; clear only if an earlier pass was abandoned
  if (h->ray_dirty)
  {
    const int r = fill_map(h, h->d_ray, 0.0f);
    if (r != VOFOD_OK)
      return r;
  }
  h->ray_dirty = true;
  HIPCHK(hipMemsetAsync(h->d_counter + 1, 0, sizeof(unsigned long long), h->stream));
  vr::RayParams rp{};
  // [3P] Affine3f::rotation() == the linear part up to rounding for rigid transforms
  for (int i = 0; i < 3; i++)
  {
    for (int j = 0; j < 3; j++)
      rp.R[3 * i + j] = tf[4 * i + j];
    rp.origin[i] = tf[4 * i + 3];
  }
  rp.max_dist = static_cast<float>(dp.raycast__max_distance);
  rp.min_intensity = static_cast<float>(dp.raycast__min_intensity);
  rp.voxel_size = h->sp.voxel_size;
  rp.n = n;
  int o[3];
  h->hg.coordToIdx(rp.origin, o);
  int ret = VOFOD_OK;
  if (h->hg.inLimits(o))  // :1432
    KLAUNCH(h, vr::k_raycast, dim3((n + 255) / 256), dim3(256), rp, h->mg, d_int, d_rng, stride, h->d_lut_dirs, h->d_lut_offs, h->d_mask, h->d_ray, reinterpret_cast<uint32_t*>(h->d_counter + 1));
  else
    ret = VOFOD_ERR_SENSOR_OUTSIDE_MAP;
  HIPCHK(hipStreamSynchronize(h->stream));
  return ret;
}

int raycast_finish_locked(vofod_handle* h)
{
  if (!h->raycast_pending)
    return VOFOD_ERR_NOT_PENDING;
  h->raycast_pending = false;
  const vofod_dyn_params& dp = h->dp;
  if (h->detection_its == h->raycast_start_its)  // :1531-1537
    return VOFOD_ERR_RAYCAST_NO_DETECTION;
  vr::SweepParams sp{};
  sp.its_diff = static_cast<float>(h->detection_its - h->raycast_start_its);
  sp.ray_score = static_cast<float>(dp.voxel_map__scores__ray);
  const float weight = static_cast<float>(dp.raycast__weight_coefficient);
  const float voxel_diag = static_cast<float>(std::sqrt(3) * h->sp.voxel_size);
  sp.weighting_factor = weight / voxel_diag;
  sp.weight = weight;
  sp.new_rule = dp.raycast__new_update_rule;
  // max_val :1542 — zero iff no ray added a positive length; the old rule needs the value itself
  HIPCHK(hipMemcpyAsync(h->h_counter + 1, h->d_counter + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (static_cast<uint32_t>(h->h_counter[1]) == 0)
    return VOFOD_ERR_RAYCAST_EMPTY;  // :1544-1548 (flags stay as they are)
  if (!sp.new_rule)
  {
    HIPCHK(hipMemsetAsync(h->d_counter + 2, 0, sizeof(unsigned long long), h->stream));
    KLAUNCH(h, vr::k_max_nonneg, dim3(2048), dim3(256), h->d_ray, h->mg.n, reinterpret_cast<uint32_t*>(h->d_counter + 2));
    HIPCHK(hipMemcpyAsync(h->h_counter + 2, h->d_counter + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const uint32_t bits = static_cast<uint32_t>(h->h_counter[2]);
    std::memcpy(&sp.max_val, &bits, 4);
  }
  KLAUNCH(h, vr::k_ray_sweep, dim3(256 * 8), dim3(256), sp, h->mg.n, h->d_map, h->d_flags, h->d_ray);
  HIPCHK(hipStreamSynchronize(h->stream));
  h->ray_dirty = false;
  h->mapbits_valid = false;
  return VOFOD_OK;
}

// ------------------------------------------------------------------ counted grid tail

// ws holds a finished weighted voxelisation of one frame; replace the weights by the positional counts.
int counted_tail(vofod_handle* h, Workspace& ws, uint32_t V, uint32_t P, const uint32_t* d_sure_flags)
{
  vr::SepState& s = h->sep;
  int r = gscan(h, d_sure_flags, P, s.d_sure_pre, s.d_bsum, nullptr);
  if (r != VOFOD_OK)
    return r;
  KLAUNCH(h, vr::k_voxel_counts, dim3((V + 255) / 256), dim3(256), ws.d_hdrs, ws.va.pts, s.d_vcnt);
  r = gscan(h, s.d_vcnt, V, s.d_first, s.d_bsum, nullptr);
  if (r != VOFOD_OK)
    return r;
  KLAUNCH(h, vr::k_counted_range, dim3((V + 255) / 256), dim3(256), ws.d_hdrs, s.d_first, s.d_sure_pre, P, ws.va.pts);
  HIPCHK(hipGetLastError());
  return VOFOD_OK;
}

// voxelise one device/host cloud into `ws` (frame 0), growing the workspace to the lattice it needs
int voxelize_cloud(vofod_handle* h, Workspace& ws, const vofod_cloud_view* in, GridParams& g, const float leaf[3], bool align, const float* align_center, FrameHdr& hdr)
{
  const uint32_t n = static_cast<uint32_t>(in->n);
  // Growth with headroom: a workspace grows by releasing and allocating its three dozen arrays (2.6 ms), and the cloud of
  // updateSeparatedBGClusters - the map's background voxels - gains a few thousand points from one call to the next while the map
  // warms: sized to the point, EVERY call of the role paid that (VOFOD_TRACE: 2.7 of the role's 2.9 ms).
  const uint32_t want = n > std::min(ws.pt_cap, ws.vox_cap) ? n + n / 2 + 4096u : std::max(n, 1u);
  if (hipError_t e = ws.ensure(1, want, want, std::max(ws.words_cap, 1u << 16)); e != hipSuccess)
  {
    h->err = std::string("workspace allocation: ") + hipGetErrorString(e);
    return VOFOD_ERR_DEVICE;
  }
  for (int attempt = 0; attempt < 2; attempt++)
  {
    int r = stage_cloud(h, ws, 0, in->x, in->y, in->z, in->intensity, nullptr, in->stride_bytes, n, in->memspace, 0, nullptr);
    if (r != VOFOD_OK)
      return r;
    const float zero[3] = {0, 0, 0};
    fill_grid_params(h, g, leaf, align, align ? align_center : zero, ws);
    r = launch_voxelize(h, ws, g, 1, n, false, true);
    if (r != VOFOD_OK)
      return r;
    HIPCHK(hipMemcpyAsync(&ws.h_packed[0].hdr, ws.d_hdrs, sizeof(FrameHdr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    hdr = ws.h_packed[0].hdr;
    const uint32_t need_bricks =
        hdr.n_in ? static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>((hdr.div_b[0] + 3) / 4) * ((hdr.div_b[1] + 3) / 4) * ((hdr.div_b[2] + 3) / 4), 1u << 28)) : 0u;
    if ((hdr.status == VOFOD_ERR_CAPACITY || (hdr.status == VOFOD_OK && need_bricks > ws.bricks_cap)) && attempt == 0)
    {
      if (hipError_t e = ws.ensure(1, n, n, hdr.need_words + hdr.need_words / 4 + 64, need_bricks + need_bricks / 4); e != hipSuccess)
      {
        h->err = std::string("workspace allocation: ") + hipGetErrorString(e);
        return VOFOD_ERR_DEVICE;
      }
      continue;
    }
    break;
  }
  if (hdr.status != VOFOD_OK)
    return hdr.status;
  if (hdr.n_in == 0)
  {
    hdr.V = 0;
    return VOFOD_OK;
  }
  int r = launch_voxelize_rest(h, ws, g, 1, n, false);
  if (r != VOFOD_OK)
    return r;
  HIPCHK(hipMemcpyAsync(&ws.h_packed[0].hdr, ws.d_hdrs, sizeof(FrameHdr), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  hdr = ws.h_packed[0].hdr;
  return hdr.status;
}

// ------------------------------------------------------------------ updateSeparatedBGClusters :1126-1277

int sepclusters_begin_locked(vofod_handle* h, int* sure_out)
{
  const vofod_dyn_params& dp = h->dp;
  if (sure_out)
    *sure_out = h->sure_background_sufficient;
  if (dp.sepclusters__pause)
    return VOFOD_ERR_PAUSED;
  h->sep_pending = false;
  h->sep_start_its = h->detection_its;
  vr::SepState& s = h->sep;
  static const bool trace = std::getenv("VOFOD_TRACE") != nullptr;
  const auto t0 = clk::now();
  double tr[5] = {0, 0, 0, 0, 0};
  const float thr_new = static_cast<float>(dp.voxel_map__thresholds__new_obstacles);
  const float thr_sure = static_cast<float>(dp.voxel_map__thresholds__sure_obstacles);
  const float max_dist_idx = static_cast<float>(dp.sepclusters__max_bg_distance / h->sp.voxel_size);
  const int max_voxel_dist = static_cast<int>(std::ceil(max_dist_idx));
  const float lsz = static_cast<float>(std::max(max_voxel_dist - 1, 0));
  if (!(lsz > 0.0f))
    return VOFOD_ERR_INVALID_ARG;

  // K16: thresholded voxels in x-outer / z-inner order (voxelsAsVoxelPC :1153)
  int r = ensure_mapbits(h, thr_new);
  if (r != VOFOD_OK)
    return r;
  const uint32_t ncol = static_cast<uint32_t>(h->mg.sx) * h->mg.sy;
  if ((r = sep_ensure_words(h, ncol + 2)) != VOFOD_OK)
    return r;
  if ((r = sep_ensure_pts(h, std::max<size_t>(1 << 16, ncol))) != VOFOD_OK)  // also sizes the scan's block sums
    return r;
  KLAUNCH(h, vr::k_col_count, dim3((ncol + 255) / 256), dim3(256), h->mg, h->d_mapbits, s.d_tpop);
  if ((r = gscan(h, s.d_tpop, ncol, s.d_tprefix, s.d_bsum, s.d_small)) != VOFOD_OK)
    return r;
  HIPCHK(hipMemcpyAsync(s.h_small, s.d_small, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const uint32_t P = s.h_small[0];
  s.P = P;
  tr[0] = ms_since(t0);
  if (P == 0)
    return VOFOD_ERR_EMPTY;  // :1155-1159
  if ((r = sep_ensure_pts(h, P)) != VOFOD_OK)
    return r;
  KLAUNCH(h, vr::k_col_emit, dim3((ncol + 255) / 256), dim3(256), h->mg, h->d_map, h->d_mapbits, s.d_tprefix, thr_sure, s.d_px, s.d_py, s.d_pz, s.d_pi, s.d_sure);

  // K6': VoxelGridCounted with leaf lsz on the index cloud (:1162-1167)
  vofod_cloud_view view{};
  view.x = s.d_px;
  view.y = s.d_py;
  view.z = s.d_pz;
  view.intensity = s.d_pi;
  view.stride_bytes = 4;
  view.n = P;
  view.memspace = VOFOD_MEM_DEVICE;
  const float leaf[3] = {lsz, lsz, lsz};
  FrameHdr hdr;
  r = voxelize_cloud(h, h->sepws, &view, s.g, leaf, false, nullptr, hdr);
  if (r != VOFOD_OK)
    return r;
  Workspace& ws = h->sepws;
  tr[1] = ms_since(t0);
  if ((r = counted_tail(h, ws, hdr.V, P, s.d_sure)) != VOFOD_OK)
    return r;
  tr[2] = ms_since(t0);

  // clusterCloud(vmap_pc_ds, max_voxel_dist) :1171
  const float cmax = static_cast<float>(std::max({h->mg.sx, h->mg.sy, h->mg.sz})) + 2 * lsz;
  if ((r = launch_cluster(h, ws, s.g, 1, static_cast<float>(max_voxel_dist), cmax)) != VOFOD_OK)
    return r;
  tr[3] = ms_since(t0);
  // sure voxels per cluster :1175-1183 and the latch :1188-1206
  HIPCHK(hipMemsetAsync(s.d_nsure, 0, sizeof(uint32_t) * std::max(hdr.V, 1u), h->stream));
  HIPCHK(hipMemsetAsync(s.d_small + 1, 0, 2 * sizeof(uint32_t), h->stream));
  const uint32_t gv = (hdr.V + 255) / 256;
  KLAUNCH(h, vr::k_cluster_sure, dim3(gv), dim3(256), ws.d_hdrs, ws.va.pts, ws.d_labels, s.d_nsure);
  KLAUNCH(h, vr::k_any_sure, dim3(gv), dim3(256), ws.d_hdrs, ws.d_labels, s.d_nsure, static_cast<uint32_t>(dp.sepclusters__min_sure_points), s.d_small + 1);
  HIPCHK(hipMemcpyAsync(s.h_small, s.d_small, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (trace)
    std::fprintf(stderr, "[vofod trace] sepclusters_begin: thresholded cloud (P = %u) %.3f, counted grid (V = %u) %.3f, counts %.3f, clustering enqueued %.3f, end %.3f ms\n", P, tr[0], hdr.V, tr[1], tr[2],
                 tr[3], ms_since(t0));
  if (s.h_small[1] == 0)
  {
    h->sure_background_sufficient = false;  // :1195
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

int sepclusters_finish_locked(vofod_handle* h)
{
  if (!h->sep_pending)
    return VOFOD_ERR_NOT_PENDING;
  h->sep_pending = false;
  const vofod_dyn_params& dp = h->dp;
  vr::SepState& s = h->sep;
  const float max_dist_idx = static_cast<float>(dp.sepclusters__max_bg_distance / h->sp.voxel_size);
  const int mvd = static_cast<int>(std::ceil(max_dist_idx));
  const float its_diff = static_cast<float>(std::max(h->detection_its - h->sep_start_its, 1));  // :1212
  static const bool trace = std::getenv("VOFOD_TRACE") != nullptr;
  const auto t0 = clk::now();
  std::vector<int> offs;  // :1219-1237
  for (int x = -mvd; x <= mvd; x++)
    for (int y = -mvd; y <= mvd; y++)
      for (int z = -mvd; z <= mvd; z++)
      {
        const int norm = static_cast<int>(std::sqrt(static_cast<double>(x * x + y * y + z * z)));
        if (static_cast<float>(norm) <= max_dist_idx)
        {
          offs.push_back(x);
          offs.push_back(y);
          offs.push_back(z);
        }
      }
  if (mvd > MAX_R)
    return VOFOD_ERR_INVALID_ARG;
  HIPCHK(hipMemcpyAsync(s.d_offsets, offs.data(), offs.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  vr::EraseParams ep{};
  ep.update_val = static_cast<float>(dp.voxel_map__scores__ray);
  ep.w1 = std::clamp(std::pow(1.0f - 0.5f, its_diff), 0.0f, 1.0f);  // :1240-1241
  ep.w2 = 1.0f - ep.w1;
  ep.min_sure = static_cast<uint32_t>(dp.sepclusters__min_sure_points);
  ep.n_offsets = static_cast<int>(offs.size() / 3);
  Workspace& ws = h->sepws;
  const uint32_t gv = (ws.vox_cap + 255) / 256;
  const uint32_t oy = static_cast<uint32_t>(std::max(1, std::min(ep.n_offsets / 16, 256)));
  KLAUNCH(h, vr::k_sep_erase, dim3(gv, oy), dim3(256), ep, h->mg, ws.d_hdrs, ws.va.pts, ws.d_labels, s.d_nsure, s.d_offsets, h->d_map);
  HIPCHK(hipStreamSynchronize(h->stream));
  if (trace)
    std::fprintf(stderr, "[vofod trace] sepclusters_finish: %d stencil offsets, %.3f ms\n", ep.n_offsets, ms_since(t0));
  h->mapbits_valid = false;
  return VOFOD_OK;
}

int copy_grid_out(vofod_handle* h, Workspace& ws, const GridParams& g, const FrameHdr& hdr, vofod_point_xyzr* out, uint32_t* keys, size_t cap, size_t* n_out,
                  vofod_grid_desc* grid)
{
  if (grid)
    for (int a = 0; a < 3; a++)
    {
      grid->leaf[a] = g.leaf[a];
      grid->offset[a] = hdr.offset[a];
      grid->min_b[a] = hdr.min_b[a];
      grid->div_b[a] = hdr.div_b[a];
    }
  if (n_out)
    *n_out = hdr.V;
  if (hdr.V > cap)
    return VOFOD_ERR_CAPACITY;
  if (hdr.V && out)
    HIPCHK(hipMemcpy(out, ws.va.pts, sizeof(float4) * hdr.V, hipMemcpyDeviceToHost));
  if (hdr.V && keys)
    HIPCHK(hipMemcpy(keys, ws.va.key, sizeof(uint32_t) * hdr.V, hipMemcpyDeviceToHost));
  return VOFOD_OK;
}

}  // namespace

// =============================================================================== extern "C"
namespace
{
struct WireOut
{
  uint8_t* buf;
  size_t cap, n = 0;
  template <class T>
  void put(const T& v)
  {
    if (buf && n + sizeof(T) <= cap)
      std::memcpy(buf + n, &v, sizeof(T));
    n += sizeof(T);
  }
  void str(const char* s)
  {
    const uint32_t len = s ? static_cast<uint32_t>(std::strlen(s)) : 0u;
    put(len);
    if (buf && n + len <= cap)
      std::memcpy(buf + n, s, len);
    n += len;
  }
  void header(const vofod_msg_header* h)
  {
    put(h->seq);
    put(h->stamp_sec);
    put(h->stamp_nsec);
    str(h->frame_id);
  }
};
}  // namespace

extern "C" {

void vofod_default_params(vofod_static_params* sp, vofod_dyn_params* dp)
{
  if (sp)
  {
    *sp = vofod_static_params{};
    sp->voxel_size = 0.5f;                           // detection_params.yaml:17
    sp->score_init = -740.0f;                        // :21
    sp->background_sufficient_points_ratio = 0.15f;  // :9
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

int vofod_sim_lut(int32_t w, int32_t hh, float vfov, float* directions)
{
  if (w < 2 || hh < 2 || !directions)
    return VOFOD_ERR_INVALID_ARG;
  // initialize_sensor_lut_simulation vofod_nodelet.cpp:374-420
  const double yaw_step = (2.0 * M_PI - 0.0) / (w - 1);
  const double pitch_min = -vfov / 2.0, pitch_max = vfov / 2.0;
  const double pitch_step = (pitch_max - pitch_min) / (hh - 1);
  for (int row = 0; row < hh; row++)
    for (int col = 0; col < w; col++)
    {
      const double y = col * yaw_step + 0.0, p = row * pitch_step + pitch_min;
      float* d = directions + 3 * (static_cast<size_t>(row) * w + col);
      d[0] = static_cast<float>(std::cos(p) * std::cos(y));
      d[1] = static_cast<float>(std::cos(p) * std::sin(y));
      d[2] = static_cast<float>(std::sin(p));
    }
  return VOFOD_OK;
}


// initialize_sensor_lut (vofod_nodelet.cpp:358-372): the beam model of [3P] ouster::make_xyz_lut (ouster_client
// lidar_scan.cpp; the reference pins no version) followed by the nodelet's float cast and per-column normalisation.
// A beam of ring u at encoder column v leaves the lidar frame's z axis at radius n (= lidar_origin_to_beam_origin_mm) in the
// encoder direction and points along (azimuth, altitude) of its ring:
//   theta_enc = 2 pi - v 2 pi / w,   theta = theta_enc - azimuth[u],   phi = altitude[u]
//   dir = (cos theta cos phi, sin theta cos phi, sin phi),   off = ((cos theta_enc, sin theta_enc, 0) - dir) n
// both taken to the sensor frame by lidar_to_sensor (off also translated) and scaled by range_unit.
// Product implementation: ring constants first, then a plain structure-of-rows sweep (the oracle restates the library loop).
namespace
{
struct BeamRing
{
  double azimuth, cos_alt, sin_alt;
};
inline void to_sensor(const double rot[3][3], const double* trans, const double in[3], double out[3])
{
  // Eigen row-vector product with the transposed rotation block: ((x r0 + y r1) + z r2), then the translation
  for (int j = 0; j < 3; j++)
  {
    out[j] = (in[0] * rot[j][0] + in[1] * rot[j][1]) + in[2] * rot[j][2];
    if (trans)
      out[j] += trans[j];
  }
}
}  // namespace

int vofod_ouster_lut(int32_t w, int32_t hh, double range_unit, double lidar_origin_to_beam_origin_mm, const double* tf16, const double* azimuth_deg, const double* altitude_deg,
                       float* directions, float* offsets)
{
  if (w < 1 || hh < 1 || !azimuth_deg || !altitude_deg || !directions || !offsets)
    return VOFOD_ERR_INVALID_ARG;
  double rot[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, trans[3] = {0, 0, 0};
  if (tf16)
    for (int r = 0; r < 3; r++)
    {
      std::copy(tf16 + 4 * r, tf16 + 4 * r + 3, rot[r]);
      trans[r] = tf16[4 * r + 3];
    }
  std::vector<BeamRing> rings(hh);
  for (int u = 0; u < hh; u++)
  {
    const double alt = altitude_deg[u] * M_PI / 180.0;
    rings[u] = BeamRing{-azimuth_deg[u] * M_PI / 180.0, std::cos(alt), std::sin(alt)};
  }
  const double step = M_PI * 2.0 / w;
  float* dir_out = directions;
  float* off_out = offsets;
  for (int u = 0; u < hh; u++)
  {
    const BeamRing& ring = rings[u];
    for (int v = 0; v < w; v++, dir_out += 3, off_out += 3)
    {
      const double theta_enc = 2.0 * M_PI - (v * step), theta = theta_enc + ring.azimuth;
      const double dir[3] = {std::cos(theta) * ring.cos_alt, std::sin(theta) * ring.cos_alt, ring.sin_alt};
      const double off[3] = {(std::cos(theta_enc) - dir[0]) * lidar_origin_to_beam_origin_mm, (std::sin(theta_enc) - dir[1]) * lidar_origin_to_beam_origin_mm,
                             (-dir[2]) * lidar_origin_to_beam_origin_mm};
      double dir_s[3], off_s[3];
      to_sensor(rot, nullptr, dir, dir_s);
      to_sensor(rot, trans, off, off_s);
      const float d32[3] = {static_cast<float>(dir_s[0] * range_unit), static_cast<float>(dir_s[1] * range_unit), static_cast<float>(dir_s[2] * range_unit)};
      const float len = std::sqrt((d32[0] * d32[0] + d32[1] * d32[1]) + d32[2] * d32[2]);  // colwise().normalize() :369
      for (int j = 0; j < 3; j++)
      {
        dir_out[j] = d32[j] / len;
        off_out[j] = static_cast<float>(off_s[j] * range_unit);
      }
    }
  }
  return VOFOD_OK;
}

// load_mask (vofod_nodelet.cpp:506-560) after cv::imread.  `image` is the decoded w x h picture (row-major) or NULL when the
// file is missing or has the wrong size (then every ray is valid, :558).  With `mangle` (:527-541) the picture is rearranged
// into the staggered column-major order of the raw Ouster packets: pixel (u, v) lands at ((v + pixel_shift_by_row[u]) % w) * h + u.
// Product implementation: destination-major gather through the inverse shift (the oracle scatters source-major).
int vofod_mask_layout(const uint8_t* image, int32_t w, int32_t hh, const int32_t* pixel_shift_by_row, int32_t mangle, uint8_t* mask)
{
  if (w < 1 || hh < 1 || !mask)
    return VOFOD_ERR_INVALID_ARG;
  const size_t n_px = static_cast<size_t>(w) * hh;
  if (!image)
    std::memset(mask, 1, n_px);
  else if (!mangle)
    std::memcpy(mask, image, n_px);
  else
  {
    // source column of destination column 0, per ring: v = (vv - shift) mod w.  A negative shift makes the reference index
    // its vector out of range for the first columns (std::out_of_range): refused here.
    std::vector<int32_t> back(hh);
    for (int u = 0; u < hh; u++)
    {
      const int32_t sft = pixel_shift_by_row ? pixel_shift_by_row[u] : 0;
      if (sft < 0)
        return VOFOD_ERR_INVALID_ARG;
      back[u] = (w - sft % w) % w;
    }
    uint8_t* dst = mask;
    for (int32_t vv = 0; vv < w; vv++)
      for (int u = 0; u < hh; u++, dst++)
      {
        int32_t v = vv + back[u];
        if (v >= w)
          v -= w;
        *dst = image[static_cast<size_t>(u) * w + v];
      }
  }
  return VOFOD_OK;
}

// check_sensor_params (vofod_nodelet.cpp:1869-1917): the first valid pixel of an organised cloud (mask set, range != 0)
// must agree with the sensor model: direction of (point - beam offset) equal to the LUT direction within 1e-3, its length
// equal to range * 0.001 m within 1e-3, LUT direction of unit length within 1e-3.  *checked tells whether a valid pixel was
// found (m_sensor_params_checked); the return value is VOFOD_OK when the parameters fit (or nothing could be checked) and
// VOFOD_ERR_SIZE_MISMATCH - the reference's "parameters do not match the data" - otherwise.  Host arrays, w * h elements.
int vofod_check_sensor_params(const vofod_scan* scan, const float* lut_directions, const float* lut_offsets, const uint8_t* mask, int32_t* checked)
{
  if (!scan || !scan->x || !scan->y || !scan->z || !scan->range || !lut_directions || scan->memspace != VOFOD_MEM_HOST)
    return VOFOD_ERR_INVALID_ARG;
  if (checked)
    *checked = 0;
  const size_t n_px = static_cast<size_t>(scan->width) * scan->height;
  auto f32 = [&](const void* base, size_t i) {
    float v;
    std::memcpy(&v, static_cast<const char*>(base) + i * scan->stride_bytes, 4);
    return v;
  };
  auto u32 = [&](const void* base, size_t i) {
    uint32_t v;
    std::memcpy(&v, static_cast<const char*>(base) + i * scan->stride_bytes, 4);
    return v;
  };
  for (size_t idx = 0; idx < n_px; idx++)  // row-major: idx = row * width + col, rows outer as in the reference
  {
    const uint32_t range = u32(scan->range, idx);
    if ((mask && !mask[idx]) || range == 0)
      continue;
    const float* ld = lut_directions + 3 * idx;
    const float lut_dist = 0.001f * static_cast<float>(range);
    float rel[3] = {f32(scan->x, idx), f32(scan->y, idx), f32(scan->z, idx)};
    if (lut_offsets)
      for (int a = 0; a < 3; a++)
        rel[a] -= lut_offsets[3 * idx + a];
    const float pt_dist = std::sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2]);
    float diff2 = 0.0f;
    for (int a = 0; a < 3; a++)
    {
      const float d = rel[a] / pt_dist - ld[a];
      diff2 += d * d;
    }
    const float lut_norm = std::sqrt((ld[0] * ld[0] + ld[1] * ld[1]) + ld[2] * ld[2]);
    const bool ok = !(std::sqrt(diff2) > 1e-3f) && !(std::fabs(pt_dist - lut_dist) > 1e-3f) && !(1.0f - lut_norm > 1e-3f);
    if (checked)
      *checked = 1;
    return ok ? VOFOD_OK : VOFOD_ERR_SIZE_MISMATCH;
  }
  return VOFOD_OK;
}

void vofod_destroy(vofod_handle* h)
{
  if (!h)
    return;
  (void)hipSetDevice(h->device);
  if (h->stream)
    (void)hipStreamSynchronize(h->stream);
  h->ws.release();
  for (auto& w : h->wsx)
    w.release();
  h->aux.release();
  h->sepws.release();
  void* ptrs[] = {h->d_map, h->d_flags, h->d_ray, h->d_mapbits, h->d_mapclose, h->d_prof_slab, h->d_prof_ccl, h->d_counter, h->d_lut_dirs, h->d_lut_offs, h->d_mask, h->d_rows, h->d_crows, h->d_boxstage, h->d_idxstage,
                  h->sep.d_tbits, h->sep.d_tpop, h->sep.d_tprefix, h->sep.d_bsum, h->sep.d_px, h->sep.d_py, h->sep.d_pz, h->sep.d_pi, h->sep.d_sure, h->sep.d_sure_pre,
                  h->sep.d_vcnt, h->sep.d_first, h->sep.d_nsure, h->sep.d_offsets, h->sep.d_small};
  for (void* p : ptrs)
    if (p)
      (void)hipFree(p);
  std::vector<ExploreBufs*> explore_all{&h->explore};
  for (ExploreBufs& e : h->explore_slot)
    explore_all.push_back(&e);
  for (ExploreBufs* e : explore_all)
    for (void* p : {static_cast<void*>(e->d_overlay), static_cast<void*>(e->d_stack), static_cast<void*>(e->d_explored), static_cast<void*>(e->d_touched), static_cast<void*>(e->d_ovl_list),
                    static_cast<void*>(e->d_ovl_count), static_cast<void*>(e->d_job_begin), static_cast<void*>(e->d_visited), static_cast<void*>(e->d_jobs), static_cast<void*>(e->d_results),
                    static_cast<void*>(e->d_members)})
      if (p)
        (void)hipFree(p);
  for (auto& c : h->ctab)
    for (void* p : {static_cast<void*>(c.d_rows), static_cast<void*>(c.d_boffs), static_cast<void*>(c.d_sure), static_cast<void*>(c.d_amb), static_cast<void*>(c.d_pair), static_cast<void*>(c.d_lbtab)})
      if (p)
        (void)hipFree(p);
  if (h->h_counter)
    (void)hipHostFree(h->h_counter);
  if (h->d_bgcount)
    (void)hipFree(h->d_bgcount);
  if (h->h_bgcount)
    (void)hipHostFree(h->h_bgcount);
  if (h->sep.h_small)
    (void)hipHostFree(h->sep.h_small);
  for (hipEvent_t e : {h->ev_stagger, h->ev_explore, h->ev_bgcount})
    if (e)
      (void)hipEventDestroy(e);
  if (h->stream)
    (void)hipStreamDestroy(h->stream);
  if (h->stream_tail)
    (void)hipStreamDestroy(h->stream_tail);
  if (h->stream_key)
    (void)hipStreamDestroy(h->stream_key);
  for (hipStream_t st : h->stream_frames)
    if (st)
      (void)hipStreamDestroy(st);
  for (int t = 1; t < vofod_handle::MAX_INFLIGHT; t++)
    if (h->chain_stream[t])
      (void)hipStreamDestroy(h->chain_stream[t]);
  delete h;
}

int vofod_create(const vofod_static_params* sp, const vofod_dyn_params* dp, vofod_handle** out)
{
  if (!sp || !dp || !out || !(sp->voxel_size > 0) || sp->sensor_hrays < 2 || sp->sensor_vrays < 2)
    return VOFOD_ERR_INVALID_ARG;
  *out = nullptr;
  vofod_handle* h = new vofod_handle;
  h->sp = *sp;
  h->dp = *dp;
  h->device = sp->device;
  h->sp.lut_directions = nullptr;
  h->sp.lut_offsets = nullptr;
  h->sp.mask = nullptr;
  auto fail = [&](int code) {
    std::fprintf(stderr, "vofod_create: %s\n", h->err.c_str());
    vofod_destroy(h);
    return code;
  };
#define CREATE_CHK(expr)                                                        \
  do                                                                            \
  {                                                                             \
    const hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess)                                                       \
    {                                                                           \
      h->err = std::string(#expr) + ": " + hipGetErrorString(e_);               \
      return fail(VOFOD_ERR_DEVICE);                                            \
    }                                                                           \
  } while (0)
  CREATE_CHK(hipSetDevice(h->device));
  CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  {
    // the tail's one small kernel (k_explore) has the host waiting for it: highest priority, so that it is dispatched at the
    // next kernel boundary of the batches in flight instead of behind their queued kernels
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    int prio_tail = prio_hi;
    if (const char* e = std::getenv("VOFOD_TAIL_PRIO"))  // (diagnostics) hi | mid | lo
      prio_tail = e[0] == 'l' ? prio_lo : e[0] == 'm' ? (prio_lo + prio_hi) / 2 : prio_hi;
    CREATE_CHK(hipStreamCreateWithPriority(&h->stream_tail, hipStreamNonBlocking, prio_tail));
    // staged pipeline of submitted batches (process_frames): streaming kernels below the frame kernels
    CREATE_CHK(hipStreamCreateWithPriority(&h->stream_key, hipStreamNonBlocking, prio_lo));
    // (VOFOD_FRAME_STREAMS: diagnostics - the number of frame streams, 1..8)
    h->n_frame_streams = 2;  // (more streams cost more than they bring: 811 k / 763 k / 656 k frames/s with 2 / 4 / 8 of them, 32-frame batches 435 k / 268 k / 216 k)
    if (const char* e = std::getenv("VOFOD_FRAME_STREAMS"))
      h->n_frame_streams = std::min(std::max(std::atoi(e), 1), static_cast<int>(vofod_handle::MAX_FRAME_STREAMS));
    for (int i = 0; i < h->n_frame_streams; i++)
      CREATE_CHK(hipStreamCreateWithPriority(&h->stream_frames[i], hipStreamNonBlocking, (prio_lo + prio_hi) / 2));
    h->stream_frame = h->stream_frames[0];
  }
  h->chain_stream[0] = h->stream;
  // (tickets 1-3 have their streams from the start; tickets 4-7 - only small batches gain from more than four in flight - get
  // theirs when they are first taken: streams that merely exist are not free)
  for (int t = 1; t < 4; t++)
    CREATE_CHK(hipStreamCreateWithFlags(&h->chain_stream[t], hipStreamNonBlocking));
  for (int a = 0; a < 3; a++)
  {
    h->exclude_center[a] = sp->exclude_offset[a];
    h->oparea_center[a] = sp->oparea_offset[a];
  }
  h->exclude_center[2] = sp->exclude_offset[2] + sp->exclude_size[2] / 2.0f;  // vofod_nodelet.cpp:204
  h->oparea_center[2] = sp->oparea_offset[2] + sp->oparea_size[2] / 2.0f;     // :212
  const float n_voxels_xy = sp->oparea_size[0] / sp->voxel_size * sp->oparea_size[1] / sp->voxel_size;  // :229
  h->background_min_sufficient_pts = static_cast<uint64_t>(n_voxels_xy * sp->background_sufficient_points_ratio);
  // VoxelMap::resize voxel_map.cpp:11-48
  const float inv = 1.0f / sp->voxel_size;
  int sizes[3];
  for (int a = 0; a < 3; a++)
  {
    h->mg.off[a] = h->oparea_center[a] - sp->oparea_size[a] / 2.0f;
    sizes[a] = static_cast<int>(std::ceil(inv * sp->oparea_size[a])) + 1;
    h->hg.off[a] = h->mg.off[a];
    h->hg.s[a] = sizes[a];
  }
  h->mg.vs = h->hg.vs = sp->voxel_size;
  h->mg.vs_inv = h->hg.vs_inv = 1.0f / sp->voxel_size;
  h->mg.sx = sizes[0];
  h->mg.sy = sizes[1];
  h->mg.sz = sizes[2];
  h->mg.n = static_cast<uint64_t>(sizes[0]) * sizes[1] * sizes[2];
  const size_t M = h->mg.n;
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_map), M * sizeof(float)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_flags), M * sizeof(float)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_ray), M * sizeof(float)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_mapbits), ((M + 63) / 64 + 2) * sizeof(unsigned long long)));
  CREATE_CHK(hipMemset(h->d_mapbits, 0, ((M + 63) / 64 + 2) * sizeof(unsigned long long)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_mapclose), ((M + 63) / 64 + 2) * sizeof(unsigned long long)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_counter), 8 * sizeof(unsigned long long)));
  CREATE_CHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_counter), 8 * sizeof(unsigned long long)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_bgcount), 8 * MB_SLOTS * sizeof(unsigned long long)));
  CREATE_CHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_bgcount), 8 * MB_SLOTS * sizeof(unsigned long long)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_rows), MAX_STENCIL_ROWS * sizeof(StencilRow)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_crows), MAX_STENCIL_ROWS * sizeof(CloseRow)));
  // sensor
  const size_t n = static_cast<size_t>(sp->sensor_hrays) * sp->sensor_vrays;
  std::vector<float> dirs(3 * n), offs(3 * n, 0.0f);
  if (sp->lut_directions)
    std::copy(sp->lut_directions, sp->lut_directions + 3 * n, dirs.begin());
  else
    vofod_sim_lut(sp->sensor_hrays, sp->sensor_vrays, sp->sensor_vfov, dirs.data());
  if (sp->lut_offsets)
    std::copy(sp->lut_offsets, sp->lut_offsets + 3 * n, offs.begin());
  std::vector<uint8_t> mask(n, 1);
  if (sp->mask)
    std::copy(sp->mask, sp->mask + n, mask.begin());
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_lut_dirs), 3 * n * sizeof(float)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_lut_offs), 3 * n * sizeof(float)));
  CREATE_CHK(hipMalloc(reinterpret_cast<void**>(&h->d_mask), n));
  CREATE_CHK(hipMemcpy(h->d_lut_dirs, dirs.data(), 3 * n * sizeof(float), hipMemcpyHostToDevice));
  CREATE_CHK(hipMemcpy(h->d_lut_offs, offs.data(), 3 * n * sizeof(float), hipMemcpyHostToDevice));
  CREATE_CHK(hipMemcpy(h->d_mask, mask.data(), n, hipMemcpyHostToDevice));
  // per-frame workspace: the crops bound the voxel-grid lattice by the map lattice plus one cell per side
  const uint64_t cells = static_cast<uint64_t>(sizes[0] + 2) * (sizes[1] + 2) * (sizes[2] + 2);
  if (cells > 0x7fffffffull)
  {
    h->err = "voxel map too fine for 32-bit voxel keys";
    return fail(VOFOD_ERR_INDEX_OVERFLOW);
  }
  const uint32_t F = static_cast<uint32_t>(std::max(sp->max_batch_frames, 1));
  const uint32_t nbricks = static_cast<uint32_t>(((sizes[0] + 2 + 3) / 4) * ((sizes[1] + 2 + 3) / 4) * static_cast<uint64_t>((sizes[2] + 2 + 3) / 4));
  CREATE_CHK(h->ws.ensure(F, static_cast<uint32_t>(n), static_cast<uint32_t>(n), static_cast<uint32_t>((cells + 63) / 64), nbricks));
#undef CREATE_CHK
  {
    // host workers for the per-frame tail of a batch (frames are independent); VOFOD_THREADS overrides
    unsigned nt = std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
    if (const char* e = std::getenv("VOFOD_THREADS"))
      nt = std::max(1, std::atoi(e));
    if (F == 1)
      nt = 1;
    h->pool.reset(new vt::Pool(nt));
  }
  if (do_reset(h) != VOFOD_OK)
    return fail(VOFOD_ERR_DEVICE);
  h->sure_background_sufficient = false;  // :283-284
  h->background_pts_sufficient = false;
  h->last_detection_id = 0;  // :296
  (void)hipDeviceSynchronize();  // null-stream memsets of the set-up are complete before any non-blocking stream runs
  *out = h;
  return VOFOD_OK;
}

// A submitted batch reads the voxel map, its images and (ticket 0) the synchronous workspace until it is collected.
// Calls that would overwrite that state are refused instead of racing it (the reference serialises them on m_voxels_mtx;
// a batch in flight is "inside the lock" until vofod_batch_collect).
static int busy_check(vofod_handle* h, bool uses_ws0, bool writes_map)
{
  if (uses_ws0 && h->ws.pending)
  {
    h->err = "ticket 0 is in flight on the synchronous workspace: collect it first";
    return VOFOD_ERR_BUSY;
  }
  if (writes_map)
    for (int t = 0; t < vofod_handle::MAX_INFLIGHT; t++)
      if (h->slot(t)->pending)
      {
        h->err = "a submitted batch still reads the voxel map: collect it first";
        return VOFOD_ERR_BUSY;
      }
  return VOFOD_OK;
}

int vofod_reset(vofod_handle* h)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, true); b != VOFOD_OK)
    return b;
  const int r = do_reset(h);
  h->sure_background_sufficient = false;
  h->background_pts_sufficient = false;
  h->last_detection_id = 0;
  h->cf_off = false;
  return r;
}

int vofod_set_dynamic_params(vofod_handle* h, const vofod_dyn_params* dp)
{
  if (!h || !dp)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  h->dp = *dp;
  return VOFOD_OK;
}

const char* vofod_last_error_string(vofod_handle* h) { return h ? h->err.c_str() : "null handle"; }

int vofod_get_status(vofod_handle* h, vofod_status_info* out)
{
  if (!h || !out)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  out->detection_its = h->detection_its;
  out->last_detection_id = h->last_detection_id;
  out->background_pts_sufficient = h->background_pts_sufficient;
  out->sure_background_sufficient = h->sure_background_sufficient;
  out->raycast_pending = h->raycast_pending;
  out->map_size[0] = h->mg.sx;
  out->map_size[1] = h->mg.sy;
  out->map_size[2] = h->mg.sz;
  for (int a = 0; a < 3; a++)
    out->map_offset[a] = h->mg.off[a];
  return VOFOD_OK;
}

int vofod_load_apriori(vofod_handle* h, const float* xyz, size_t n)
{
  if (!h || (!xyz && n))
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, true); b != VOFOD_OK)
    return b;
  if (n)
  {
    float* d = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&d), n * 3 * sizeof(float)));
    HIPCHK(hipMemcpy(d, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    KLAUNCH(h, k_apriori, dim3((n + 255) / 256), dim3(256), h->d_map, h->mg, d, static_cast<uint32_t>(n));
    HIPCHK(hipStreamSynchronize(h->stream));
    (void)hipFree(d);
  }
  h->sure_background_sufficient = true;  // vofod_nodelet.cpp:343-344
  h->background_pts_sufficient = true;
  h->mapbits_valid = false;
  return VOFOD_OK;
}


static float* pick_map(vofod_handle* h, int which)
{
  switch (which)
  {
    case VOFOD_MAP_VOXELS: return h->d_map;
    case VOFOD_MAP_FLAGS: return h->d_flags;
    case VOFOD_MAP_RAYCAST: return h->d_ray;
  }
  return nullptr;
}

int vofod_read_map(vofod_handle* h, int which, float* dst, size_t n)
{
  if (!h || !dst)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  float* m = pick_map(h, which);
  if (!m || n != h->mg.n)
    return VOFOD_ERR_SIZE_MISMATCH;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(dst, m, n * sizeof(float), hipMemcpyDeviceToHost));
  return VOFOD_OK;
}

int vofod_voxels_as_pc(vofod_handle* h, int which, float threshold, int greater_than, vofod_point_xyzi* out, size_t cap, size_t* n_out)
{
  if (!h || !n_out || (cap && !out))
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  const float* m = pick_map(h, which);
  if (!m)
    return VOFOD_ERR_INVALID_ARG;
  *n_out = 0;
  vr::SepState& s = h->sep;
  const uint32_t ncol = static_cast<uint32_t>(h->mg.sx) * h->mg.sy;
  int r;
  if ((r = sep_ensure_words(h, ncol + 2)) != VOFOD_OK || (r = sep_ensure_pts(h, std::max<size_t>(1 << 16, ncol))) != VOFOD_OK)
    return r;
  KLAUNCH(h, vr::k_col_count_thr, dim3((ncol + 255) / 256), dim3(256), h->mg, m, threshold, greater_than, s.d_tpop);
  if ((r = gscan(h, s.d_tpop, ncol, s.d_tprefix, s.d_bsum, s.d_small)) != VOFOD_OK)
    return r;
  HIPCHK(hipMemcpyAsync(s.h_small, s.d_small, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const uint32_t P = s.h_small[0];
  *n_out = P;
  if (P > cap)
    return VOFOD_ERR_CAPACITY;  // n_out holds the required size
  if (P == 0)
    return VOFOD_OK;
  if ((r = ensure_boxstage(h, static_cast<size_t>(P) * 4)) != VOFOD_OK)
    return r;
  KLAUNCH(h, vr::k_col_emit_xyzi, dim3((ncol + 255) / 256), dim3(256), h->mg, m, threshold, greater_than, s.d_tprefix, P, reinterpret_cast<float4*>(h->d_boxstage));
  HIPCHK(hipMemcpyAsync(out, h->d_boxstage, sizeof(vofod_point_xyzi) * P, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VOFOD_OK;
}

int vofod_update_ground(vofod_handle* h, float range, float min_range, float max_range, const float tf[12])
{
  if (!h || !tf)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, false, true); b != VOFOD_OK)
    return b;
  if (range <= min_range && range >= max_range)  // vofod_nodelet.cpp:585, as written
    return VOFOD_OK;
  // one voxel at the range-finder's rate: read, blend on the host in the reference's double expression, write back
  const float p[3] = {tf[0] * range + tf[3], tf[4] * range + tf[7], tf[8] * range + tf[11]};  // :597
  int idx[3];
  h->hg.coordToIdx(p, idx);
  if (!h->hg.inLimits(idx))  // :601-605
    return VOFOD_ERR_MAP_RANGE;
  float* cell = h->d_map + h->hg.lin(idx);
  float m = 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(&m, cell, sizeof(float), hipMemcpyDeviceToHost));
  m = static_cast<float>((static_cast<double>(m) + h->dp.voxel_map__scores__point) / 2.0);  // :609
  HIPCHK(hipMemcpy(cell, &m, sizeof(float), hipMemcpyHostToDevice));
  h->mapbits_valid = false;
  return VOFOD_OK;
}

int vofod_write_map(vofod_handle* h, int which, const float* src, size_t n)
{
  if (!h || !src)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, false, true); b != VOFOD_OK)
    return b;
  float* m = pick_map(h, which);
  if (!m || n != h->mg.n)
    return VOFOD_ERR_SIZE_MISMATCH;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(m, src, n * sizeof(float), hipMemcpyHostToDevice));
  h->mapbits_valid = false;
  return VOFOD_OK;
}

int vofod_process_scan(vofod_handle* h, const vofod_scan* scan, const float tf[12], int flags, vofod_detection* out, size_t cap, size_t* n_out, vofod_scan_debug* dbg)
{
  if (!h || !scan || !tf || !n_out)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, !(flags & VOFOD_SCAN_NO_MAP_UPDATE)); b != VOFOD_OK)
    return b;
  *n_out = 0;
  int r = process_frames(h, h->ws, FRAMES_SYNC, scan, tf, 1, flags, out, cap, nullptr, n_out, dbg);
  // (a cold map: more far voxels than the close-first path of a single scan takes - nothing of the scan was applied; once more,
  // through the full clustering)
  if (r == CCL_RETRY_STATUS)
    r = process_frames(h, h->ws, FRAMES_SYNC, scan, tf, 1, flags, out, cap, nullptr, n_out, dbg);
  return r;
}

int vofod_process_batch(vofod_handle* h, const vofod_scan* scans, const float* tfs, size_t n, vofod_detection* out, size_t cap, uint32_t* n_out_per_frame, size_t* n_out,
                        vofod_scan_debug* dbg)
{
  if (!h || !scans || !tfs || !n_out)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, false); b != VOFOD_OK)
    return b;
  *n_out = 0;
  int ret = VOFOD_OK;
  size_t total = 0;
  for (size_t base = 0; base < n; base += h->ws.F)  // larger batches run as successive launch groups
  {
    const uint32_t m = static_cast<uint32_t>(std::min<size_t>(h->ws.F, n - base));
    size_t got = 0;
    int r = process_frames(h, h->ws, FRAMES_SYNC, scans + base, tfs + 12 * base, m, VOFOD_SCAN_NO_MAP_UPDATE, out ? out + total : nullptr, total < cap ? cap - total : 0,
                           n_out_per_frame ? n_out_per_frame + base : nullptr, &got, dbg ? dbg + base : nullptr);
    // a frame beyond the capacities of the frame kernel: this batch once more - with the full clustering when the close-first
    // kernel gave up (a cold map), on the global kernels when the LDS image did; the two can follow each other
    for (int attempt = 0; attempt < 2 && r == CCL_RETRY_STATUS; attempt++)
      r = process_frames(h, h->ws, FRAMES_SYNC, scans + base, tfs + 12 * base, m, VOFOD_SCAN_NO_MAP_UPDATE, out ? out + total : nullptr, total < cap ? cap - total : 0,
                         n_out_per_frame ? n_out_per_frame + base : nullptr, &got, dbg ? dbg + base : nullptr);
    for (size_t i = total; i < std::min(total + got, cap); i++)
      out[i].frame += static_cast<uint32_t>(base);
    total += got;
    if (r != VOFOD_OK)
      ret = r;
  }
  *n_out = total;
  return ret;
}

int vofod_batch_submit(vofod_handle* h, const vofod_scan* scans, const float* tfs, size_t n, int* ticket)
{
  if (!h || !scans || !tfs || !ticket)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (n > h->ws.F)
  {
    h->err = "batch larger than max_batch_frames";
    return VOFOD_ERR_CAPACITY;
  }
  int t = -1;
  for (int i = 0; i < vofod_handle::MAX_INFLIGHT && t < 0; i++)
    if (!h->slot(i)->pending)
      t = i;
  if (t < 0)
  {
    h->err = "eight batches already in flight: collect one first";
    return VOFOD_ERR_CAPACITY;
  }
  if (t >= 4)
  {
    // more than four chains in flight: they only run side by side when the runtime may use more than its default of four
    // hardware queues (a process-wide setting read when the runtime initialises: INTEGRATION.md) - say so once
    static std::once_flag warned;
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    if (!q || std::atoi(q) < 8)
      std::call_once(warned, [] {
        std::fprintf(stderr, "[vofod] more than four batches in flight but GPU_MAX_HW_QUEUES is %s: their streams share four hardware queues and take turns "
                             "(32-frame batches: ~140 k instead of ~230 k frames/s); export GPU_MAX_HW_QUEUES=16 before the process starts\n",
                     std::getenv("GPU_MAX_HW_QUEUES") ? std::getenv("GPU_MAX_HW_QUEUES") : "unset");
      });
  }
  Workspace* w = h->slot(t);
  if (w->F == 0)
  {
    if (hipError_t e = w->ensure(h->ws.F, h->ws.pt_cap, h->ws.vox_cap, h->ws.words_cap, h->ws.bricks_cap); e != hipSuccess)
    {
      h->err = std::string("extra workspace: ") + hipGetErrorString(e);
      return VOFOD_ERR_DEVICE;
    }
  }
  const int r = process_frames(h, *w, FRAMES_LAUNCH, scans, tfs, static_cast<uint32_t>(n), VOFOD_SCAN_NO_MAP_UPDATE, nullptr, 0, nullptr, nullptr, nullptr);
  if (r == VOFOD_OK)
    *ticket = t;
  return r;
}

int vofod_reserve(vofod_handle* h, int tickets)
{
  if (!h || tickets < 1 || tickets > vofod_handle::MAX_INFLIGHT)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  for (int t = 0; t < tickets; t++)
  {
    Workspace* w = h->slot(t);
    if (w->F == 0)
      if (hipError_t e = w->ensure(h->ws.F, h->ws.pt_cap, h->ws.vox_cap, h->ws.words_cap, h->ws.bricks_cap); e != hipSuccess)
      {
        h->err = std::string("extra workspace: ") + hipGetErrorString(e);
        return VOFOD_ERR_DEVICE;
      }
    if (t > 0 && !h->chain_stream[t])
      HIPCHK(hipStreamCreateWithFlags(&h->chain_stream[t], hipStreamNonBlocking));
  }
  // the device tail's flood-fill buffers: shared by the batches of 128 frames and more (their tails take turns on the tail
  // stream), per ticket for smaller batches (process_frames sizes those by the batch; reserved here for full workspaces)
  if (const int r = ensure_explore(h, h->explore, h->ws.F, static_cast<size_t>(h->ws.F) * vtd::TP_MAXC, static_cast<size_t>(h->ws.F) * vtd::TP_MAXM); r != VOFOD_OK)
    return r;
  if (h->ws.F < 128u)
    for (int t = 0; t < tickets; t++)
      if (const int r = ensure_explore(h, h->explore_slot[t], h->ws.F, static_cast<size_t>(h->ws.F) * vtd::TP_MAXC, static_cast<size_t>(h->ws.F) * vtd::TP_MAXM); r != VOFOD_OK)
        return r;
  return VOFOD_OK;
}

int vofod_batch_collect(vofod_handle* h, int ticket, vofod_detection* out, size_t cap, uint32_t* n_out_per_frame, size_t* n_out)
{
  if (!h || !n_out || ticket < 0 || ticket >= vofod_handle::MAX_INFLIGHT)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  Workspace& w = *h->slot(ticket);
  if (!w.pending)
    return VOFOD_ERR_NOT_PENDING;
  *n_out = 0;
  int r = process_frames(h, w, FRAMES_COLLECT, nullptr, nullptr, 0, VOFOD_SCAN_NO_MAP_UPDATE, out, cap, n_out_per_frame, n_out, nullptr);
  for (int attempt = 0; attempt < 2 && r == CCL_RETRY_STATUS; attempt++)
  {
    // a frame beyond the capacities of the frame kernel: the batch is enqueued again from the submitted descriptors - with the
    // full clustering when the close-first kernel gave up (a cold map), with the global-memory clustering when the LDS image did
    r = process_frames(h, w, FRAMES_LAUNCH, w.job_scans.data(), w.job_tfs.data(), w.job_n, VOFOD_SCAN_NO_MAP_UPDATE, nullptr, 0, nullptr, nullptr, nullptr);
    if (r == VOFOD_OK)
      r = process_frames(h, w, FRAMES_COLLECT, nullptr, nullptr, 0, VOFOD_SCAN_NO_MAP_UPDATE, out, cap, n_out_per_frame, n_out, nullptr);
  }
  return r;
}

int vofod_raycast_begin(vofod_handle* h, const vofod_scan* scan, const float tf[12])
{
  if (!h || !tf)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, false); b != VOFOD_OK)
    return b;
  return raycast_begin_locked(h, scan, tf);
}

int vofod_raycast_finish(vofod_handle* h)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, false, true); b != VOFOD_OK)
    return b;
  return raycast_finish_locked(h);
}

int vofod_sepclusters_begin(vofod_handle* h, int* sure)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, false); b != VOFOD_OK)
    return b;
  return sepclusters_begin_locked(h, sure);
}

int vofod_sepclusters_finish(vofod_handle* h)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, true); b != VOFOD_OK)
    return b;
  return sepclusters_finish_locked(h);
}

int vofod_voxel_grid_weighted(vofod_handle* h, const vofod_cloud_view* in, float leaf, int align, const float align_center[3], vofod_point_xyzr* out, uint32_t* keys,
                              size_t cap, size_t* n_out, vofod_grid_desc* grid)
{
  if (!h || !in || !(leaf > 0) || (align && !align_center) || in->n > 0x7fffffffu)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, false); b != VOFOD_OK)
    return b;
  GridParams g;
  FrameHdr hdr{};
  const float l[3] = {leaf, leaf, leaf};
  if (in->n == 0)
  {
    if (n_out)
      *n_out = 0;
    if (grid)
      *grid = vofod_grid_desc{{leaf, leaf, leaf}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    return VOFOD_OK;
  }
  const int r = voxelize_cloud(h, h->aux, in, g, l, align != 0, align_center, hdr);
  if (r != VOFOD_OK)
  {
    if (n_out)
      *n_out = 0;
    return r;
  }
  return copy_grid_out(h, h->aux, g, hdr, out, keys, cap, n_out, grid);
}

int vofod_voxel_grid_counted(vofod_handle* h, const vofod_cloud_view* in, float leaf, float threshold, vofod_point_xyzr* out, uint32_t* keys, size_t cap, size_t* n_out,
                             vofod_grid_desc* grid)
{
  if (!h || !in || !(leaf > 0) || !in->intensity || in->n > 0x7fffffffu)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, false); b != VOFOD_OK)
    return b;
  GridParams g;
  FrameHdr hdr{};
  const float l[3] = {leaf, leaf, leaf};
  if (in->n == 0)
  {
    if (n_out)
      *n_out = 0;
    return VOFOD_OK;
  }
  int r = voxelize_cloud(h, h->aux, in, g, l, false, nullptr, hdr);
  if (r != VOFOD_OK)
  {
    if (n_out)
      *n_out = 0;
    return r;
  }
  const uint32_t P = static_cast<uint32_t>(in->n);
  if ((r = sep_ensure_words(h, 1)) != VOFOD_OK || (r = sep_ensure_pts(h, P)) != VOFOD_OK)
    return r;
  const FrameArgs& a = h->aux.h_args[0];
  KLAUNCH(h, vr::k_flag_over, dim3((P + 255) / 256), dim3(256), a.intensity, a.stride, P, threshold, h->sep.d_sure);
  if ((r = counted_tail(h, h->aux, hdr.V, P, h->sep.d_sure)) != VOFOD_OK)
    return r;
  HIPCHK(hipStreamSynchronize(h->stream));
  return copy_grid_out(h, h->aux, g, hdr, out, keys, cap, n_out, grid);
}

int vofod_cluster(vofod_handle* h, const vofod_point_xyzr* pts, const uint32_t* keys, const vofod_grid_desc* grid, size_t n, float tolerance, uint32_t* labels,
                  size_t* n_clusters)
{
  if (!h || (!pts && n) || !labels || !keys || !grid || !(tolerance > 0) || n > 0x7fffffffu)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  if (const int b = busy_check(h, true, false); b != VOFOD_OK)
    return b;
  if (n_clusters)
    *n_clusters = 0;
  if (n == 0)
    return VOFOD_OK;
  const uint64_t cells = static_cast<uint64_t>(grid->div_b[0]) * grid->div_b[1] * grid->div_b[2];
  if (grid->div_b[0] <= 0 || grid->div_b[1] <= 0 || grid->div_b[2] <= 0 || cells > 0x7fffffffull)
    return VOFOD_ERR_INVALID_ARG;
  for (size_t i = 0; i < n; i++)
    if (keys[i] >= cells || (i && keys[i] <= keys[i - 1]))
    {
      h->err = "vofod_cluster: keys must be strictly ascending lattice keys";
      return VOFOD_ERR_INVALID_ARG;
    }
  Workspace& ws = h->aux;
  const uint32_t words = static_cast<uint32_t>((cells + 63) / 64);
  const uint32_t need_bricks = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>((grid->div_b[0] + 3) / 4) * ((grid->div_b[1] + 3) / 4) * ((grid->div_b[2] + 3) / 4), 1u << 28));
  if (hipError_t e = ws.ensure(1, static_cast<uint32_t>(n), static_cast<uint32_t>(n), words + 64, need_bricks); e != hipSuccess)
  {
    h->err = std::string("workspace allocation: ") + hipGetErrorString(e);
    return VOFOD_ERR_DEVICE;
  }
  // rebuild the frame state the clustering kernels consume: header, bitmap, word prefix, voxel arrays
  GridParams g;
  const float zero[3] = {0, 0, 0};
  fill_grid_params(h, g, grid->leaf, false, zero, ws);
  FrameHdr hdr{};
  hdr.n_in = static_cast<uint32_t>(n);
  hdr.status = VOFOD_OK;
  for (int a = 0; a < 3; a++)
  {
    hdr.offset[a] = grid->offset[a];
    hdr.min_b[a] = grid->min_b[a];
    hdr.div_b[a] = grid->div_b[a];
  }
  hdr.n_cells = static_cast<uint32_t>(cells);
  hdr.n_words = words;
  hdr.V = static_cast<uint32_t>(n);
  std::vector<unsigned long long> bm(words + 2, 0ull);
  for (size_t i = 0; i < n; i++)
    bm[keys[i] >> 6] |= 1ull << (keys[i] & 63);
  std::vector<uint32_t> prefix(words + 2, 0u), ident(n);
  uint32_t run = 0;
  for (uint32_t w = 0; w < words; w++)
  {
    prefix[w] = run;
    run += __builtin_popcountll(bm[w]);
  }
  for (size_t i = 0; i < n; i++)
    ident[i] = static_cast<uint32_t>(i);
  std::vector<int32_t> cbox(6 * n);
  for (size_t i = 0; i < n; i++)
    for (int c = 0; c < 3; c++)
    {
      cbox[6 * i + c] = 0x7fffffff;
      cbox[6 * i + 3 + c] = static_cast<int32_t>(0x80000000u);
    }
  HIPCHK(hipMemcpyAsync(ws.d_hdrs, &hdr, sizeof(hdr), hipMemcpyHostToDevice, h->stream));
  ws.bitmap_clean = false;
  HIPCHK(hipMemcpyAsync(ws.d_bitmaps, bm.data(), sizeof(unsigned long long) * (words + 2), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(ws.d_wprefix, prefix.data(), sizeof(uint32_t) * (words + 2), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(ws.va.pts, pts, sizeof(float4) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(ws.va.key, keys, sizeof(uint32_t) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(ws.va.parent, ident.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(ws.va.csize, 0, sizeof(uint32_t) * n, h->stream));
  HIPCHK(hipMemsetAsync(ws.va.cclose, 0, sizeof(uint32_t) * n, h->stream));
  HIPCHK(hipMemcpyAsync(ws.va.cbox, cbox.data(), sizeof(int32_t) * 6 * n, hipMemcpyHostToDevice, h->stream));
  float cmax = 0;
  for (int a = 0; a < 3; a++)
    cmax = std::max({cmax, std::fabs(grid->offset[a]), std::fabs(grid->offset[a] + grid->leaf[a] * (grid->div_b[a] + 1))});
  const int r = launch_cluster(h, ws, g, 1, tolerance, cmax);
  if (r != VOFOD_OK)
    return r;
  HIPCHK(hipMemcpyAsync(labels, ws.d_labels, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (n_clusters)
  {
    size_t c = 0;
    for (size_t i = 0; i < n; i++)
      c += labels[i] == i;
    *n_clusters = c;
  }
  return VOFOD_OK;
}

int vofod_ingest_apriori(vofod_handle* h, const char* filename, const float tf_xyz[3], double yaw_deg, const float sim_correction[3], size_t* n_loaded, size_t* n_voxels)
{
  if (!h || !filename || !tf_xyz || !sim_correction)
    return VOFOD_ERR_INVALID_ARG;
  size_t n = 0;
  int r = vofod_load_cloud(filename, nullptr, 0, &n);
  if (r != VOFOD_OK && r != VOFOD_ERR_CAPACITY)
    return r;
  std::vector<float> xyz(3 * n), cent;
  if (n && (r = vofod_load_cloud(filename, xyz.data(), n, &n)) != VOFOD_OK)
    return r;
  vt::ingest_apriori_points(xyz, tf_xyz, yaw_deg, sim_correction, h->sp.voxel_size, cent);  // init-time host work, as in the reference
  if (n_loaded)
    *n_loaded = n;
  if (n_voxels)
    *n_voxels = cent.size() / 3;
  return vofod_load_apriori(h, cent.data(), cent.size() / 3);
}

// ---- row N3: the nodelet's outgoing messages in the ROS 1 wire format (little endian, strings and arrays prefixed with a
// uint32 length; std_msgs/Header = seq, stamp.sec, stamp.nsec, frame_id).  A ROS-free host (examples/vofod_replay.cpp)
// writes exactly the bytes a subscriber of the reference's topics receives.

// vofod/Detections (msgs/Detections.msg, msgs/Detection.msg:1-12; filled at vofod_nodelet.cpp:968-988)
int vofod_serialize_detections(const vofod_msg_header* header, const vofod_detection* dets, size_t n, uint8_t* buf, size_t cap, size_t* n_bytes)
{
  if (!header || (n && !dets) || !n_bytes)
    return VOFOD_ERR_INVALID_ARG;
  WireOut w{buf, cap};
  w.header(header);
  w.put(static_cast<uint32_t>(n));
  for (size_t i = 0; i < n; i++)
  {
    const vofod_detection& d = dets[i];
    w.put(d.id);
    w.put(d.confidence);
    w.put(d.n_points);
    for (int a = 0; a < 3; a++)
      w.put(d.position[a]);  // geometry_msgs/Point
    for (int a = 0; a < 9; a++)
      w.put(d.covariance[a]);
    w.put(d.detection_probability);
  }
  *n_bytes = w.n;
  return w.n > cap ? VOFOD_ERR_CAPACITY : VOFOD_OK;
}

// vofod/Status (msgs/Status.msg; vofod_nodelet.cpp:1379-1385)
int vofod_serialize_status(const vofod_msg_header* header, int detection_enabled, int detection_active, uint8_t* buf, size_t cap, size_t* n_bytes)
{
  if (!header || !n_bytes)
    return VOFOD_ERR_INVALID_ARG;
  WireOut w{buf, cap};
  w.header(header);
  w.put(static_cast<uint8_t>(detection_enabled ? 1 : 0));
  w.put(static_cast<uint8_t>(detection_active ? 1 : 0));
  *n_bytes = w.n;
  return w.n > cap ? VOFOD_ERR_CAPACITY : VOFOD_OK;
}

// vofod/ProfilingInfo (msgs/ProfilingInfo.msg; vofod_nodelet.cpp:2178-2201)
int vofod_serialize_profiling_info(uint32_t stamp_sec, uint32_t stamp_nsec, uint32_t routine_id, uint64_t event_sequence, uint8_t event_type, uint8_t* buf, size_t cap, size_t* n_bytes)
{
  if (!n_bytes)
    return VOFOD_ERR_INVALID_ARG;
  WireOut w{buf, cap};
  w.put(stamp_sec);
  w.put(stamp_nsec);
  w.put(routine_id);
  w.put(event_sequence);
  w.put(event_type);
  *n_bytes = w.n;
  return w.n > cap ? VOFOD_ERR_CAPACITY : VOFOD_OK;
}

int vofod_profile_enable(vofod_handle* h, int on)
{
  if (!h)
    return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (auto& r : h->prof.recs)
  {
    h->prof.pool.push_back(r.a);
    h->prof.pool.push_back(r.b);
  }
  h->prof.recs.clear();
  h->prof.on = on != 0;
  return VOFOD_OK;
}

size_t vofod_profile_read(vofod_handle* h, char* names, double* ms, uint64_t* calls, size_t cap)
{
  if (!h || !names || !ms || !calls)
    return 0;
  std::scoped_lock lck(h->mtx);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  std::vector<std::string> order;
  std::map<std::string, std::pair<double, uint64_t>> acc;
  for (auto& r : h->prof.recs)
  {
    float t = 0;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    std::string nm = r.name;
    const size_t c = nm.rfind(':');
    if (c != std::string::npos)
      nm = nm.substr(c + 1);
    if (!acc.count(nm))
      order.push_back(nm);
    acc[nm].first += t;
    acc[nm].second += 1;
    h->prof.pool.push_back(r.a);
    h->prof.pool.push_back(r.b);
  }
  h->prof.recs.clear();
  size_t n = 0;
  for (const auto& nm : order)
  {
    if (n >= cap)
      break;
    std::memset(names + 64 * n, 0, 64);
    std::strncpy(names + 64 * n, nm.c_str(), 63);
    ms[n] = acc[nm].first;
    calls[n] = acc[nm].second;
    n++;
  }
  return n;
}

// load_cloud pc_loader.cpp:17-90 (host I/O helper, no device work)
int vofod_load_cloud(const char* filename, float* xyz, size_t cap, size_t* n_out)
{
  if (!filename || !n_out)
    return VOFOD_ERR_INVALID_ARG;
  std::ifstream fs(filename, std::ios::binary);
  if (!fs.is_open() || fs.fail())
    return VOFOD_ERR_INVALID_ARG;
  const std::string name(filename);
  const bool pts_file = name.size() >= 4 && name.compare(name.size() - 4, 4, ".pts") == 0;
  std::string line;
  if (pts_file)
    std::getline(fs, line);  // first line of a .pts file is the point count (:36-41)
  size_t n = 0;
  while (std::getline(fs, line))
  {
    // tokens are maximal runs of characters other than tab, CR, space (:61-63); blank lines are skipped (:56)
    const char* s = line.c_str();
    const char* tok[3];
    int nt = 0;
    size_t i = 0;
    const size_t L = line.size();
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; };
    while (i < L && nt < 3)
    {
      while (i < L && is_ws(s[i]))
        i++;
      if (i >= L)
        break;
      tok[nt++] = s + i;
      while (i < L && !(s[i] == ' ' || s[i] == '\t' || s[i] == '\r'))
        i++;
    }
    if (nt < 3)
      continue;  // :65-69
    if (xyz && n < cap)
      for (int c = 0; c < 3; c++)
        xyz[3 * n + c] = static_cast<float>(std::atof(tok[c]));
    n++;
  }
  *n_out = n;
  return n > cap ? VOFOD_ERR_CAPACITY : VOFOD_OK;
}

}  // extern "C"
