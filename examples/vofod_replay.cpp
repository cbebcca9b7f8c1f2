// vofod_replay — ROS-free host driver with the nodelet's three thread roles.
//
// What vofod_nodelet.cpp does around the hot path, with nothing but the C-ABI of include/vofod.h:
//   main loop      = cloud_callback / processMsg (:878-986): one vofod_process_scan per organised cloud, detections out;
//   raycast thread = raycast_cloud (:1392-1605), spawned per scan unless one is still running (:951-957): begin,
//                    wait for a detection iteration (condition variable, 800 ms timeout :1530-1537), finish;
//   sepclusters    = the timer callback updateSeparatedBGClusters (:1124-1276), every `sepclusters_period`.
// The clouds come from a small synthetic world (ground plane, two static boxes, one flying box on a circle) ray-cast
// through the simulated sensor LUT (initialize_sensor_lut_simulation :374-420 via vofod_sim_lut), exactly the shape
// the nodelet receives from the Ouster driver: organised h x w xyz + range in mm, zeros for no return.
//
//   vofod_replay [--scans N] [--rows H] [--cols W] [--voxel S] [--apriori file.xyz] [--period-ms P]
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/vofod.h"

namespace
{

struct Box
{
  float lo[3], hi[3];
};

// nearest positive hit of a ray with the ground plane z = 0 (inside +-60 m) and a list of boxes; inf when none
float cast(const float o[3], const float d[3], const std::vector<Box>& boxes)
{
  float best = INFINITY;
  if (d[2] < 0)
  {
    const float t = -o[2] / d[2];
    const float gx = o[0] + t * d[0], gy = o[1] + t * d[1];
    if (t > 0 && std::fabs(gx) < 60 && std::fabs(gy) < 60)
      best = t;
  }
  for (const Box& b : boxes)
  {
    float tmin = -INFINITY, tmax = INFINITY;
    for (int a = 0; a < 3; a++)
    {
      const float inv = 1.0f / d[a];
      float t1 = (b.lo[a] - o[a]) * inv, t2 = (b.hi[a] - o[a]) * inv;
      if (t1 > t2)
        std::swap(t1, t2);
      tmin = std::max(tmin, t1);
      tmax = std::min(tmax, t2);
    }
    if (tmax >= std::max(tmin, 0.0f) && tmin > 0 && tmin < best)
      best = tmin;
  }
  return best;
}

struct Cloud
{
  std::vector<float> x, y, z, intensity;
  std::vector<uint32_t> range;
  float tf[12];
};

}  // namespace

int main(int argc, char** argv)
{
  int n_scans = 40, rows = 32, cols = 1024, period_ms = 0;
  float voxel = 0.5f;
  std::string apriori;
  for (int i = 1; i + 1 < argc; i += 2)
  {
    const std::string k = argv[i];
    if (k == "--scans")
      n_scans = std::atoi(argv[i + 1]);
    else if (k == "--rows")
      rows = std::atoi(argv[i + 1]);
    else if (k == "--cols")
      cols = std::atoi(argv[i + 1]);
    else if (k == "--voxel")
      voxel = static_cast<float>(std::atof(argv[i + 1]));
    else if (k == "--apriori")
      apriori = argv[i + 1];
    else if (k == "--period-ms")
      period_ms = std::atoi(argv[i + 1]);
  }
  const float vfov = 45.0f * 3.14159265f / 180.0f;
  vofod_static_params sp;
  vofod_dyn_params dp;
  vofod_default_params(&sp, &dp);  // detection_params.yaml + sim.yaml defaults
  sp.voxel_size = voxel;
  sp.sensor_hrays = cols;
  sp.sensor_vrays = rows;
  sp.sensor_vfov = vfov;
  vofod_handle* h = nullptr;
  if (vofod_create(&sp, &dp, &h) != VOFOD_OK)
  {
    std::fprintf(stderr, "vofod_create failed\n");
    return 2;
  }
  std::vector<float> lut(3 * static_cast<size_t>(rows) * cols);
  vofod_sim_lut(cols, rows, vfov, lut.data());

  // static world: also the apriori background (initialize_apriori_map) unless a cloud file is given
  const std::vector<Box> statics = {{{8, -6, 0}, {12, -2, 5}}, {{-14, 5, 0}, {-9, 11, 3}}};
  if (!apriori.empty())
  {
    const float t[3] = {0, 0, 0}, c[3] = {0, 0, 0};
    size_t nl = 0, nv = 0;
    const int r = vofod_ingest_apriori(h, apriori.c_str(), t, 0.0, c, &nl, &nv);
    std::printf("apriori map: %zu points -> %zu voxels (status %d)\n", nl, nv, r);
  }
  else
  {
    std::vector<float> xyz;
    const float s = voxel;
    for (float gx = -40; gx < 40; gx += s)
      for (float gy = -40; gy < 40; gy += s)
        xyz.insert(xyz.end(), {gx + s / 2, gy + s / 2, s / 4});
    for (const Box& b : statics)
      for (float bx = b.lo[0] + s / 2; bx < b.hi[0]; bx += s)
        for (float by = b.lo[1] + s / 2; by < b.hi[1]; by += s)
          for (float bz = b.lo[2] + s / 2; bz < b.hi[2]; bz += s)
            if (bx - b.lo[0] < s || b.hi[0] - bx < s || by - b.lo[1] < s || b.hi[1] - by < s || b.hi[2] - bz < s)
              xyz.insert(xyz.end(), {bx, by, bz});
    vofod_load_apriori(h, xyz.data(), xyz.size() / 3);
    std::printf("apriori map: %zu synthetic background points\n", xyz.size() / 3);
  }

  // ---- the two background roles
  std::mutex mtx;
  std::condition_variable detection_cv;  // m_detection_cv (:951)
  std::atomic<bool> raycast_running{false}, stop{false};
  std::atomic<int> n_raycasts{0}, n_raycast_timeouts{0}, n_sep{0};
  std::vector<std::thread> raycast_threads;
  auto raycast_role = [&](std::shared_ptr<Cloud> c) {
    vofod_scan scan{c->x.data(), c->y.data(), c->z.data(), c->intensity.data(), c->range.data(), 4, cols, rows, VOFOD_MEM_HOST, 0.0};
    if (vofod_raycast_begin(h, &scan, c->tf) == VOFOD_OK)
    {
      {
        std::unique_lock<std::mutex> lck(mtx);
        detection_cv.wait_for(lck, std::chrono::milliseconds(800));  // :1530-1537
      }
      const int r = vofod_raycast_finish(h);
      n_raycasts++;
      if (r == VOFOD_ERR_RAYCAST_NO_DETECTION)
        n_raycast_timeouts++;
    }
    raycast_running = false;
  };
  std::thread sep_thread([&] {
    const auto period = std::chrono::milliseconds(100);  // sepclusters__period (detection_params.yaml:3)
    while (!stop)
    {
      std::this_thread::sleep_for(period_ms > 0 ? period : std::chrono::milliseconds(5));
      int sure = 0;
      if (vofod_sepclusters_begin(h, &sure) == VOFOD_OK && sure)
        vofod_sepclusters_finish(h);
      n_sep++;
    }
  });

  // ---- main role: one cloud per iteration
  size_t total_det = 0, det_on_target = 0, det_off_target = 0, msg_bytes = 0;
  std::vector<uint8_t> msg_buf;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n_scans; k++)
  {
    auto c = std::make_shared<Cloud>();
    const size_t n = static_cast<size_t>(rows) * cols;
    c->x.assign(n, 0), c->y.assign(n, 0), c->z.assign(n, 0), c->intensity.assign(n, 100.0f), c->range.assign(n, 0);
    const float yaw = 0.02f * k;
    const float o[3] = {0.5f * std::cos(0.1f * k), 0.5f * std::sin(0.1f * k), 4.0f};  // sensor 4 m above ground
    const float R[9] = {std::cos(yaw), -std::sin(yaw), 0, std::sin(yaw), std::cos(yaw), 0, 0, 0, 1};
    for (int r = 0; r < 3; r++)
    {
      for (int cc = 0; cc < 3; cc++)
        c->tf[4 * r + cc] = R[3 * r + cc];
      c->tf[4 * r + 3] = o[r];
    }
    std::vector<Box> boxes = statics;
    const float a = 0.15f * k;  // the flying target: 0.5 m cube on a circle of 6 m radius, 3 m above ground
    boxes.push_back({{6 * std::cos(a) - 0.25f, 6 * std::sin(a) - 0.25f, 2.75f}, {6 * std::cos(a) + 0.25f, 6 * std::sin(a) + 0.25f, 3.25f}});
    for (size_t i = 0; i < n; i++)
    {
      const float* ds = &lut[3 * i];
      const float d[3] = {R[0] * ds[0] + R[1] * ds[1] + R[2] * ds[2], R[3] * ds[0] + R[4] * ds[1] + R[5] * ds[2], R[6] * ds[0] + R[7] * ds[1] + R[8] * ds[2]};
      const float t = cast(o, d, boxes);
      if (!(t > 0.3f && t < 100.0f))
        continue;
      c->range[i] = static_cast<uint32_t>(std::lround(t * 1000.0f));
      const float rm = static_cast<float>(c->range[i]) * 0.001f;
      c->x[i] = ds[0] * rm, c->y[i] = ds[1] * rm, c->z[i] = ds[2] * rm;
    }
    vofod_scan scan{c->x.data(), c->y.data(), c->z.data(), c->intensity.data(), c->range.data(), 4, cols, rows, VOFOD_MEM_HOST, 0.1 * k};
    vofod_detection dets[64];
    size_t n_det = 0;
    const int st = vofod_process_scan(h, &scan, c->tf, VOFOD_SCAN_DEFAULT, dets, 64, &n_det, nullptr);
    if (st != VOFOD_OK && st != VOFOD_ERR_CAPACITY)
    {
      std::fprintf(stderr, "process_scan: status %d (%s)\n", st, vofod_last_error_string(h));
      stop = true;
      break;
    }
    detection_cv.notify_one();  // :951
    if (!raycast_running.exchange(true))  // :953-957
      raycast_threads.emplace_back(raycast_role, c);
    for (size_t i = 0; i < std::min<size_t>(n_det, 64); i++)
    {
      std::printf("scan %3d detection id %u conf %.3f at (%.2f, %.2f, %.2f), %llu points\n", k, dets[i].id, dets[i].confidence, dets[i].position[0], dets[i].position[1],
                  dets[i].position[2], static_cast<unsigned long long>(dets[i].n_points));
      // result check: the only thing flying in this world is the 0.5 m cube (OBB centre of its visible faces: within 0.6 m)
      const double e[3] = {dets[i].position[0] - 6 * std::cos(a), dets[i].position[1] - 6 * std::sin(a), dets[i].position[2] - 3.0};
      (std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) <= 0.6 ? det_on_target : det_off_target)++;
    }
    total_det += n_det;
    {
      // what m_pub_detections / m_pub_status would carry (:968-989, :1379-1385), byte for byte in the ROS 1 wire format
      const vofod_msg_header mh{static_cast<uint32_t>(k), static_cast<uint32_t>(k / 10), static_cast<uint32_t>((k % 10) * 100000000), "world_origin"};
      size_t nb = 0;
      msg_buf.resize(4096);
      if (vofod_serialize_detections(&mh, dets, std::min<size_t>(n_det, 64), msg_buf.data(), msg_buf.size(), &nb) == VOFOD_OK)
        msg_bytes += nb;
      vofod_status_info st_now;
      vofod_get_status(h, &st_now);
      if (vofod_serialize_status(&mh, 1, st_now.background_pts_sufficient && st_now.sure_background_sufficient, msg_buf.data(), msg_buf.size(), &nb) == VOFOD_OK)
        msg_bytes += nb;
    }
    if (period_ms > 0)
      std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop = true;
  sep_thread.join();
  for (auto& t : raycast_threads)
    t.join();
  vofod_status_info si;
  vofod_get_status(h, &si);
  std::printf("done: %d scans in %.3f s (%.1f scans/s incl. synthesis), detections %zu, detection_its %d, raycasts %d (timeouts %d), sepclusters passes %d\n", n_scans, secs, n_scans / secs,
              total_det, si.detection_its, n_raycasts.load(), n_raycast_timeouts.load(), n_sep.load());
  std::printf("check: %zu detections on the flying target, %zu elsewhere; %zu message bytes serialised\n", det_on_target, det_off_target, msg_bytes);
  vofod_destroy(h);
  return det_off_target > det_on_target ? 3 : 0;  // (the target is found in most scans once the map has settled; strays are rare)
}
