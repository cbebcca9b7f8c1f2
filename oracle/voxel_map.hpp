// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's VoxelMap
// (src/voxel_map.cpp, include/vofod/voxel_map.h).  Never linked by the product
// (vofod_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// PARITY UNPINNED: the reference ships no tests/golden vectors and cannot be built
// here (needs PCL/Eigen/ROS: SURVEY.md §8c), so this restatement is pinned only by
// the hand-derived known-answer tests in tests/test_oracle_kat.py.
//
// Every function cites the reference lines it follows.  Arithmetic is kept in the
// reference's types (float coordinates, int indices) and evaluation order; build
// with -ffp-contract=off so no FMA contraction changes a floor().
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <tuple>
#include <unordered_set>
#include <vector>

namespace vo
{

using idx3_t = std::array<int, 3>;

struct idx3_hash
{
  size_t operator()(const idx3_t& t) const noexcept
  {
    // any hash works: the reference's hash_tuple (voxel_map.h:146-202) only feeds an unordered_set
    uint64_t h = 1469598103934665603ull;
    for (int v : t)
    {
      h ^= static_cast<uint32_t>(v);
      h *= 1099511628211ull;
    }
    return static_cast<size_t>(h);
  }
};

class VoxelMap
{
public:
  float off[3] = {0, 0, 0};
  float vs = 0, vs_inv = 0, half = 0;
  int sx = 0, sy = 0, sz = 0;
  std::vector<float> data;

  // voxel_map.cpp:11-19
  void resize_center(const float center[3], const float dims[3], const float voxel_size)
  {
    const float inv = 1.0f / voxel_size;
    float offset[3];
    int sizes[3];
    for (int i = 0; i < 3; i++)
    {
      offset[i] = center[i] - dims[i] / 2.0f;
      sizes[i] = static_cast<int>(std::ceil(inv * dims[i])) + 1;
    }
    resize(offset, sizes, voxel_size);
  }

  // voxel_map.cpp:21-48
  void resize(const float offset[3], const int sizes[3], const float voxel_size)
  {
    vs = voxel_size;
    half = voxel_size / 2.0f;
    vs_inv = 1.0f / vs;
    for (int i = 0; i < 3; i++)
      off[i] = offset[i];
    sx = sizes[0];
    sy = sizes[1];
    sz = sizes[2];
    data.resize(static_cast<size_t>(sx) * sy * sz);
  }

  size_t size() const { return data.size(); }
  size_t lin(int ix, int iy, int iz) const { return static_cast<size_t>(ix) + static_cast<size_t>(iy) * sx + static_cast<size_t>(iz) * sx * sy; }  // voxel_map.cpp:81
  float& at(int ix, int iy, int iz) { return data.at(lin(ix, iy, iz)); }
  float at(int ix, int iy, int iz) const { return data.at(lin(ix, iy, iz)); }

  // voxel_map.cpp:592-599
  idx3_t coordToIdx(float x, float y, float z) const
  {
    const int ix = static_cast<int>(std::floor((x - off[0]) * vs_inv));
    const int iy = static_cast<int>(std::floor((y - off[1]) * vs_inv));
    const int iz = static_cast<int>(std::floor((z - off[2]) * vs_inv));
    return {ix, iy, iz};
  }

  // voxel_map.cpp:607-613
  std::array<float, 3> idxToCoord(int ix, int iy, int iz) const
  {
    const float x = (ix + 0.5f) * vs + off[0];
    const float y = (iy + 0.5f) * vs + off[1];
    const float z = (iz + 0.5f) * vs + off[2];
    return {x, y, z};
  }

  // voxel_map.cpp:297-300
  bool inLimitsIdx(int ix, int iy, int iz) const { return ix >= 0 && ix < sx && iy >= 0 && iy < sy && iz >= 0 && iz < sz; }
  // voxel_map.cpp:289-293
  bool inLimits(float x, float y, float z) const
  {
    const auto i = coordToIdx(x, y, z);
    return inLimitsIdx(i[0], i[1], i[2]);
  }

  void setTo(float v) { std::fill(data.begin(), data.end(), v); }  // voxel_map.cpp:275-278

  // voxel_map.cpp:216-222
  uint64_t nVoxelsOver(float threshold) const
  {
    uint64_t ret = 0;
    for (const float v : data)
      ret += v > threshold;
    return ret;
  }

  // voxel_map.cpp:187-212: x outer, y, z inner; coordinates are the integer indices as floats
  void voxelsAsVoxelPC(float threshold, std::vector<float>& px, std::vector<float>& py, std::vector<float>& pz, std::vector<float>& pi) const
  {
    for (int x = 0; x < sx; x++)
      for (int y = 0; y < sy; y++)
        for (int z = 0; z < sz; z++)
        {
          const float m = data[lin(x, y, z)];
          if (m > threshold)
          {
            px.push_back(static_cast<float>(x));
            py.push_back(static_cast<float>(y));
            pz.push_back(static_cast<float>(z));
            pi.push_back(m);
          }
        }
  }

  // voxel_map.cpp:376-400.  Eigen's integer-vector norm() truncates sqrt;
  // the cube scanned is half-open [o-d, o+d).
  bool hasCloseTo(float x, float y, float z, float max_dist, float threshold) const
  {
    const idx3_t o = coordToIdx(x, y, z);
    const float max_dist_idx = max_dist * vs_inv;
    const int d = static_cast<int>(std::ceil(max_dist_idx));
    const int bx = std::max(o[0] - d, 0), by = std::max(o[1] - d, 0), bz = std::max(o[2] - d, 0);
    const int ex = std::min(o[0] + d, sx), ey = std::min(o[1] + d, sy), ez = std::min(o[2] + d, sz);
    for (int xi = bx; xi < ex; xi++)
      for (int yi = by; yi < ey; yi++)
        for (int zi = bz; zi < ez; zi++)
        {
          if (!(at(xi, yi, zi) > threshold))
            continue;
          const int dx = xi - o[0], dy = yi - o[1], dz = zi - o[2];
          const int n2 = dx * dx + dy * dy + dz * dz;
          const int norm = static_cast<int>(std::sqrt(static_cast<double>(n2)));  // Eigen int norm(): sqrt via double, truncated
          if (static_cast<float>(norm) <= max_dist_idx)
            return true;
        }
    return false;
  }

  static int manhattan(const idx3_t& a, const idx3_t& b) { return std::abs(a[0] - b[0]) + std::abs(a[1] - b[1]) + std::abs(a[2] - b[2]); }  // voxel_map.cpp:318-323

  // voxel_map.cpp:402-488, literal DFS including the duplicate pushes.
  std::pair<bool, std::vector<idx3_t>> exploreToGround(float x, float y, float z, float unknown_threshold, float ground_threshold, float max_voxel_dist) const
  {
    const idx3_t orig = coordToIdx(x, y, z);
    if (orig[0] <= 0 || orig[1] <= 0 || orig[2] <= 0)
      return {true, {}};
    if (orig[0] >= sx - 1 || orig[1] >= sy - 1 || orig[2] >= sz - 1)
      return {true, {}};

    std::unordered_set<idx3_t, idx3_hash> explored;
    std::vector<idx3_t> explored_unknown;
    std::vector<idx3_t> to_explore;
    to_explore.push_back(orig);
    while (!to_explore.empty())
    {
      const idx3_t cur = to_explore.back();
      to_explore.pop_back();
      const float cur_val = at(cur[0], cur[1], cur[2]);
      if (cur_val > ground_threshold)
        return {true, {}};
      if (cur_val > unknown_threshold)
      {
        explored_unknown.push_back(cur);
        if (static_cast<float>(manhattan(orig, cur)) == max_voxel_dist - 1)
          return {true, {}};
        const int lim[3] = {sx, sy, sz};
        for (int a = 0; a < 3; a++)  // positive directions x, y, z (:437-454)
          if (cur[a] < lim[a] - 1)
          {
            idx3_t t = cur;
            t[a] += 1;
            if (explored.count(t) == 0 && static_cast<float>(manhattan(orig, t)) <= max_voxel_dist)
              to_explore.push_back(t);
          }
        for (int a = 0; a < 3; a++)  // negative directions x, y, z (:460-477)
          if (cur[a] > 0)
          {
            idx3_t t = cur;
            t[a] -= 1;
            if (explored.count(t) == 0 && static_cast<float>(manhattan(orig, t)) <= max_voxel_dist)
              to_explore.push_back(t);
          }
      }
      explored.insert(cur);
    }
    return {false, explored_unknown};
  }

  // voxel_map.cpp:229-263, Amanatides-Woo.  f(ddist, ix, iy, iz).
  template <class F>
  void forEachRay(const float start[3], const float dir[3], const float length, F f) const
  {
    float absdir[3], tdelta[3], tmax[3], last[3];
    int step[3];
    idx3_t cur = coordToIdx(start[0], start[1], start[2]);
    const auto ctr = idxToCoord(cur[0], cur[1], cur[2]);
    const int lim[3] = {sx, sy, sz};
    for (int i = 0; i < 3; i++)
    {
      absdir[i] = std::fabs(dir[i]);
      step[i] = (dir[i] > 0.0f) - (dir[i] < 0.0f);        // cwiseSign
      tdelta[i] = (1.0f / absdir[i]) * vs;                // absdir.cwiseInverse()*m_voxel_size
      const float ctr_offset = ctr[i] - start[i];
      tmax[i] = (half + static_cast<float>(step[i]) * ctr_offset) / absdir[i];
      last[i] = step[i] > 0 ? static_cast<float>(lim[i] - 1) : 0.0f;
    }
    float prev_dist = 0.0f;
    while (prev_dist < length)
    {
      int i = 0;  // minCoeff(&i): first minimum wins
      if (tmax[1] < tmax[i])
        i = 1;
      if (tmax[2] < tmax[i])
        i = 2;
      const float dist = tmax[i];
      const float ddist = std::min(dist, length) - prev_dist;
      f(ddist, cur[0], cur[1], cur[2]);
      prev_dist = dist;
      if (static_cast<float>(cur[i]) == last[i])
        break;
      cur[i] += step[i];
      tmax[i] += tdelta[i];
    }
  }

  // voxel_map.cpp:547-584
  VoxelMap getSubmapCopy(const float min_pt[3], const float max_pt[3], const int inflate) const
  {
    idx3_t mn = coordToIdx(min_pt[0], min_pt[1], min_pt[2]);
    idx3_t mx = coordToIdx(max_pt[0], max_pt[1], max_pt[2]);
    const int lim[3] = {sx, sy, sz};
    for (int a = 0; a < 3; a++)
    {
      mn[a] = std::clamp(mn[a] - inflate, 0, lim[a] - 1);
      mx[a] = std::clamp(mx[a] + inflate, 0, lim[a] - 1);
    }
    const auto c = idxToCoord(mn[0], mn[1], mn[2]);
    float sub_off[3];
    int sub_size[3];
    for (int a = 0; a < 3; a++)
    {
      sub_off[a] = c[a] - vs / 2.0f;
      sub_size[a] = mx[a] - mn[a] + 1;
    }
    VoxelMap ret;
    ret.resize(sub_off, sub_size, vs);
    for (int xi = 0; xi < sub_size[0]; xi++)
      for (int yi = 0; yi < sub_size[1]; yi++)
        for (int zi = 0; zi < sub_size[2]; zi++)
          ret.at(xi, yi, zi) = data.at(lin(xi + mn[0], yi + mn[1], zi + mn[2]));
    return ret;
  }
};

}  // namespace vo
