// Host-side classification tail of the product (C++): classify_cluster (vofod_nodelet.cpp:1648-1730)
// and extractDetections (:834-879) for the few far clusters that survive the device-side gates.
// It works on small sub-boxes of the voxel map read back from the device.  This is
// product code: it never touches oracle/.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <unordered_set>
#include <vector>

#include "common.h"
#include "eigsolve3.h"

#ifdef __HIPCC__
#define VT_HD __host__ __device__
#else
#define VT_HD
#endif

namespace vt
{

// VoxelMap geometry (voxel_map.cpp:592-613) on the host, same float expressions as the kernels.
struct Geom
{
  float off[3];
  float vs, vs_inv;
  int s[3];
  void coordToIdx(const float p[3], int o[3]) const
  {
    for (int a = 0; a < 3; a++)
      o[a] = static_cast<int>(std::floor((p[a] - off[a]) * vs_inv));
  }
  void idxToCoord(const int i[3], float c[3]) const
  {
    for (int a = 0; a < 3; a++)
      c[a] = (i[a] + 0.5f) * vs + off[a];
  }
  bool inLimits(const int i[3]) const { return i[0] >= 0 && i[0] < s[0] && i[1] >= 0 && i[1] < s[1] && i[2] >= 0 && i[2] < s[2]; }
  uint64_t lin(const int i[3]) const { return (static_cast<uint64_t>(i[2]) * s[1] + i[1]) * s[0] + i[0]; }
};

// A dense host copy of the map cells [lo, lo+n) (x fastest), with this scan's pending frontier writes applied.
struct Box
{
  int lo[3] = {0, 0, 0}, n[3] = {0, 0, 0};
  std::vector<float> v;
  bool has(const int i[3]) const { return i[0] >= lo[0] && i[0] < lo[0] + n[0] && i[1] >= lo[1] && i[1] < lo[1] + n[1] && i[2] >= lo[2] && i[2] < lo[2] + n[2]; }
  size_t at(const int i[3]) const { return (static_cast<size_t>(i[2] - lo[2]) * n[1] + (i[1] - lo[1])) * n[0] + (i[0] - lo[0]); }
};

struct Member
{
  uint32_t v;
  float p[3];
  uint32_t count;
};

// members of the candidate clusters of one frame, grouped by cluster (root) and ascending inside a group
struct MemberSpan
{
  const Member* p = nullptr;
  size_t n = 0;
  const Member* begin() const { return p; }
  const Member* end() const { return p + n; }
  size_t size() const { return n; }
  const Member& operator[](size_t i) const { return p[i]; }
};

struct MemberIndex
{
  std::vector<Member> all;                 // sorted by (root, v)
  std::vector<uint32_t> roots;             // parallel to `all`
  std::vector<std::pair<uint32_t, uint32_t>> first;  // (root, begin) sorted by root, plus a sentinel
  void build(std::vector<std::pair<uint64_t, Member>>& tmp)
  {
    std::sort(tmp.begin(), tmp.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    all.resize(tmp.size());
    first.clear();
    for (size_t i = 0; i < tmp.size(); i++)
    {
      all[i] = tmp[i].second;
      const uint32_t root = static_cast<uint32_t>(tmp[i].first >> 32);
      if (first.empty() || first.back().first != root)
        first.emplace_back(root, static_cast<uint32_t>(i));
    }
    first.emplace_back(0xffffffffu, static_cast<uint32_t>(tmp.size()));
  }
  MemberSpan of(uint32_t root) const
  {
    const auto it = std::lower_bound(first.begin(), first.end(), std::make_pair(root, 0u));
    if (it == first.end() || it->first != root)
      return MemberSpan{};
    return MemberSpan{all.data() + it->second, static_cast<size_t>((it + 1)->second - it->second)};
  }
};

struct Boxes
{
  float aabb_min[3], aabb_max[3];
  float obb_min[3], obb_max[3], obb_center[3];
};

// [3P] pcl::MomentOfInertiaEstimation: mean, covariance/n^2, principal axes (major >= middle >= minor,
// right-handed), AABB and OBB (centre = mean + R*shift).  Members in ascending index order; `get(i, p)` fetches the centre
// of member i.  Host (classification tail, fallback) and device (k_tail_prep) run this very function.
template <class Get>
VT_HD inline Boxes boxes_of_n(size_t n, Get get)
{
  Boxes b;
  float mean[3] = {0, 0, 0};
  for (int a = 0; a < 3; a++)
  {
    b.aabb_min[a] = FLT_MAX;
    b.aabb_max[a] = -FLT_MAX;
  }
  for (size_t i = 0; i < n; i++)
  {
    float p[3];
    get(i, p);
    for (int a = 0; a < 3; a++)
    {
      mean[a] += p[a];
      b.aabb_min[a] = p[a] < b.aabb_min[a] ? p[a] : b.aabb_min[a];
      b.aabb_max[a] = b.aabb_max[a] < p[a] ? p[a] : b.aabb_max[a];
    }
  }
  const float nf = static_cast<float>(static_cast<unsigned>(n));
  for (int a = 0; a < 3; a++)
    mean[a] /= nf;
  float cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (size_t i = 0; i < n; i++)
  {
    float p[3];
    get(i, p);
    const float c[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++)
        cov[r][q] += c[r] * c[q];
  }
  const float mass = 1.0f / static_cast<float>(n * n);
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++)
      cov[r][q] *= mass;
  // computeEigenVectors: Eigen::EigenSolver<Matrix3f> (eigsolve3.h), eigenvalues' real parts ordered major >= middle >= minor
  float w[3], V[3][3];
  ve::eigsolve3(cov, w, V);
  int ord[3] = {0, 1, 2};
  if (w[ord[0]] < w[ord[1]])
  {
    const int t = ord[0];
    ord[0] = ord[1];
    ord[1] = t;
  }
  if (w[ord[0]] < w[ord[2]])
  {
    const int t = ord[0];
    ord[0] = ord[2];
    ord[2] = t;
  }
  if (w[ord[1]] < w[ord[2]])
  {
    const int t = ord[1];
    ord[1] = ord[2];
    ord[2] = t;
  }
  float ax[3][3];
  for (int k = 0; k < 3; k++)
  {
    float u[3] = {V[0][ord[k]], V[1][ord[k]], V[2][ord[k]]};
    const float nn = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int a = 0; a < 3; a++)
      ax[k][a] = u[a] / nn;
  }
  const float cr[3] = {ax[1][1] * ax[2][2] - ax[1][2] * ax[2][1], ax[1][2] * ax[2][0] - ax[1][0] * ax[2][2], ax[1][0] * ax[2][1] - ax[1][1] * ax[2][0]};
  if (ax[0][0] * cr[0] + ax[0][1] * cr[1] + ax[0][2] * cr[2] <= 0.0f)
    for (int a = 0; a < 3; a++)
      ax[0][a] = -ax[0][a];
  for (int k = 0; k < 3; k++)
  {
    b.obb_min[k] = FLT_MAX;
    b.obb_max[k] = FLT_MIN;  // PCL initialises the OBB maximum with numeric_limits<float>::min()
  }
  for (size_t i = 0; i < n; i++)
  {
    float p[3];
    get(i, p);
    const float c[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
    for (int k = 0; k < 3; k++)
    {
      const float pr = c[0] * ax[k][0] + c[1] * ax[k][1] + c[2] * ax[k][2];
      if (pr <= b.obb_min[k])
        b.obb_min[k] = pr;
      if (pr >= b.obb_max[k])
        b.obb_max[k] = pr;
    }
  }
  float shift[3];
  for (int k = 0; k < 3; k++)
  {
    shift[k] = (b.obb_max[k] + b.obb_min[k]) / 2.0f;
    b.obb_min[k] -= shift[k];
    b.obb_max[k] -= shift[k];
  }
  for (int a = 0; a < 3; a++)
    b.obb_center[a] = mean[a] + (ax[0][a] * shift[0] + ax[1][a] * shift[1] + ax[2][a] * shift[2]);
  return b;
}

inline Boxes boxes_of(const MemberSpan& m)
{
  return boxes_of_n(m.size(), [&](size_t i, float p[3]) {
    p[0] = m[i].p[0];
    p[1] = m[i].p[1];
    p[2] = m[i].p[2];
  });
}

// VoxelMap::exploreToGround (voxel_map.cpp:402-488) as a flood fill over a read-back box.  The
// reference's DFS returns "connected" iff some reachable voxel satisfies one of its exit tests, and
// otherwise the set of reachable unknown voxels, both independent of the visiting order,
// so a visited-on-push fill yields the same answer.  `box` must cover the Manhattan ball of radius
// max_voxel_dist around the start voxel clipped to the map.
inline bool explore_to_ground(const Geom& g, const Box& box, const float p[3], float unknown_thr, float ground_thr, float max_voxel_dist,
                              std::vector<uint64_t>& explored_unknown)
{
  explored_unknown.clear();
  int o[3];
  g.coordToIdx(p, o);
  for (int a = 0; a < 3; a++)
    if (o[a] <= 0 || o[a] >= g.s[a] - 1)
      return true;
  std::unordered_set<uint64_t> seen;
  std::vector<std::array<int, 3>> stack;
  stack.push_back({o[0], o[1], o[2]});
  seen.insert(g.lin(o));
  while (!stack.empty())
  {
    const std::array<int, 3> cur = stack.back();
    stack.pop_back();
    const int ci[3] = {cur[0], cur[1], cur[2]};
    const float val = box.v[box.at(ci)];
    if (val > ground_thr)
      return true;
    if (!(val > unknown_thr))
      continue;
    explored_unknown.push_back(g.lin(ci));
    const int md = std::abs(ci[0] - o[0]) + std::abs(ci[1] - o[1]) + std::abs(ci[2] - o[2]);
    if (static_cast<float>(md) == max_voxel_dist - 1)
      return true;
    for (int a = 0; a < 3; a++)
      for (int s = 1; s >= -1; s -= 2)
      {
        int t[3] = {ci[0], ci[1], ci[2]};
        t[a] += s;
        if (t[a] < 0 || t[a] > g.s[a] - 1)
          continue;
        const int tmd = std::abs(t[0] - o[0]) + std::abs(t[1] - o[1]) + std::abs(t[2] - o[2]);
        if (!(static_cast<float>(tmd) <= max_voxel_dist))
          continue;
        if (seen.insert(g.lin(t)).second)
          stack.push_back({t[0], t[1], t[2]});
      }
  }
  return false;
}


// initialize_apriori_map (vofod_nodelet.cpp:214-226, 306-345): the loaded cloud goes through the rigid transform
// tf = Identity . rotate(AngleAxisf(yaw, UnitZ)) . translate(t + sim_correction) and the stock pcl::VoxelGrid<PointXYZ>
// centroid filter at the map's voxel size; the centroids come back as xyz triples in ascending cell order.
// Product implementation (the oracle has its own, sort based): cells are found by a counting sort over the occupied cell
// indices' hash buckets, which keeps the points of a cell in input order - the order the float sums are defined for
// ([3P] PCL 1.10 voxel_grid.hpp sorts (idx, point) pairs; equal idx keep no specified order, input order is the stable one).
struct AprioriGrid
{
  float inv;
  int min_b[3], div_b[3];
  bool ok;
  uint32_t cell(const float* q) const
  {
    int ijk[3];
    for (int r = 0; r < 3; r++)
      ijk[r] = static_cast<int>(std::floor(q[r] * inv) - static_cast<float>(min_b[r]));
    return static_cast<uint32_t>(ijk[0] + div_b[0] * (ijk[1] + div_b[1] * ijk[2]));
  }
};

inline void ingest_apriori_points(const std::vector<float>& xyz_in, const float t[3], double yaw_deg, const float corr[3], float leaf, std::vector<float>& out)
{
  out.clear();
  const size_t count = xyz_in.size() / 3;
  if (!count)
    return;
  // Eigen: AngleAxisf(angle, UnitZ).toRotationMatrix() -> [[c,-s,0],[s,c,0],[0,0,(1-c)*1*1+c]]; translate(v) adds linear * v
  const float angle = static_cast<float>(yaw_deg / 180.0 * M_PI);
  const float cs = std::cos(angle), sn = std::sin(angle);
  const float rot[3][3] = {{cs, -sn, 0.0f}, {sn, cs, 0.0f}, {0.0f, 0.0f, ((1.0f - cs) * 1.0f) * 1.0f + cs}};
  float shift[3], trans[3];
  for (int r = 0; r < 3; r++)
    shift[r] = t[r] + corr[r];
  for (int r = 0; r < 3; r++)
    trans[r] = (rot[r][0] * shift[0] + rot[r][1] * shift[1]) + rot[r][2] * shift[2];
  std::vector<float> world(xyz_in.size());
  float lo[3], hi[3];
  for (size_t i = 0; i < count; i++)
  {
    const float* src = &xyz_in[3 * i];
    float* dst = &world[3 * i];
    for (int r = 0; r < 3; r++)  // pcl::detail::Transformer<float>::se3: c0*x + (c1*y + (c2*z + c3))
      dst[r] = rot[r][0] * src[0] + (rot[r][1] * src[1] + (rot[r][2] * src[2] + trans[r]));
    for (int r = 0; r < 3; r++)
    {
      lo[r] = i ? std::min(lo[r], dst[r]) : dst[r];
      hi[r] = i ? std::max(hi[r], dst[r]) : dst[r];
    }
  }
  AprioriGrid grid;
  grid.inv = 1.0f / leaf;
  int64_t cells = 1;
  for (int r = 0; r < 3; r++)
  {
    grid.min_b[r] = static_cast<int>(std::floor(lo[r] * grid.inv));
    grid.div_b[r] = static_cast<int>(std::floor(hi[r] * grid.inv)) - grid.min_b[r] + 1;
    cells *= grid.div_b[r];
  }
  if (cells > 0x7fffffffll)
    return;  // "Leaf size is too small": PCL warns and copies the input; an apriori cloud that large is not supported here
  // counting sort by cell index, 16 bits at a time (stable: points of one cell stay in input order)
  std::vector<uint32_t> key(count), idx(count), idx2(count);
  for (size_t i = 0; i < count; i++)
  {
    key[i] = grid.cell(&world[3 * i]);
    idx[i] = static_cast<uint32_t>(i);
  }
  for (int pass = 0; pass < 2; pass++)
  {
    std::vector<uint32_t> start(65537, 0u);
    const int sh = 16 * pass;
    for (size_t i = 0; i < count; i++)
      start[((key[idx[i]] >> sh) & 0xffffu) + 1]++;
    for (size_t d = 0; d < 65536; d++)
      start[d + 1] += start[d];
    for (size_t i = 0; i < count; i++)
      idx2[start[(key[idx[i]] >> sh) & 0xffffu]++] = idx[i];
    idx.swap(idx2);
  }
  size_t run_begin = 0;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (size_t i = 0; i < count; i++)
  {
    const float* q = &world[3 * idx[i]];
    for (int r = 0; r < 3; r++)
      acc[r] += q[r];
    if (i + 1 == count || key[idx[i + 1]] != key[idx[i]])
    {
      const float members = static_cast<float>(i + 1 - run_begin);
      for (int r = 0; r < 3; r++)
      {
        out.push_back(acc[r] / members);
        acc[r] = 0.0f;
      }
      run_begin = i + 1;
    }
  }
}

}  // namespace vt
