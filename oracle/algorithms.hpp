// ORACLE — TEST INFRASTRUCTURE ONLY (see voxel_map.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the reference's L4 filters and of the third-party PCL pieces the
// hot path calls.  [3P] marks behaviour restated from PCL 1.10 /
// FLANN 1.9.1 / Eigen 3.3.7 (Ubuntu 20.04 / ROS Noetic distro versions; an inference,
// the reference pins nothing: package.xml:19-33).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "../include/vofod.h"

namespace vo
{

struct Cloud
{
  std::vector<float> x, y, z;
  std::vector<float> intensity;  // optional
  size_t size() const { return x.size(); }
};

// ---------------------------------------------------------------- CropBox [3P]
// pcl::CropBox<PointT>::applyFilter with identity transform: a point is inside iff
// min <= p <= max on every axis (inclusive); negative=true keeps the points outside.
// Non-finite points are dropped (PCL does so when !is_dense; a NaN also fails every
// comparison, i.e. counts as "inside", and is dropped by the negative crop).
inline bool crop_inside(const float p[3], const float mn[3], const float mx[3])
{
  if (p[0] < mn[0] || p[1] < mn[1] || p[2] < mn[2])
    return false;
  if (p[0] > mx[0] || p[1] > mx[1] || p[2] > mx[2])
    return false;
  return true;
}

// ------------------------------------------------------ transformPointCloud [3P]
// pcl::detail::Transformer<float>::se3 (SSE form, the one an x86-64 build uses):
//   out = c0*x + (c1*y + (c2*z + c3))   per row, mul and add rounded separately.
// tf is row-major 3x4.
inline void transform_point(const float tf[12], const float p[3], float out[3])
{
  for (int r = 0; r < 3; r++)
  {
    const float p0 = tf[4 * r + 0] * p[0];
    const float p1 = tf[4 * r + 1] * p[1];
    const float p2 = tf[4 * r + 2] * p[2];
    out[r] = p0 + (p1 + (p2 + tf[4 * r + 3]));
  }
}

// ------------------------------------------------- VoxelGridWeighted / Counted
struct GridOut
{
  std::vector<vofod_point_xyzr> pts;
  std::vector<uint32_t> keys;
  vofod_grid_desc grid{};
  int status = VOFOD_OK;
};

// voxel_grid_weighted.cpp:41-190 and voxel_grid_counted.cpp:49-196 (identical up to the
// weight: weighted -> number of points in the voxel :181; counted -> count_if over the
// *positional* range [first_index,last_index) of the input cloud :185-187).
// [3P] pcl::VoxelGrid base: inverse_leaf_size_ = 1/leaf, min_points_per_voxel_ = 0,
// pcl::getMinMax3D = plain min/max over the (finite) points.
inline GridOut voxel_grid(const Cloud& in, const float leaf_in[3], const bool align_voxels, const float align_center[3], const bool counted,
                          const float threshold)
{
  GridOut out;
  const size_t n = in.size();
  float leaf[3], inv[3];
  for (int a = 0; a < 3; a++)
  {
    leaf[a] = leaf_in[a];
    inv[a] = 1.0f / leaf[a];
    out.grid.leaf[a] = leaf[a];
  }
  if (n == 0)
    return out;

  // getMinMax3D :58
  float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < n; i++)
  {
    const float p[3] = {in.x[i], in.y[i], in.z[i]};
    for (int a = 0; a < 3; a++)
    {
      min_p[a] = std::min(min_p[a], p[a]);
      max_p[a] = std::max(max_p[a], p[a]);
    }
  }

  // overflow guard :61-69
  const int64_t dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inv[0]) + 2;
  const int64_t dy = static_cast<int64_t>((max_p[1] - min_p[1]) * inv[1]) + 2;
  const int64_t dz = static_cast<int64_t>((max_p[2] - min_p[2]) * inv[2]) + 2;
  if (dx * dy * dz > static_cast<int64_t>(std::numeric_limits<int32_t>::max()))
  {
    out.status = VOFOD_ERR_INDEX_OVERFLOW;
    return out;
  }

  // :72-80
  int min_b[3], max_b[3];
  float offset[3];
  for (int a = 0; a < 3; a++)
  {
    min_b[a] = static_cast<int>(std::floor(min_p[a] * inv[a]));
    max_b[a] = static_cast<int>(std::floor(max_p[a] * inv[a]));
    offset[a] = static_cast<float>(min_b[a]) * leaf[a];
  }
  // :81-106
  if (align_voxels)
  {
    for (int a = 0; a < 3; a++)
    {
      float aco = std::fmod(align_center[a] - leaf[a] / 2, leaf[a]);
      if (aco < 0)
        aco += leaf[a];
      offset[a] -= aco;
      min_b[a] = static_cast<int>(std::floor(offset[a] * inv[a]));
    }
  }
  // :109-113
  int div_b[3];
  for (int a = 0; a < 3; a++)
    div_b[a] = max_b[a] - min_b[a] + 1;
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  for (int a = 0; a < 3; a++)
  {
    out.grid.offset[a] = offset[a];
    out.grid.min_b[a] = min_b[a];
    out.grid.div_b[a] = div_b[a];
  }

  // first pass :122-139
  struct item
  {
    uint32_t idx;
    uint32_t pt;
    int ijk[3];
  };
  std::vector<item> index_vector;
  index_vector.reserve(n);
  for (size_t i = 0; i < n; i++)
  {
    item it;
    it.ijk[0] = static_cast<int>(std::floor((in.x[i] - offset[0]) * inv[0]));
    it.ijk[1] = static_cast<int>(std::floor((in.y[i] - offset[1]) * inv[1]));
    it.ijk[2] = static_cast<int>(std::floor((in.z[i] - offset[2]) * inv[2]));
    const int idx = it.ijk[0] * mul[0] + it.ijk[1] * mul[1] + it.ijk[2] * mul[2];
    it.idx = static_cast<uint32_t>(idx);
    it.pt = static_cast<uint32_t>(i);
    index_vector.push_back(it);
  }
  // second pass :143 (std::sort in the reference; stable here so that the ijk read from the
  // first element of a run is reproducible — equal keys carry equal ijk except when a
  // rounding artefact pushes ijk0 to div_b[0], which aliases two cells onto one key)
  std::stable_sort(index_vector.begin(), index_vector.end(), [](const item& a, const item& b) { return a.idx < b.idx; });

  // third + fourth pass :147-188 / counted :179-194
  size_t index = 0;
  while (index < index_vector.size())
  {
    size_t i = index + 1;
    while (i < index_vector.size() && index_vector[i].idx == index_vector[index].idx)
      ++i;
    const item& first = index_vector[index];
    vofod_point_xyzr p;
    p.x = (static_cast<float>(first.ijk[0]) + 0.5f) * leaf[0] + offset[0];
    p.y = (static_cast<float>(first.ijk[1]) + 0.5f) * leaf[1] + offset[1];
    p.z = (static_cast<float>(first.ijk[2]) + 0.5f) * leaf[2] + offset[2];
    if (!counted)
      p.range = static_cast<uint32_t>(i - index);
    else
    {
      // voxel_grid_counted.cpp:185-187: positions [first_index,last_index) of the *input cloud*
      uint32_t c = 0;
      for (size_t pos = index; pos < i; pos++)
        c += in.intensity[pos] > threshold;
      p.range = c;
    }
    out.pts.push_back(p);
    out.keys.push_back(first.idx);
    index = i;
  }
  return out;
}

// -------------------------------------------- EuclideanClusterExtraction [3P]
// pcl::extractEuclideanClusters + EuclideanClusterExtraction::extract with the defaults
// clusterCloud leaves in place (vofod_nodelet.cpp:689-698: min 1, max INT_MAX):
// connected components of { (a,b) : d2(a,b) < tol*tol }, d2 accumulated in float as
// ((dx*dx + dy*dy) + dz*dz) (FLANN L2_Simple, RadiusResultSet keeps dist < radius).
// Returned labels: label[i] = smallest member index of i's component.  The neighbour
// search is a uniform hash grid of cell size tol (the reference uses a kd-tree; the set of
// neighbours, hence the components, is the same).
inline std::vector<uint32_t> euclidean_labels(const std::vector<vofod_point_xyzr>& pts, const float tolerance)
{
  const size_t n = pts.size();
  std::vector<uint32_t> label(n, UINT32_MAX);
  if (n == 0)
    return label;
  const float r2 = tolerance * tolerance;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  for (const auto& p : pts)
  {
    mn[0] = std::min(mn[0], p.x);
    mn[1] = std::min(mn[1], p.y);
    mn[2] = std::min(mn[2], p.z);
  }
  const double cell = std::max(static_cast<double>(tolerance), 1e-9) * 1.0001;  // strictly > tol so +-1 cell suffices
  auto cellof = [&](const vofod_point_xyzr& p, int64_t c[3]) {
    c[0] = static_cast<int64_t>(std::floor((static_cast<double>(p.x) - mn[0]) / cell));
    c[1] = static_cast<int64_t>(std::floor((static_cast<double>(p.y) - mn[1]) / cell));
    c[2] = static_cast<int64_t>(std::floor((static_cast<double>(p.z) - mn[2]) / cell));
  };
  auto keyof = [](const int64_t c[3]) { return (static_cast<uint64_t>(c[0] + 1) << 42) ^ (static_cast<uint64_t>(c[1] + 1) << 21) ^ static_cast<uint64_t>(c[2] + 1); };
  std::unordered_map<uint64_t, std::vector<uint32_t>> cells;
  cells.reserve(n);
  for (size_t i = 0; i < n; i++)
  {
    int64_t c[3];
    cellof(pts[i], c);
    cells[keyof(c)].push_back(static_cast<uint32_t>(i));
  }

  std::vector<uint32_t> queue;
  for (size_t seed = 0; seed < n; seed++)
  {
    if (label[seed] != UINT32_MAX)
      continue;
    queue.clear();
    queue.push_back(static_cast<uint32_t>(seed));
    label[seed] = static_cast<uint32_t>(seed);
    for (size_t qi = 0; qi < queue.size(); qi++)
    {
      const vofod_point_xyzr& q = pts[queue[qi]];
      int64_t c[3];
      cellof(q, c);
      for (int64_t cx = c[0] - 1; cx <= c[0] + 1; cx++)
        for (int64_t cy = c[1] - 1; cy <= c[1] + 1; cy++)
          for (int64_t cz = c[2] - 1; cz <= c[2] + 1; cz++)
          {
            const int64_t cc[3] = {cx, cy, cz};
            const auto it = cells.find(keyof(cc));
            if (it == cells.end())
              continue;
            for (const uint32_t j : it->second)
            {
              if (label[j] != UINT32_MAX)
                continue;
              const float ddx = q.x - pts[j].x, ddy = q.y - pts[j].y, ddz = q.z - pts[j].z;
              float d2 = ddx * ddx;
              d2 += ddy * ddy;
              d2 += ddz * ddz;
              if (d2 < r2)
              {
                label[j] = static_cast<uint32_t>(seed);
                queue.push_back(j);
              }
            }
          }
    }
  }
  return label;
}

// Brute-force O(n^2) union-find over the same predicate; used by the tests to check
// euclidean_labels itself.
inline std::vector<uint32_t> euclidean_labels_bruteforce(const std::vector<vofod_point_xyzr>& pts, const float tolerance)
{
  const size_t n = pts.size();
  std::vector<uint32_t> parent(n);
  std::iota(parent.begin(), parent.end(), 0u);
  std::function<uint32_t(uint32_t)> find = [&](uint32_t v) {
    while (parent[v] != v)
    {
      parent[v] = parent[parent[v]];
      v = parent[v];
    }
    return v;
  };
  const float r2 = tolerance * tolerance;
  for (size_t i = 0; i < n; i++)
    for (size_t j = i + 1; j < n; j++)
    {
      const float ddx = pts[i].x - pts[j].x, ddy = pts[i].y - pts[j].y, ddz = pts[i].z - pts[j].z;
      float d2 = ddx * ddx;
      d2 += ddy * ddy;
      d2 += ddz * ddz;
      if (d2 < r2)
      {
        const uint32_t a = find(static_cast<uint32_t>(i)), b = find(static_cast<uint32_t>(j));
        if (a != b)
          parent[std::max(a, b)] = std::min(a, b);
      }
    }
  std::vector<uint32_t> label(n);
  for (size_t i = 0; i < n; i++)
    label[i] = find(static_cast<uint32_t>(i));
  return label;
}

struct Cluster
{
  std::vector<int> indices;  // ascending (extract_clusters sorts them)
};

// Canonical cluster order: size descending (EuclideanClusterExtraction::extract's reverse
// sort by size), ties by smallest member (what libstdc++'s insertion sort yields for <= 16
// clusters; unspecified by the reference beyond that — SURVEY H3).
inline std::vector<Cluster> clusters_from_labels(const std::vector<uint32_t>& label)
{
  std::unordered_map<uint32_t, size_t> slot;
  std::vector<Cluster> cl;
  for (size_t i = 0; i < label.size(); i++)
  {
    auto it = slot.find(label[i]);
    if (it == slot.end())
    {
      it = slot.emplace(label[i], cl.size()).first;
      cl.emplace_back();
    }
    cl[it->second].indices.push_back(static_cast<int>(i));
  }
  std::stable_sort(cl.begin(), cl.end(), [](const Cluster& a, const Cluster& b) {
    if (a.indices.size() != b.indices.size())
      return a.indices.size() > b.indices.size();
    return a.indices.front() < b.indices.front();
  });
  return cl;
}

// ------------------------------------------- MomentOfInertiaEstimation [3P]
struct Boxes
{
  float aabb_min[3], aabb_max[3];
  float obb_min[3], obb_max[3], obb_center[3];
  float axes[3][3];  // major, middle, minor
  float eig[3];
};

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi, double).  The reference reaches its
// eigenvectors through Eigen::EigenSolver<Matrix3f> (general real solver); any correct
// solver gives the same eigen-spaces, and the quantities the hot path consumes (OBB centre,
// OBB diagonal) do not depend on the basis chosen inside a degenerate eigen-space whenever
// the projections on it are symmetric or zero — parity there is by tolerance.
inline void jacobi_eigen3(const double A_in[3][3], double w[3], double V[3][3])
{
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      A[i][j] = A_in[i][j];
      V[i][j] = i == j;
    }
  for (int sweep = 0; sweep < 64; sweep++)
  {
    const double offd = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    if (offd < 1e-300)
      break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++)
      {
        if (std::fabs(A[p][q]) < 1e-300)
          continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++)
        {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++)
        {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++)
        {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; i++)
    w[i] = A[i][i];
}

// ------------------------------------------------------------------------------------------------------------------
// [3P] Eigen 3.3.7 `EigenSolver<Matrix3f>` restated (Eigen is not in /root/reference: package.xml pulls it through PCL;
// version by distro inference, see the header).  pcl::MomentOfInertiaEstimation::computeEigenVectors
// (moment_of_inertia_estimation.hpp) hands the float covariance to the GENERAL real solver, not to the self-adjoint one:
//   EigenSolver::compute        -> RealSchur::compute (scale by max |a_ij|, HessenbergDecomposition, computeFromHessenberg:
//                                  findSmallSubdiagEntry / splitOffTwoRows / computeShift / initFrancisQRStep /
//                                  performFrancisQRStep), eigenvalues from the quasi-triangular T, doComputeEigenvectors
//                                  (back substitution + multiplication by the Schur vectors)
//   EigenSolver::eigenvectors   -> columns normalised; a complex pair gives (v_j + i v_j+1) and its conjugate,
//                                  of which PCL keeps `.real()`.
// All arithmetic in float, operation order as in the Eigen sources (dynamic-size blocks: plain left-to-right loops).
// For a symmetric matrix the eigen-SPACES are those of any solver; inside a degenerate eigen-space (2 x 2 voxel squares,
// cubes ...) the basis - and with it the OBB's extents - is decided by this algorithm's rounding path, which is why the
// restatement follows it step by step.  It cannot be pinned against the real library here (parity unpinned).
struct EigenSolver3f
{
  float T[3][3], U[3][3];
  float re[3], im[3];
  float vec_re[3][3];  // column j = real part of eigenvector j after EigenSolver::eigenvectors()
  bool ok = true;

  static void make_householder(const float* v, int size, float* ess, float& tau, float& beta)
  {
    float tail_sq = 0.0f;
    for (int i = 1; i < size; i++)
      tail_sq += v[i] * v[i];
    const float c0 = v[0];
    const float tol = std::numeric_limits<float>::min();
    if (tail_sq <= tol)
    {
      tau = 0.0f;
      beta = c0;
      for (int i = 0; i < size - 1; i++)
        ess[i] = 0.0f;
    }
    else
    {
      beta = std::sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.0f)
        beta = -beta;
      for (int i = 0; i < size - 1; i++)
        ess[i] = v[i + 1] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
  }
  // M.block(r0, c0, nr, nc).applyHouseholderOnTheLeft(ess, tau)
  static void house_left(float M[3][3], int r0, int c0, int nr, int nc, const float* ess, float tau)
  {
    if (nr == 1)
    {
      for (int c = 0; c < nc; c++)
        M[r0][c0 + c] *= 1.0f - tau;
      return;
    }
    if (tau == 0.0f)
      return;
    for (int c = 0; c < nc; c++)
    {
      float tmp = 0.0f;
      for (int r = 1; r < nr; r++)
        tmp += ess[r - 1] * M[r0 + r][c0 + c];
      tmp += M[r0][c0 + c];
      M[r0][c0 + c] -= tau * tmp;
      for (int r = 1; r < nr; r++)
        M[r0 + r][c0 + c] -= tau * ess[r - 1] * tmp;
    }
  }
  static void house_right(float M[3][3], int r0, int c0, int nr, int nc, const float* ess, float tau)
  {
    if (nc == 1)
    {
      for (int r = 0; r < nr; r++)
        M[r0 + r][c0] *= 1.0f - tau;
      return;
    }
    if (tau == 0.0f)
      return;
    for (int r = 0; r < nr; r++)
    {
      float tmp = 0.0f;
      for (int c = 1; c < nc; c++)
        tmp += M[r0 + r][c0 + c] * ess[c - 1];
      tmp += M[r0 + r][c0];
      M[r0 + r][c0] -= tau * tmp;
      for (int c = 1; c < nc; c++)
        M[r0 + r][c0 + c] -= tau * tmp * ess[c - 1];
    }
  }
  // JacobiRotation::makeGivens(p, q) (real case) and the two applications of splitOffTwoRows
  static void make_givens(float p, float q, float& c, float& s)
  {
    if (q == 0.0f)
    {
      c = p < 0.0f ? -1.0f : 1.0f;
      s = 0.0f;
    }
    else if (p == 0.0f)
    {
      c = 0.0f;
      s = q < 0.0f ? 1.0f : -1.0f;
    }
    else if (std::fabs(p) > std::fabs(q))
    {
      const float t = q / p;
      float u = std::sqrt(1.0f + t * t);
      if (p < 0.0f)
        u = -u;
      c = 1.0f / u;
      s = -t * c;
    }
    else
    {
      const float t = p / q;
      float u = std::sqrt(1.0f + t * t);
      if (q < 0.0f)
        u = -u;
      s = -1.0f / u;
      c = -t * s;
    }
  }

  void compute(const float A[3][3])
  {
    const int size = 3;
    const float eps = std::numeric_limits<float>::epsilon();
    const float consider_as_zero = std::numeric_limits<float>::min();
    float scale = 0.0f;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        scale = std::max(scale, std::fabs(A[i][j]));
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        T[i][j] = 0.0f;
        U[i][j] = i == j ? 1.0f : 0.0f;
      }
    if (scale < consider_as_zero)
    {
      finish();
      return;
    }
    // ---- HessenbergDecomposition::_compute on A / scale
    float H[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        H[i][j] = A[i][j] / scale;
    float hcoef[2] = {0.0f, 0.0f}, ess0 = 0.0f;
    for (int i = 0; i < size - 1; i++)
    {
      const int rem = size - i - 1;
      float v[2] = {H[i + 1][i], rem > 1 ? H[i + 2][i] : 0.0f}, ess[1] = {0.0f}, tau, beta;
      make_householder(v, rem, ess, tau, beta);
      H[i + 1][i] = beta;
      if (rem > 1)
        H[i + 2][i] = ess[0];  // (makeHouseholderInPlace keeps the essential part below the subdiagonal)
      hcoef[i] = tau;
      if (i == 0)
        ess0 = ess[0];
      house_left(H, i + 1, i + 1, rem, rem, ess, tau);
      house_right(H, 0, i + 1, size, rem, ess, tau);
    }
    // matrixH: upper Hessenberg part; matrixQ = H_0 (H_1 acts on one row: tau = 0)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        T[i][j] = (i <= j + 1) ? H[i][j] : 0.0f;
    {
      // HouseholderSequence(matA, hCoeffs).setLength(2).setShift(1) evaluated on the identity (evalTo: from the last vector
      // up): Q = I - tau0 * u u^T with u = (0, 1, ess0)
      const float u[3] = {0.0f, 1.0f, ess0};
      (void)hcoef[1];
      float Q[3][3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          Q[i][j] = i == j ? 1.0f : 0.0f;
      // applyHouseholderOnTheLeft on the bottom-right 2 x 2 corner of the identity (rows 1..2, cols 1..2)
      const float ess[1] = {u[2]};
      house_left(Q, 1, 1, 2, 2, ess, hcoef[0]);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          U[i][j] = Q[i][j];
    }
    // ---- RealSchur::computeFromHessenberg
    const int max_iters = 40 * size;
    int iu = size - 1, iter = 0, total_iter = 0;
    float exshift = 0.0f;
    float norm = 0.0f;
    for (int j = 0; j < size; j++)
      for (int i = 0; i < std::min(size, j + 2); i++)
        norm += std::fabs(T[i][j]);
    if (norm != 0.0f)
    {
      while (iu >= 0)
      {
        // findSmallSubdiagEntry
        int il = iu;
        while (il > 0)
        {
          float s = std::fabs(T[il - 1][il - 1]) + std::fabs(T[il][il]);
          s = std::max(s * eps, consider_as_zero);
          if (std::fabs(T[il][il - 1]) <= s)
            break;
          il--;
        }
        if (il == iu)  // one root found
        {
          T[iu][iu] = T[iu][iu] + exshift;
          if (iu > 0)
            T[iu][iu - 1] = 0.0f;
          iu--;
          iter = 0;
        }
        else if (il == iu - 1)  // two roots found: splitOffTwoRows
        {
          const float p = 0.5f * (T[iu - 1][iu - 1] - T[iu][iu]);
          const float q = p * p + T[iu][iu - 1] * T[iu - 1][iu];
          T[iu][iu] += exshift;
          T[iu - 1][iu - 1] += exshift;
          if (q >= 0.0f)
          {
            const float z = std::sqrt(std::fabs(q));
            float c, s;
            if (p >= 0.0f)
              make_givens(p + z, T[iu][iu - 1], c, s);
            else
              make_givens(p - z, T[iu][iu - 1], c, s);
            // m_matT.rightCols(size-iu+1).applyOnTheLeft(iu-1, iu, rot.adjoint()): x = row iu-1, y = row iu,
            // (x, y) <- (c x - s y, s x + c y) for the adjoint of (c, s)  [apply_rotation_in_the_plane with (c, -s) conj]
            for (int col = iu - 1; col < size; col++)
            {
              const float x = T[iu - 1][col], y = T[iu][col];
              T[iu - 1][col] = c * x - s * y;
              T[iu][col] = s * x + c * y;
            }
            // m_matT.topRows(iu+1).applyOnTheRight(iu-1, iu, rot): x = col iu-1, y = col iu, (x, y) <- (c x - s y, s x + c y)
            for (int row = 0; row <= iu; row++)
            {
              const float x = T[row][iu - 1], y = T[row][iu];
              T[row][iu - 1] = c * x - s * y;
              T[row][iu] = s * x + c * y;
            }
            T[iu][iu - 1] = 0.0f;
            for (int row = 0; row < size; row++)
            {
              const float x = U[row][iu - 1], y = U[row][iu];
              U[row][iu - 1] = c * x - s * y;
              U[row][iu] = s * x + c * y;
            }
          }
          if (iu > 1)
            T[iu - 1][iu - 2] = 0.0f;
          iu -= 2;
          iter = 0;
        }
        else  // no convergence yet (il == 0, iu == 2 for a 3 x 3)
        {
          float shift[3] = {T[iu][iu], T[iu - 1][iu - 1], T[iu][iu - 1] * T[iu - 1][iu]};
          if (iter == 10)  // Wilkinson's original ad hoc shift
          {
            exshift += shift[0];
            for (int i = 0; i <= iu; i++)
              T[i][i] -= shift[0];
            const float s = std::fabs(T[iu][iu - 1]) + std::fabs(T[iu - 1][iu - 2]);
            shift[0] = 0.75f * s;
            shift[1] = 0.75f * s;
            shift[2] = -0.4375f * s * s;
          }
          if (iter == 30)  // MATLAB's new ad hoc shift
          {
            float s = (shift[1] - shift[0]) / 2.0f;
            s = s * s + shift[2];
            if (s > 0.0f)
            {
              s = std::sqrt(s);
              if (shift[1] < shift[0])
                s = -s;
              s = s + (shift[1] - shift[0]) / 2.0f;
              s = shift[0] - shift[2] / s;
              exshift += s;
              for (int i = 0; i <= iu; i++)
                T[i][i] -= s;
              shift[0] = shift[1] = shift[2] = 0.964f;
            }
          }
          iter++;
          total_iter++;
          if (total_iter > max_iters)
          {
            ok = false;
            break;
          }
          // initFrancisQRStep
          int im;
          float v[3] = {0.0f, 0.0f, 0.0f};
          for (im = iu - 2; im >= il; --im)
          {
            const float Tmm = T[im][im];
            const float r = shift[0] - Tmm;
            const float s = shift[1] - Tmm;
            v[0] = (r * s - shift[2]) / T[im + 1][im] + T[im][im + 1];
            v[1] = T[im + 1][im + 1] - Tmm - r - s;
            v[2] = T[im + 2][im + 1];
            if (im == il)
              break;
            const float lhs = T[im][im - 1] * (std::fabs(v[1]) + std::fabs(v[2]));
            const float rhs = v[0] * (std::fabs(T[im - 1][im - 1]) + std::fabs(Tmm) + std::fabs(T[im + 1][im + 1]));
            if (std::fabs(lhs) < eps * rhs)
              break;
          }
          // performFrancisQRStep
          for (int k = im; k <= iu - 2; ++k)
          {
            const bool first = k == im;
            float w[3];
            if (first)
              for (int i = 0; i < 3; i++)
                w[i] = v[i];
            else
              for (int i = 0; i < 3; i++)
                w[i] = T[k + i][k - 1];
            float ess[2], tau, beta;
            make_householder(w, 3, ess, tau, beta);
            if (beta != 0.0f)
            {
              if (first && k > il)
                T[k][k - 1] = -T[k][k - 1];
              else if (!first)
                T[k][k - 1] = beta;
              house_left(T, k, k, 3, size - k, ess, tau);
              house_right(T, 0, k, std::min(iu, k + 3) + 1, 3, ess, tau);
              house_right(U, 0, k, size, 3, ess, tau);
            }
          }
          {
            float w[2] = {T[iu - 1][iu - 2], T[iu][iu - 2]}, ess[1], tau, beta;
            make_householder(w, 2, ess, tau, beta);
            if (beta != 0.0f)
            {
              T[iu - 1][iu - 2] = beta;
              house_left(T, iu - 1, iu - 1, 2, size - iu + 1, ess, tau);
              house_right(T, 0, iu - 1, iu + 1, 2, ess, tau);
              house_right(U, 0, iu - 1, size, 2, ess, tau);
            }
          }
          for (int i = im + 2; i <= iu; ++i)  // clean up pollution due to round-off errors
          {
            T[i][i - 2] = 0.0f;
            if (i > im + 2)
              T[i][i - 3] = 0.0f;
          }
        }
      }
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        T[i][j] *= scale;
    finish();
  }

  // EigenSolver::compute after the Schur form + doComputeEigenvectors + eigenvectors()
  void finish()
  {
    const int size = 3;
    const float eps = std::numeric_limits<float>::epsilon();
    int i = 0;
    while (i < size)
    {
      if (i == size - 1 || T[i + 1][i] == 0.0f)
      {
        re[i] = T[i][i];
        im[i] = 0.0f;
        ++i;
      }
      else
      {
        const float p = 0.5f * (T[i][i] - T[i + 1][i + 1]);
        float t0 = T[i + 1][i], t1 = T[i][i + 1];
        const float maxval = std::max(std::fabs(p), std::max(std::fabs(t0), std::fabs(t1)));
        t0 /= maxval;
        t1 /= maxval;
        const float p0 = p / maxval;
        const float z = maxval * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
        re[i] = re[i + 1] = T[i + 1][i + 1] + p;
        im[i] = z;
        im[i + 1] = -z;
        i += 2;
      }
    }
    float V[3][3];  // m_eivec
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        V[r][c] = U[r][c];
    float norm = 0.0f;
    for (int j = 0; j < size; j++)
      for (int c = std::max(j - 1, 0); c < size; c++)
        norm += std::fabs(T[j][c]);
    if (norm != 0.0f)
    {
      for (int n = size - 1; n >= 0; n--)
      {
        const float p = re[n], q = im[n];
        if (q == 0.0f)  // scalar vector
        {
          float lastr = 0.0f, lastw = 0.0f;
          int l = n;
          T[n][n] = 1.0f;
          for (int k = n - 1; k >= 0; k--)
          {
            const float w = T[k][k] - p;
            float r = 0.0f;
            for (int c = l; c <= n; c++)
              r += T[k][c] * T[c][n];
            if (im[k] < 0.0f)
            {
              lastw = w;
              lastr = r;
            }
            else
            {
              l = k;
              if (im[k] == 0.0f)
              {
                if (w != 0.0f)
                  T[k][n] = -r / w;
                else
                  T[k][n] = -r / (eps * norm);
              }
              else  // solve real equations
              {
                const float x = T[k][k + 1], y = T[k + 1][k];
                const float denom = (re[k] - p) * (re[k] - p) + im[k] * im[k];
                const float t = (x * lastr - lastw * r) / denom;
                T[k][n] = t;
                if (std::fabs(x) > std::fabs(lastw))
                  T[k + 1][n] = (-r - w * t) / x;
                else
                  T[k + 1][n] = (-lastr - y * t) / lastw;
              }
              const float t = std::fabs(T[k][n]);  // overflow control
              if ((eps * t) * t > 1.0f)
                for (int r2 = k; r2 < size; r2++)
                  T[r2][n] /= t;
            }
          }
        }
        else if (q < 0.0f && n > 0)  // complex vector (columns n-1, n)
        {
          int l = n - 1;
          if (std::fabs(T[n][n - 1]) > std::fabs(T[n - 1][n]))
          {
            T[n - 1][n - 1] = q / T[n][n - 1];
            T[n - 1][n] = -(T[n][n] - p) / T[n][n - 1];
          }
          else
          {
            // (0, -T(n-1,n)) / (T(n-1,n-1) - p, q)
            const float a = 0.0f, b = -T[n - 1][n], c = T[n - 1][n - 1] - p, d = q, den = c * c + d * d;
            T[n - 1][n - 1] = (a * c + b * d) / den;
            T[n - 1][n] = (b * c - a * d) / den;
          }
          T[n][n - 1] = 0.0f;
          T[n][n] = 1.0f;
          for (int k = n - 2; k >= 0; k--)
          {
            float ra = 0.0f, sa = 0.0f;
            for (int c = l; c <= n; c++)
            {
              ra += T[k][c] * T[c][n - 1];
              sa += T[k][c] * T[c][n];
            }
            const float w = T[k][k] - p;
            if (im[k] < 0.0f)
              continue;  // (a second complex pair cannot exist in a 3 x 3)
            l = k;
            if (im[k] == 0.0f)
            {
              const float a = -ra, b = -sa, c = w, d = q, den = c * c + d * d;
              T[k][n - 1] = (a * c + b * d) / den;
              T[k][n] = (b * c - a * d) / den;
            }
            const float t = std::max(std::fabs(T[k][n - 1]), std::fabs(T[k][n]));
            if ((eps * t) * t > 1.0f)
              for (int r2 = k; r2 < size; r2++)
              {
                T[r2][n - 1] /= t;
                T[r2][n] /= t;
              }
          }
          n--;
        }
      }
      // back transformation: m_eivec.col(j) = m_eivec.leftCols(j+1) * m_matT.col(j).head(j+1)
      for (int j = size - 1; j >= 0; j--)
      {
        float tmp[3];
        for (int r = 0; r < 3; r++)
        {
          float acc = 0.0f;
          for (int c = 0; c <= j; c++)
            acc += V[r][c] * T[c][j];
          tmp[r] = acc;
        }
        for (int r = 0; r < 3; r++)
          V[r][j] = tmp[r];
      }
    }
    // eigenvectors(): real columns normalised; a complex pair (j, j+1): (V_j + i V_j+1) / its norm and the conjugate
    const float precision = 2.0f * eps;
    for (int j = 0; j < size; j++)
    {
      const bool is_real = std::fabs(im[j]) <= std::fabs(re[j]) * precision || j + 1 == size;
      if (is_real)
      {
        const float nn = std::sqrt(V[0][j] * V[0][j] + V[1][j] * V[1][j] + V[2][j] * V[2][j]);
        for (int r = 0; r < 3; r++)
          vec_re[r][j] = V[r][j] / nn;
      }
      else
      {
        float sq = 0.0f;
        for (int r = 0; r < 3; r++)
          sq += V[r][j] * V[r][j] + V[r][j + 1] * V[r][j + 1];
        const float nn = std::sqrt(sq);
        for (int r = 0; r < 3; r++)
          vec_re[r][j] = vec_re[r][j + 1] = V[r][j] / nn;
        ++j;
      }
    }
  }
};

// which solver moie() uses: 0 = the EigenSolver restatement above (what PCL calls), 1 = cyclic Jacobi in double (the
// independent cross-check of tests/test_oracle_kat.py).  Test hook: vofod_oracle_set_obb_solver.
inline int& obb_solver()
{
  static int s = 0;
  return s;
}

// moment_of_inertia_estimation.hpp: computeMeanValue, computeCovarianceMatrix (normalised by
// point_mass_ = 1/n^2), computeEigenVectors (sort major >= middle >= minor, right-handed),
// computeOBB (obb max initialised to FLT_MIN as PCL does, position = mean + R*shift).
inline Boxes moie(const std::vector<vofod_point_xyzr>& pts, const std::vector<int>& indices)
{
  Boxes b;
  const size_t n = indices.size();
  float mean[3] = {0, 0, 0};
  for (int a = 0; a < 3; a++)
  {
    b.aabb_min[a] = FLT_MAX;
    b.aabb_max[a] = -FLT_MAX;
  }
  for (const int i : indices)
  {
    const float p[3] = {pts[i].x, pts[i].y, pts[i].z};
    for (int a = 0; a < 3; a++)
    {
      mean[a] += p[a];
      if (p[a] <= b.aabb_min[a])
        b.aabb_min[a] = p[a];
      if (p[a] >= b.aabb_max[a])
        b.aabb_max[a] = p[a];
    }
  }
  const unsigned np = n == 0 ? 1u : static_cast<unsigned>(n);
  for (int a = 0; a < 3; a++)
    mean[a] /= static_cast<float>(np);

  float cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (const int i : indices)
  {
    const float c[3] = {pts[i].x - mean[0], pts[i].y - mean[1], pts[i].z - mean[2]};
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++)
        cov[r][cc] += c[r] * c[cc];
  }
  const float point_mass = 1.0f / static_cast<float>(n * n);
  double A[3][3], w[3], V[3][3];
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++)
    {
      cov[r][cc] *= point_mass;
      A[r][cc] = cov[r][cc];
    }
  if (obb_solver() == 1)
    jacobi_eigen3(A, w, V);
  else
  {
    EigenSolver3f es;
    es.compute(cov);
    for (int k = 0; k < 3; k++)
    {
      w[k] = es.re[k];
      for (int r = 0; r < 3; r++)
        V[r][k] = es.vec_re[r][k];
    }
  }
  int major = 0, middle = 1, minor = 2;
  if (w[major] < w[middle])
    std::swap(major, middle);
  if (w[major] < w[minor])
    std::swap(major, minor);
  if (w[middle] < w[minor])
    std::swap(minor, middle);
  const int order[3] = {major, middle, minor};
  for (int k = 0; k < 3; k++)
  {
    b.eig[k] = static_cast<float>(w[order[k]]);
    float v[3] = {static_cast<float>(V[0][order[k]]), static_cast<float>(V[1][order[k]]), static_cast<float>(V[2][order[k]])};
    const float nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int a = 0; a < 3; a++)
      b.axes[k][a] = v[a] / nrm;
  }
  {
    const float* M = b.axes[0];
    const float* D = b.axes[1];
    const float* N = b.axes[2];
    const float cr[3] = {D[1] * N[2] - D[2] * N[1], D[2] * N[0] - D[0] * N[2], D[0] * N[1] - D[1] * N[0]};
    const float det = M[0] * cr[0] + M[1] * cr[1] + M[2] * cr[2];
    if (det <= 0.0f)
      for (int a = 0; a < 3; a++)
        b.axes[0][a] = -b.axes[0][a];
  }
  for (int a = 0; a < 3; a++)
  {
    b.obb_min[a] = FLT_MAX;
    b.obb_max[a] = FLT_MIN;  // sic (PCL)
  }
  for (const int i : indices)
  {
    const float c[3] = {pts[i].x - mean[0], pts[i].y - mean[1], pts[i].z - mean[2]};
    for (int k = 0; k < 3; k++)
    {
      const float proj = c[0] * b.axes[k][0] + c[1] * b.axes[k][1] + c[2] * b.axes[k][2];
      if (proj <= b.obb_min[k])
        b.obb_min[k] = proj;
      if (proj >= b.obb_max[k])
        b.obb_max[k] = proj;
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
    b.obb_center[a] = mean[a] + (b.axes[0][a] * shift[0] + b.axes[1][a] * shift[1] + b.axes[2][a] * shift[2]);
  return b;
}

// ---------------------------------------------------------- sensor LUT (sim)
// initialize_sensor_lut_simulation, vofod_nodelet.cpp:374-420
inline void sim_lut(const int w, const int h, const float vfov, float* directions)
{
  const double minAngle = 0.0, maxAngle = 2.0 * M_PI;
  const double verticalMinAngle = -vfov / 2.0, verticalMaxAngle = vfov / 2.0;
  const double yAngle_step = (maxAngle - minAngle) / (w - 1);
  const double pAngle_step = (verticalMaxAngle - verticalMinAngle) / (h - 1);
  for (int row = 0; row < h; row++)
    for (int col = 0; col < w; col++)
    {
      const double yAngle = col * yAngle_step + minAngle;
      const double pAngle = row * pAngle_step + verticalMinAngle;
      float* d = directions + 3 * (static_cast<size_t>(row) * w + col);
      d[0] = static_cast<float>(std::cos(pAngle) * std::cos(yAngle));
      d[1] = static_cast<float>(std::cos(pAngle) * std::sin(yAngle));
      d[2] = static_cast<float>(std::sin(pAngle));
    }
}


// initialize_apriori_map (vofod_nodelet.cpp:214-226, 306-345): rigid transform of the loaded cloud, stock
// pcl::VoxelGrid<PointXYZ> centroid filter at the map's voxel size, centroids returned as xyz triples.
// [3P] restated from PCL 1.10 voxel_grid.hpp / Eigen 3.3.7:
//   tf = Identity; tf.rotate(AngleAxisf(yaw, UnitZ)); tf.translate(t + sim_correction)  ->  p' = R*p + R*(t+c)
//   VoxelGrid: min_b = floor(min_p*inv), ijk = floor(x*inv) - min_b, idx = ijk . (1, dx, dx*dy), sort by idx,
//   centroid = float sum of the run / count (order inside a run: input order; std::sort leaves it unspecified).
inline void apriori_points(const std::vector<float>& xyz_in, const float t[3], double yaw_deg, const float corr[3], float leaf, std::vector<float>& out)
{
  out.clear();
  const size_t n = xyz_in.size() / 3;
  if (n == 0)
    return;
  const float angle = static_cast<float>(yaw_deg / 180.0 * M_PI);
  const float c = std::cos(angle), s = std::sin(angle);
  // AngleAxisf::toRotationMatrix() for axis (0,0,1)
  const float R[9] = {c, -s, 0.0f, s, c, 0.0f, 0.0f, 0.0f, ((1.0f - c) * 1.0f) * 1.0f + c};
  const float v[3] = {t[0] + corr[0], t[1] + corr[1], t[2] + corr[2]};
  float tr[3];
  for (int r = 0; r < 3; r++)
    tr[r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1]) + R[3 * r + 2] * v[2];  // translationExt() += linearExt() * v
  std::vector<float> p(3 * n);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < n; i++)
    for (int r = 0; r < 3; r++)
    {
      const float p0 = R[3 * r] * xyz_in[3 * i], p1 = R[3 * r + 1] * xyz_in[3 * i + 1], p2 = R[3 * r + 2] * xyz_in[3 * i + 2];
      const float q = p0 + (p1 + (p2 + tr[r]));  // Transformer::se3
      p[3 * i + r] = q;
      mn[r] = std::min(mn[r], q);
      mx[r] = std::max(mx[r], q);
    }
  const float inv = 1.0f / leaf;
  int min_b[3], div_b[3];
  for (int r = 0; r < 3; r++)
  {
    min_b[r] = static_cast<int>(std::floor(mn[r] * inv));
    div_b[r] = static_cast<int>(std::floor(mx[r] * inv)) - min_b[r] + 1;
  }
  if (static_cast<int64_t>(div_b[0]) * div_b[1] * div_b[2] > 0x7fffffffll)
    return;  // "Leaf size is too small": PCL warns and copies the input; an apriori cloud that large is not supported here
  std::vector<std::pair<uint32_t, uint32_t>> order(n);
  for (size_t i = 0; i < n; i++)
  {
    const int i0 = static_cast<int>(std::floor(p[3 * i] * inv) - static_cast<float>(min_b[0]));
    const int i1 = static_cast<int>(std::floor(p[3 * i + 1] * inv) - static_cast<float>(min_b[1]));
    const int i2 = static_cast<int>(std::floor(p[3 * i + 2] * inv) - static_cast<float>(min_b[2]));
    order[i] = {static_cast<uint32_t>(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]), static_cast<uint32_t>(i)};
  }
  std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  for (size_t a = 0; a < n;)
  {
    size_t b = a;
    float sum[3] = {0, 0, 0};
    while (b < n && order[b].first == order[a].first)
    {
      for (int r = 0; r < 3; r++)
        sum[r] += p[3 * order[b].second + r];
      b++;
    }
    for (int r = 0; r < 3; r++)
      out.push_back(sum[r] / static_cast<float>(b - a));
    a = b;
  }
}

}  // namespace vo
