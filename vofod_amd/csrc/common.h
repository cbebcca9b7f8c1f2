// Shared host/device structures of the HIP implementation (gfx950 only).
#pragma once
#include <stdint.h>

#include "../../include/vofod.h"

namespace vk
{

constexpr int WAVE = 64;
constexpr int MAX_STENCIL_ROWS = 4096;  // (2R+1)^2/2 rows for R <= 31 would be 1985; hasCloseTo rows (2d)^2 <= 3969
constexpr int MAX_R = 31;               // neighbour windows are at most 63 bits wide
constexpr int TAIL_MAXM = 1024;         // device classification tail: candidate members per frame
constexpr int TAIL_MAXC = 64;           // ... candidate clusters per frame (one lane each)

// Per-frame launch arguments (host -> device once per call).
struct FrameArgs
{
  const char* x;  // strided float columns (sensor frame for scans)
  const char* y;
  const char* z;
  const char* intensity;  // counted grid only
  uint64_t stride;
  uint32_t n;      // points in the input cloud
  uint32_t flags;  // bit0: apply crop/transform (scan path); bit1: counted grid
  float tf[12];    // row-major 3x4
};
enum : uint32_t { FA_SCAN = 1u, FA_COUNTED = 2u };

// Per-frame device header: reductions, the voxel-grid lattice and counters.
struct FrameHdr
{
  int32_t bb_min[3];  // order-preserving int image of the float bbox (getMinMax3D)
  int32_t bb_max[3];
  uint32_t n_in;      // points entering the voxel grid (after the crops)
  int32_t status;     // vofod_status raised on the device
  // lattice (voxel_grid_weighted.cpp:72-113)
  float offset[3];
  int32_t min_b[3];
  int32_t div_b[3];
  uint32_t n_cells;
  uint32_t n_words;
  // outputs
  uint32_t V;       // occupied voxels
  uint32_t C;       // clusters
  uint32_t n_cand;  // voxels of candidate (far, small) clusters appended to the member list
  uint32_t need_words;  // bitmap words the lattice needs (reported even when it exceeds the workspace)
  uint32_t n_bricks;    // occupied 4x4x4 bricks (brick-level clustering)
  uint32_t n_undecided; // voxels whose own map row is empty: k_closefar_sweep tests their whole stencil
  uint32_t n_far;       // (close-first on the general path) voxels the sweep found no background voxel for
  uint32_t far_only;         // k_frame_lds clustered close first: the cluster table holds the far clusters only, no labels were written
  uint32_t n_cand_clusters;  // ... and begins with this many candidate clusters in the canonical order, their members sorted (k_tail_far)
};

// Parameters constant over a call (passed by value).
struct GridParams
{
  float leaf[3];
  float inv[3];
  float aco[3];   // align_corner_offset (voxel_grid_weighted.cpp:86-97), host-computed
  int32_t align;  // setVoxelAlign was called
  float ex_min[3], ex_max[3];  // exclude box, sensor frame  (vofod_nodelet.cpp:626-629)
  float op_min[3], op_max[3];  // operation area, world frame (vofod_nodelet.cpp:645-648)
  uint32_t words_cap;          // bitmap words available per frame
  uint32_t vox_cap;            // voxel records available per frame
  uint32_t n_frames;           // frames covered by the launch
  uint32_t sparse_prefix;      // 1: word-prefix entries of all-empty bitmap blocks are not written (brick clustering only)
};

// One (dj,dk) row of the Euclidean-clustering half stencil.
struct StencilRow
{
  int16_t dj, dk;
  int16_t r_sure;   // |di| <= r_sure is connected for certain (-1: none)
  int16_t r_max;    // largest |di| that may be connected (sure or ambiguous)
  uint32_t amb;     // bit a set: |di| == a sits on the tolerance boundary -> evaluate in float
};

struct ClusterParams
{
  int32_t n_rows;
  int32_t row_gap;  // r_sure of the (0,0) row: in-row bits that close together are connected for certain
  float r2;         // tolerance*tolerance (float product, as FLANN receives it)
};

// One (dy,dz) row of hasCloseTo's half-open cube (voxel_map.cpp:384-393).
struct CloseRow
{
  int16_t dy, dz;
  int16_t x_lo, x_hi;  // inclusive dx range of cells passing the truncated-norm test
};

struct MapGeom
{
  float off[3];
  float vs, vs_inv;
  int32_t sx, sy, sz;
  uint64_t n;  // sx*sy*sz
};

struct CloseParams
{
  int32_t n_rows;
  float threshold;
};

struct UpdateParams
{
  float score_point, score_unknown;
  int32_t min_points;      // classification__min_points
  float cand_max_extent;   // AABB extent above which a far cluster can never pass the max_size gate
  int32_t no_update;       // VOFOD_SCAN_NO_MAP_UPDATE
};

// Cluster table entry produced on the device (unordered; the host puts it in canonical order).
struct ClusterRec
{
  uint32_t root;   // smallest member index (= label)
  uint32_t size;
  int32_t imin[3], imax[3];  // lattice AABB
  uint32_t close;
  uint32_t cand;   // offset of this cluster's slot range in the candidate member list, or 0xffffffff
};

// Candidate member: a voxel of a far cluster small enough to reach classification.
struct CandMember
{
  uint32_t root;
  uint32_t v;
};

}  // namespace vk
