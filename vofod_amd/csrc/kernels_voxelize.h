// Voxelisation kernels (K1-K6 of SURVEY.md §2.1) for gfx950.
//
// Replaces VoxelGridWeighted/VoxelGridCounted::filterImpl (voxel_grid_weighted.cpp:41-190,
// voxel_grid_counted.cpp:49-196) together with the two pcl::CropBox passes and
// pcl::transformPointCloud of filterAndTransform (vofod_nodelet.cpp:621-684).
//
// Design: instead of the reference's sort of (key, point) pairs, occupied cells are marked in a
// per-frame *occupancy bitmap* whose bit order is the reference's key order (x fastest).  A
// prefix sum of the bitmap's word popcounts gives every occupied cell its rank = its position in
// the reference's sorted output, so a plain sweep of the bitmap emits the weighted cloud in the
// reference's order, the per-voxel point counts are integer atomics on rank slots, and the same
// bitmap + prefix array later serve as the O(1) neighbour lookup of the clustering kernel.
// All index arithmetic reproduces the reference's float expressions exactly (no FMA contraction).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace vk
{

__device__ __forceinline__ int f2ord(float f)
{
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// Data-parallel-primitive moves of the gfx9 VALU (no LDS crossbar involved): lanes without a valid source read 0.
//   0x110 + s: row_shr:s (shift right by s lanes inside each row of 16), 0x142: row_bcast:15 (lane 15 of every row to the
//   next row), 0x143: row_bcast:31 (lane 31 to the upper half); row_mask selects the rows that take the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v)
{
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}

// inclusive prefix sum over the 64 lanes of a wave (every lane must be active)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
  v += dpp_mov0<0x111, 0xf>(v);
  v += dpp_mov0<0x112, 0xf>(v);
  v += dpp_mov0<0x114, 0xf>(v);
  v += dpp_mov0<0x118, 0xf>(v);
  v += dpp_mov0<0x142, 0xa>(v);
  v += dpp_mov0<0x143, 0xc>(v);
  return v;
}

// Wave-wide reductions as DPP moves (no LDS crossbar).  A lane without a valid source keeps its own value, which is the
// identity of min / max; sums use the scan above.  The result is valid in every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov_self(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_min(int v)
{
  v = min(v, dpp_mov_self<0x111, 0xf>(v));
  v = min(v, dpp_mov_self<0x112, 0xf>(v));
  v = min(v, dpp_mov_self<0x114, 0xf>(v));
  v = min(v, dpp_mov_self<0x118, 0xf>(v));
  v = min(v, dpp_mov_self<0x142, 0xa>(v));
  v = min(v, dpp_mov_self<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v)
{
  v = max(v, dpp_mov_self<0x111, 0xf>(v));
  v = max(v, dpp_mov_self<0x112, 0xf>(v));
  v = max(v, dpp_mov_self<0x114, 0xf>(v));
  v = max(v, dpp_mov_self<0x118, 0xf>(v));
  v = max(v, dpp_mov_self<0x142, 0xa>(v));
  v = max(v, dpp_mov_self<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }

// Frame/block decode of the 1-D launches that cover n_frames frames with gx blocks each.  (Round 1 also had an XCD-aware
// dealing - every block of frame f on XCD f % 8 - behind a switch: measured 5-9 % slower than the hardware's round robin, removed.)
__device__ __forceinline__ bool frame_block(const GridParams& g, uint32_t& frame, uint32_t& bx, uint32_t& gx)
{
  const uint32_t L = blockIdx.x;
  gx = gridDim.x / g.n_frames;
  frame = L / gx;
  bx = L - frame * gx;
  return true;
}

__device__ __forceinline__ float ldf(const char* base, uint64_t stride, uint32_t i) { return *reinterpret_cast<const float*>(base + static_cast<uint64_t>(i) * stride); }

// CropBox (negative, sensor frame) -> transformPointCloud -> CropBox (positive, world frame):
// vofod_nodelet.cpp:625-655.  Inclusive boxes; non-finite points are dropped.
__device__ __forceinline__ bool fetch_point(const FrameArgs& a, const GridParams& g, uint32_t i, float q[3])
{
  const float px = ldf(a.x, a.stride, i), py = ldf(a.y, a.stride, i), pz = ldf(a.z, a.stride, i);
  if (!(a.flags & FA_SCAN))
  {
    q[0] = px;
    q[1] = py;
    q[2] = pz;
    return true;
  }
  if (!(isfinite(px) && isfinite(py) && isfinite(pz)))
    return false;
  const bool in_ex = !(px < g.ex_min[0] || py < g.ex_min[1] || pz < g.ex_min[2] || px > g.ex_max[0] || py > g.ex_max[1] || pz > g.ex_max[2]);
  if (in_ex)
    return false;
#pragma unroll
  for (int r = 0; r < 3; r++)
  {
    // pcl::detail::Transformer<float>::se3: c0*x + (c1*y + (c2*z + c3)), every op rounded
    const float p0 = __fmul_rn(a.tf[4 * r + 0], px);
    const float p1 = __fmul_rn(a.tf[4 * r + 1], py);
    const float p2 = __fmul_rn(a.tf[4 * r + 2], pz);
    q[r] = __fadd_rn(p0, __fadd_rn(p1, __fadd_rn(p2, a.tf[4 * r + 3])));
  }
  const bool in_op = !(q[0] < g.op_min[0] || q[1] < g.op_min[1] || q[2] < g.op_min[2] || q[0] > g.op_max[0] || q[1] > g.op_max[1] || q[2] > g.op_max[2]);
  return in_op;
}

// voxel_grid_weighted.cpp:131-136
__device__ __forceinline__ uint32_t cell_key(const FrameHdr& h, const GridParams& g, const float q[3])
{
  const int i0 = static_cast<int>(floorf(__fmul_rn(__fsub_rn(q[0], h.offset[0]), g.inv[0])));
  const int i1 = static_cast<int>(floorf(__fmul_rn(__fsub_rn(q[1], h.offset[1]), g.inv[1])));
  const int i2 = static_cast<int>(floorf(__fmul_rn(__fsub_rn(q[2], h.offset[2]), g.inv[2])));
  return static_cast<uint32_t>(i0 + i1 * h.div_b[0] + i2 * h.div_b[0] * h.div_b[1]);
}

__global__ void k_init_hdr(FrameHdr* hdrs, uint32_t* counts2)
{
  FrameHdr& h = hdrs[blockIdx.x];
  if (counts2 && threadIdx.x < 2)
    counts2[2 * blockIdx.x + threadIdx.x] = 0;  // the frame's two list counters of k_key1 (one kernel less in the chain than a memset)
  if (threadIdx.x == 0)
  {
    for (int a = 0; a < 3; a++)
    {
      h.bb_min[a] = 0x7fffffff;
      h.bb_max[a] = static_cast<int>(0x80000000u);
    }
    h.n_in = 0;
    h.status = VOFOD_OK;
    h.n_cells = h.n_words = 0;
    h.V = h.C = h.n_cand = 0;
    h.need_words = 0;
    h.n_bricks = 0;
    h.n_undecided = 0;
    h.n_far = 0;
    h.far_only = 0;
    h.n_cand_clusters = 0;
  }
}

// K1-K4a: crop + transform + crop fused with pcl::getMinMax3D (voxel_grid_weighted.cpp:58).
constexpr int BBOX_PPT = 8;  // points per thread and round
__global__ __launch_bounds__(256) void k_bbox(const FrameArgs* args, const GridParams g, FrameHdr* hdrs)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  const FrameArgs& a = args[FRAME];
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[3] = {static_cast<int>(0x80000000u), static_cast<int>(0x80000000u), static_cast<int>(0x80000000u)};
  uint32_t cnt = 0;
  // a workgroup covers BBOX_PPT * 256 consecutive points per round (one round when the launch has a workgroup per 2048
  // points); the unrolled inner loop keeps all the round's loads in flight
  for (uint32_t base = BX * 256u * BBOX_PPT; base < a.n; base += GX * 256u * BBOX_PPT)
  {
#pragma unroll
    for (int j = 0; j < BBOX_PPT; j++)
    {
      const uint32_t i = base + j * 256u + threadIdx.x;
      float q[3];
      if (i >= a.n || !fetch_point(a, g, i, q))
        continue;
      cnt++;
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        const int o = f2ord(q[c]);
        mn[c] = min(mn[c], o);
        mx[c] = max(mx[c], o);
      }
    }
  }
  cnt = wave_sum(cnt);
#pragma unroll
  for (int c = 0; c < 3; c++)
  {
    mn[c] = wave_min(mn[c]);
    mx[c] = wave_max(mx[c]);
  }
  // block-level combine in LDS, then one set of atomics per block
  __shared__ int s_red[4][7];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
  {
    s_red[wave][0] = static_cast<int>(cnt);
    for (int c = 0; c < 3; c++)
    {
      s_red[wave][1 + c] = mn[c];
      s_red[wave][4 + c] = mx[c];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t tot = 0;
    for (int w = 0; w < 4; w++)
    {
      tot += static_cast<uint32_t>(s_red[w][0]);
      for (int c = 0; c < 3; c++)
      {
        mn[c] = min(mn[c], s_red[w][1 + c]);
        mx[c] = max(mx[c], s_red[w][4 + c]);
      }
    }
    if (tot)
    {
      FrameHdr& h = hdrs[FRAME];
      atomicAdd(&h.n_in, tot);
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        atomicMin(&h.bb_min[c], mn[c]);
        atomicMax(&h.bb_max[c], mx[c]);
      }
    }
  }
}

// voxel_grid_weighted.cpp:61-113: overflow guard, min_b/max_b, offset (+ alignment), div_b.
__device__ inline void grid_of_frame(const GridParams& g, FrameHdr& h)
{
  if (h.n_in == 0)
    return;
  float min_p[3], max_p[3];
  for (int c = 0; c < 3; c++)
  {
    min_p[c] = ord2f(h.bb_min[c]);
    max_p[c] = ord2f(h.bb_max[c]);
  }
  const int64_t dx = static_cast<int64_t>(__fmul_rn(__fsub_rn(max_p[0], min_p[0]), g.inv[0])) + 2;
  const int64_t dy = static_cast<int64_t>(__fmul_rn(__fsub_rn(max_p[1], min_p[1]), g.inv[1])) + 2;
  const int64_t dz = static_cast<int64_t>(__fmul_rn(__fsub_rn(max_p[2], min_p[2]), g.inv[2])) + 2;
  // each factor is < 2^40 for finite floats only when the spans are sane; test stepwise to stay in int64
  const int64_t lim = 0x7fffffffll;
  if (dx > lim || dy > lim || dz > lim || dx * dy > lim || dx * dy * dz > lim)
  {
    h.status = VOFOD_ERR_INDEX_OVERFLOW;
    h.n_in = 0;
    return;
  }
  int64_t cells = 1;
  for (int c = 0; c < 3; c++)
  {
    int min_b = static_cast<int>(floorf(__fmul_rn(min_p[c], g.inv[c])));
    const int max_b = static_cast<int>(floorf(__fmul_rn(max_p[c], g.inv[c])));
    float offset = __fmul_rn(static_cast<float>(min_b), g.leaf[c]);
    if (g.align)
    {
      offset = __fsub_rn(offset, g.aco[c]);
      min_b = static_cast<int>(floorf(__fmul_rn(offset, g.inv[c])));
    }
    h.offset[c] = offset;
    h.min_b[c] = min_b;
    h.div_b[c] = max_b - min_b + 1;
    cells *= h.div_b[c];
  }
  if (cells > lim)
  {
    h.status = VOFOD_ERR_INDEX_OVERFLOW;
    h.n_in = 0;
    return;
  }
  h.n_cells = static_cast<uint32_t>(cells);
  h.n_words = static_cast<uint32_t>((cells + 63) >> 6);
  h.need_words = h.n_words;
  if (h.n_words > g.words_cap)
  {
    h.status = VOFOD_ERR_CAPACITY;
    h.n_in = 0;
    h.n_cells = h.n_words = 0;
  }
}

__global__ void k_grid(const GridParams g, FrameHdr* hdrs)
{
  if (threadIdx.x == 0)
    grid_of_frame(g, hdrs[blockIdx.x]);
}

// Segmented reduction over runs of consecutive lanes holding the same key `w` (LiDAR points arrive in ring
// order, so neighbouring lanes mostly fall into the same voxel / bitmap word).  After the call the first lane
// of every run (`head`) holds the OR (or the sum) of the run.
__device__ __forceinline__ bool run_heads(uint32_t w, int lane, int& end)
{
  const uint32_t prev = __shfl_up(w, 1);
  const bool head = lane == 0 || prev != w;
  const unsigned long long H = __ballot(head);
  const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1));
  end = above ? lane + 1 + (__ffsll(static_cast<long long>(above)) - 1) : 64;
  return head;
}

// K4b: mark occupied cells.  One 64-bit atomic OR per run of lanes that hit the same bitmap word, skipped
// when the bits are already set.
__global__ __launch_bounds__(256) void k_setbits(const FrameArgs* args, const GridParams g, const FrameHdr* hdrs, unsigned long long* bitmaps)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameArgs& a = args[FRAME];
  const FrameHdr& h = hdrs[FRAME];
  if (h.n_in == 0)
    return;
  unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const int lane = threadIdx.x & 63;
  const uint32_t n_round = (a.n + 63u) & ~63u;  // whole waves stay in the loop: the shuffles need every lane
  for (uint32_t i = BX * blockDim.x + threadIdx.x; i < n_round; i += GX * blockDim.x)
  {
    float q[3];
    uint32_t word = 0xffffffffu;
    unsigned long long bits = 0ull;
    if (i < a.n && fetch_point(a, g, i, q))
    {
      const uint32_t key = cell_key(h, g, q);
      if (key < h.n_cells)  // else: rounding artefact outside the lattice (the reference would alias it onto another cell)
      {
        word = key >> 6;
        bits = 1ull << (key & 63);
      }
    }
    int end;
    const bool head = run_heads(word, lane, end);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
    {
      const unsigned long long t = __shfl_down(bits, s);
      if (lane + s < end)
        bits |= t;
    }
    if (head && word != 0xffffffffu && (bm[word] & bits) != bits)
      atomicOr(&bm[word], bits);
  }
}

// ---- exclusive prefix sum of the bitmap's word popcounts (3 phases) ------------------------
constexpr int SCAN_WPT = 4;                    // words per thread
constexpr int SCAN_WPB = 256 * SCAN_WPT;       // words per block
constexpr uint32_t EMIT_SPLIT = 4;             // workgroups of k_emit per block of the lattice (small batches)

// position of the u-th (0-based) set bit of w; u < popcount(w)
__device__ __forceinline__ int nth_set_bit(unsigned long long w, uint32_t u)
{
  int pos = 0;
  uint32_t x = static_cast<uint32_t>(w);
  uint32_t c = __popc(x);
  if (u >= c)
  {
    u -= c;
    x = static_cast<uint32_t>(w >> 32);
    pos = 32;
  }
#pragma unroll
  for (int width = 16; width >= 1; width >>= 1)
  {
    c = __popc(x & ((1u << width) - 1u));
    if (u >= c)
    {
      u -= c;
      x >>= width;
      pos += width;
    }
  }
  return pos;
}

// exclusive scan of one value per thread over a 256-thread block; returns the block total via *total
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* lds4, uint32_t* total)
{
  const uint32_t incl = wave_incl_scan(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 63)
    lds4[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++)
    base += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return base + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_a(const GridParams g, const FrameHdr* hdrs, const unsigned long long* bitmaps, uint32_t* blocksums, uint32_t nblk_cap)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t w0 = BX * SCAN_WPB + threadIdx.x * SCAN_WPT;
  if (BX * SCAN_WPB >= h.n_words)
    return;
  const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < SCAN_WPT; k++)
    if (w0 + k < h.n_words)
      c += __popcll(bm[w0 + k]);
  __shared__ uint32_t lds4[4];
  uint32_t total;
  block_excl_scan_256(c, lds4, &total);
  if (threadIdx.x == 0)
    blocksums[static_cast<size_t>(FRAME) * nblk_cap + BX] = total;
}

// one 1024-thread block per frame: exclusive scan of the block sums, total -> hdr.V
__global__ __launch_bounds__(1024) void k_scan_b(const GridParams g, FrameHdr* hdrs, uint32_t* blocksums, uint32_t nblk_cap)
{
  FrameHdr& h = hdrs[blockIdx.x];
  const uint32_t nblk = (h.n_words + SCAN_WPB - 1) / SCAN_WPB;
  uint32_t* bs = blocksums + static_cast<size_t>(blockIdx.x) * nblk_cap;
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblk; base += 1024)
  {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblk ? bs[i] : 0;
    const uint32_t incl = wave_incl_scan(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63)
      wsum[wave] = incl;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; w++)
      off += wsum[w];
    if (i < nblk)
      bs[i] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry_s = off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    h.V = carry_s;
    if (carry_s > g.vox_cap)
    {
      h.status = VOFOD_ERR_CAPACITY;
      h.V = 0;
    }
  }
}

// Per-voxel device arrays of one frame (struct of arrays, V entries used).
struct VoxelArrays
{
  float4* pts;         // x, y, z, bits(count)  == vofod_point_xyzr
  uint32_t* key;       // lattice key (ascending)
  uint32_t* parent;    // union-find forest, afterwards the label
  uint32_t* csize;     // per root: cluster size
  int32_t* cbox;       // per root: imin[3], imax[3]
  uint32_t* cclose;    // per root: close flag
  uint32_t* bb;        // (4x4x4 brick id << 6) | bit inside the brick: spares the LDS clustering kernel the key decode
};

__device__ __forceinline__ VoxelArrays frame_voxels(const VoxelArrays& base, uint32_t frame, uint32_t vox_cap)
{
  VoxelArrays v;
  const size_t o = static_cast<size_t>(frame) * vox_cap;
  v.pts = base.pts + o;
  v.key = base.key + o;
  v.parent = base.parent + o;
  v.csize = base.csize + o;
  v.cbox = base.cbox + o * 6;
  v.cclose = base.cclose + o;
  v.bb = base.bb + o;
  return v;
}

// ---- brick bookkeeping shared by the emission kernel and kernels_brick.h -------------------------------------
struct BrickParams
{
  int32_t n_off;
  float r2;
  uint32_t bricks_cap;
};

struct BrickArrays
{
  unsigned long long* bricks;  // occupancy words, all-zero outside a call (cleaned after use)
  uint32_t* bparent;
  uint32_t* bmin;   // per brick: smallest voxel rank inside the brick (0xffffffff outside a call)
  uint32_t* bcmin;  // per brick root: smallest voxel rank of the whole component (0xffffffff outside a call)
  uint32_t* blist;  // occupied bricks of the frame (unordered), hdr.n_bricks entries
};

__device__ __forceinline__ BrickArrays frame_bricks(const BrickArrays& base, uint32_t frame, uint32_t bricks_cap, uint32_t vox_cap)
{
  BrickArrays b;
  b.bricks = base.bricks + static_cast<size_t>(frame) * bricks_cap;
  b.bparent = base.bparent + static_cast<size_t>(frame) * bricks_cap;
  b.bmin = base.bmin + static_cast<size_t>(frame) * bricks_cap;
  b.bcmin = base.bcmin + static_cast<size_t>(frame) * bricks_cap;
  b.blist = base.blist + static_cast<size_t>(frame) * vox_cap;
  return b;
}

__device__ __forceinline__ void key_to_ijk(const FrameHdr& h, uint32_t key, int& i, int& j, int& k)
{
  const int dx = h.div_b[0], dxy = h.div_b[0] * h.div_b[1];
  k = key / dxy;
  const int rem = key - k * dxy;
  j = rem / dx;
  i = rem - j * dx;
}

__device__ __forceinline__ uint32_t brick_of(const FrameHdr& h, int i, int j, int k, int& bit)
{
  const int nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2;
  bit = (i & 3) | ((j & 3) << 2) | ((k & 3) << 4);
  return static_cast<uint32_t>(((k >> 2) * nby + (j >> 2)) * nbx + (i >> 2));
}


// register voxel `rank` (lattice cell i,j,k) in its 4x4x4 brick; returns true for the brick's first voxel
__device__ __forceinline__ bool brick_mark(const FrameHdr& h, const BrickArrays& ba, int i, int j, int k, uint32_t rank, uint32_t& b)
{
  int bit;
  b = brick_of(h, i, j, k, bit);
  (void)rank;
  return atomicOr(&ba.bricks[b], 1ull << bit) == 0ull;
}

// append the bricks whose first voxel sits in this wave to the frame's brick list: one counter atomic per wave
__device__ __forceinline__ void brick_append(FrameHdr& h, const BrickArrays& ba, bool first, uint32_t b)
{
  const unsigned long long m = __ballot(first);
  if (!m)
    return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll(static_cast<long long>(m)) - 1;
  uint32_t base = 0;
  if (lane == leader)
    base = atomicAdd(&h.n_bricks, static_cast<uint32_t>(__popcll(m)));
  base = __shfl(base, leader);
  if (first)
  {
    ba.blist[base + __popcll(m & ((1ull << lane) - 1ull))] = b;
    ba.bparent[b] = b;
  }
}

// K6: phase c of the scan fused with the emission of the weighted cloud in key order
// (voxel_grid_weighted.cpp:155-188): centre = (ijk + 0.5)*leaf + offset, weight filled by k_count.
// Emission is load-balanced: the block's words and per-thread rank offsets are staged in LDS and every
// thread then produces output slots t, t+256, ... (locating the owning word by binary search), so the
// voxel records leave the CU as coalesced 16-byte-per-lane stores regardless of how the set bits cluster.
__global__ __launch_bounds__(256) void k_emit(const GridParams g, FrameHdr* hdrs, const unsigned long long* bitmaps, const uint32_t* blocksums,
                                              uint32_t nblk_cap, uint32_t* wprefix_all, VoxelArrays va_all, const BrickParams bp, BrickArrays ba_all, int brick_on, uint32_t init_count, uint32_t split)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  // `split` workgroups share a block of the lattice: each scans the block's words (cheap) and emits its share of the
  // block's voxels - a scan's ground sheet sits in a handful of blocks, whose emission loops set the kernel's time
  const uint32_t SUB = BX % split;
  BX /= split;
  FrameHdr& h = hdrs[FRAME];
  if (BX * SCAN_WPB >= h.n_words || h.V == 0)
    return;
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const uint32_t wbase = BX * SCAN_WPB;
  const uint32_t w0 = wbase + threadIdx.x * SCAN_WPT;
  __shared__ unsigned long long s_words[SCAN_WPB];
  __shared__ uint32_t s_start[256 + 1];
  __shared__ uint32_t lds4[4];
  {
    // most blocks of the lattice hold no voxel at all: their total is known from the scan, so they leave without touching
    // the bitmap.  Their words' prefix entries are only needed by the voxel-level clustering kernel (a neighbour window
    // may start in an empty word); the brick family looks up ranks of set bits only (GridParams::sparse_prefix).
    const uint32_t* bs = blocksums + static_cast<size_t>(FRAME) * nblk_cap;
    const uint32_t nblk = (h.n_words + SCAN_WPB - 1) / SCAN_WPB;
    const uint32_t mine = bs[BX], next = (BX + 1 < nblk) ? bs[BX + 1] : h.V;
    if (next == mine)
    {
      if (!g.sparse_prefix && SUB == 0)
#pragma unroll
        for (int k = 0; k < SCAN_WPT; k++)
          if (w0 + k < h.n_words)
            wprefix[w0 + k] = mine;
      return;
    }
  }
  unsigned long long words[SCAN_WPT];
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < SCAN_WPT; k++)
  {
    words[k] = (w0 + k < h.n_words) ? bm[w0 + k] : 0ull;
    s_words[threadIdx.x * SCAN_WPT + k] = words[k];
    c += __popcll(words[k]);
  }
  uint32_t total;
  const uint32_t excl = block_excl_scan_256(c, lds4, &total);
  const uint32_t base = blocksums[static_cast<size_t>(FRAME) * nblk_cap + BX];
  s_start[threadIdx.x] = excl;
  if (threadIdx.x == 0)
    s_start[256] = total;
  if (SUB == 0)
  {
    uint32_t run = base + excl;
#pragma unroll
    for (int k = 0; k < SCAN_WPT; k++)
    {
      if (w0 + k < h.n_words)
        wprefix[w0 + k] = run;  // also for empty words: a neighbour window may start in one
      run += __popcll(words[k]);
    }
  }
  __syncthreads();
  const int dx = h.div_b[0], dxy = h.div_b[0] * h.div_b[1];
  // this workgroup's share of the block's voxels: [t_lo, t_hi), shares are whole multiples of 256 slots
  const uint32_t share = (((total + split - 1u) / split) + 255u) & ~255u;
  const uint32_t t_lo = SUB * share;
  if (t_lo >= total)
    return;
  const uint32_t t_hi = min(total, t_lo + share);
  const uint32_t t_round = t_lo + ((t_hi - t_lo + 63u) & ~63u);  // whole waves stay in the loop (brick_append shuffles)
  for (uint32_t t = t_lo + threadIdx.x; t < t_round; t += 256)
  {
    bool first = false;
    uint32_t brick = 0;
    if (t < t_hi)
    {
      // owner thread j: s_start[j] <= t < s_start[j+1]
      int lo = 0, hi = 256;
      while (hi - lo > 1)
      {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= t)
          lo = mid;
        else
          hi = mid;
      }
      uint32_t u = t - s_start[lo];
      int wi = lo * SCAN_WPT;
      unsigned long long w = s_words[wi];
      uint32_t pc = __popcll(w);
      while (u >= pc)
      {
        u -= pc;
        w = s_words[++wi];
        pc = __popcll(w);
      }
      const int b = nth_set_bit(w, u);
      const uint32_t key = (wbase + wi) * 64u + b;
      const uint32_t rank = base + t;
      const int k2 = key / dxy;
      const int rem = key - k2 * dxy;
      const int k1 = rem / dx;
      const int k0 = rem - k1 * dx;
      float4 p;
      p.x = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k0), 0.5f), g.leaf[0]), h.offset[0]);
      p.y = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k1), 0.5f), g.leaf[1]), h.offset[1]);
      p.z = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k2), 0.5f), g.leaf[2]), h.offset[2]);
      p.w = __uint_as_float(init_count & 0x7fffffffu);  // weight: 0 when k_count adds every point, 1 when only the extras are added
      va.pts[rank] = p;
      va.key[rank] = key;
      va.bb[rank] = (static_cast<uint32_t>(((k2 >> 2) * ((h.div_b[1] + 3) >> 2) + (k1 >> 2)) * ((h.div_b[0] + 3) >> 2) + (k0 >> 2)) << 6) | static_cast<uint32_t>((k0 & 3) | ((k1 & 3) << 2) | ((k2 & 3) << 4));
      if (!(init_count & 0x80000000u))  // lean emission: the LDS clustering kernel initialises the slots of the roots only
      {
        va.parent[rank] = rank;
        va.csize[rank] = 0;
        va.cclose[rank] = 0;
        int* cb = &va.cbox[6 * rank];
        cb[0] = cb[1] = cb[2] = 0x7fffffff;
        cb[3] = cb[4] = cb[5] = static_cast<int>(0x80000000u);
      }
      if (brick_on)
        first = brick_mark(h, ba, k0, k1, k2, rank, brick);
    }
    if (brick_on)
      brick_append(h, ba, first, brick);
  }
}

__device__ __forceinline__ uint32_t rank_of(const unsigned long long* bm, const uint32_t* wprefix, uint32_t key)
{
  const uint32_t w = key >> 6;
  return wprefix[w] + __popcll(bm[w] & ((1ull << (key & 63)) - 1ull));
}

// K6 weights: number of input points per voxel (voxel_grid_weighted.cpp:181), integer atomics on rank slots,
// one per run of lanes that fall into the same voxel.
// `pt_rank` (nullable) records each input point's voxel rank (0xffffffff when dropped) for the counted grid.
__global__ __launch_bounds__(256) void k_count(const FrameArgs* args, const GridParams g, const FrameHdr* hdrs, const unsigned long long* bitmaps,
                                               const uint32_t* wprefix_all, VoxelArrays va_all, uint32_t* pt_rank, uint32_t pt_cap)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameArgs& a = args[FRAME];
  const FrameHdr& h = hdrs[FRAME];
  if (h.n_in == 0 || h.V == 0)
    return;
  const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const int lane = threadIdx.x & 63;
  const uint32_t n_round = (a.n + 63u) & ~63u;
  for (uint32_t i = BX * blockDim.x + threadIdx.x; i < n_round; i += GX * blockDim.x)
  {
    float q[3];
    uint32_t key = 0xffffffffu;
    if (i < a.n && fetch_point(a, g, i, q))
    {
      key = cell_key(h, g, q);
      if (key >= h.n_cells)
        key = 0xffffffffu;
    }
    int end;
    const bool head = run_heads(key, lane, end);
    uint32_t r = 0xffffffffu;
    if (key != 0xffffffffu)
      r = rank_of(bm, wprefix, key);
    if (head && key != 0xffffffffu)
      atomicAdd(reinterpret_cast<uint32_t*>(&va.pts[r].w), static_cast<uint32_t>(end - lane));
    if (pt_rank && i < a.n)
      pt_rank[static_cast<size_t>(FRAME) * pt_cap + i] = r;
  }
}

}  // namespace vk
