// Euclidean clustering (K7), close/far split (K8, K9) and voxel-map update (K10) kernels for gfx950.
//
// Clustering replaces pcl::EuclideanClusterExtraction (kd-tree radius BFS) called from clusterCloud
// (vofod_nodelet.cpp:689-698).  The points are centres of an integer lattice, so the neighbours of a
// voxel within the tolerance are the set bits of the frame's occupancy bitmap inside a fixed stencil.
// The stencil is a host-built table of (dj,dk) rows; a row's x-run is fetched as one <=63-bit window
// out of two bitmap words, neighbour ids come from the word-prefix array, and components are merged
// with a lock-free union-find that always hooks the larger root under the smaller one, so the final
// label of a component is its smallest member — the same canonical label the oracle uses.
// Offsets whose nominal distance sits on the tolerance boundary are decided by the float expression
// FLANN evaluates (((dx*dx)+dy*dy)+dz*dz < tol*tol) on the actual centres.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_voxelize.h"

namespace vk
{

// MODE 0: agent-scope relaxed loads (L2-served) and stores; 1: agent loads, no path compression;
//      2: volatile loads/stores; 3: plain loads/stores (L1-cached; stale parents are ancestors, the hooking CAS stays coherent)
template <int MODE>
__device__ __forceinline__ uint32_t uf_ld(const uint32_t* p)
{
  if (MODE == 2)
    return *reinterpret_cast<const volatile uint32_t*>(p);
  if (MODE == 3)
  {
    const uint32_t v = *p;  // plain, L1-cached: a stale parent is still an ancestor; the compiler barrier forbids reuse across iterations
    asm volatile("" ::: "memory");
    return v;
  }
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ void uf_st(uint32_t* p, uint32_t v)
{
  if (MODE == 0)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 2)
    *reinterpret_cast<volatile uint32_t*>(p) = v;
  else if (MODE == 3)
    *p = v;
}

// representative with intermediate pointer jumping (parents only ever decrease)
template <int MODE>
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t v)
{
  uint32_t curr = uf_ld<MODE>(&parent[v]);
  if (curr != v)
  {
    uint32_t prev = v, next;
    while (curr > (next = uf_ld<MODE>(&parent[curr])))
    {
      uf_st<MODE>(&parent[prev], next);
      prev = curr;
      curr = next;
    }
  }
  return curr;
}

template <int MODE>
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b)
{
  uint32_t ra = uf_find<MODE>(parent, a), rb = uf_find<MODE>(parent, b);
  while (ra != rb)
  {
    if (ra < rb)
    {
      const uint32_t t = ra;
      ra = rb;
      rb = t;
    }
    // ra > rb: hook ra under rb if ra is still a root
    const uint32_t old = atomicCAS(&parent[ra], ra, rb);
    if (old == ra)
      break;
    ra = old;  // somebody re-parented ra meanwhile: climb and retry
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_union(const GridParams g, const ClusterParams cp, const StencilRow* __restrict__ rows, const FrameHdr* hdrs,
                                               const unsigned long long* bitmaps, const uint32_t* wprefix_all, VoxelArrays va_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t v = BX * blockDim.x + threadIdx.x;
  if (v >= h.V)
    return;
  const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const int dx = h.div_b[0], dy = h.div_b[1], dz = h.div_b[2];
  const uint32_t key = va.key[v];
  const int k = key / (dx * dy);
  const int rem = key - k * dx * dy;
  const int j = rem / dx;
  const int i = rem - j * dx;
  float4 pv = va.pts[v];
  const unsigned long long gapmask = (2ull << cp.row_gap) - 1ull;

  // Distance d to the previous occupied cell of the own row if it lies within the sure in-row gap (0: none).
  // That voxel is in v's component for certain (its own-row pass links it to v) and has already examined
  // every cell of a neighbour row up to x = i - d + r_sure, so v only has to look at the d cells beyond.
  int d = 0;
  if (cp.row_gap > 0 && i > 0)
  {
    const int back = min(cp.row_gap, i);
    const uint32_t L = key - back;
    const uint32_t wi = L >> 6;
    const int sh = L & 63;
    unsigned long long win = (bm[wi] >> sh) | (sh ? (bm[wi + 1] << (64 - sh)) : 0ull);
    win &= (1ull << back) - 1ull;
    if (win)
      d = back - (63 - __clzll(static_cast<long long>(win)));
  }

  for (int r = 0; r < cp.n_rows; r++)
  {
    const StencilRow row = rows[r];
    const int jj = j + row.dj, kk = k + row.dk;
    if (jj < 0 || jj >= dy || kk >= dz)
      continue;
    const bool own_row = (row.dj == 0 && row.dk == 0);
    // cells [i - r_sure, covered] of this row were handled by the previous voxel of the own row
    const int covered = (d > 0 && !own_row) ? i - d + row.r_sure : -0x40000000;
    int lo = own_row ? i + 1 : max(i - row.r_max, 0);
    if (!own_row && d > 0 && row.amb == 0u)
      lo = max(lo, covered + 1);
    const int hi = min(i + row.r_max, dx - 1);
    if (lo > hi)
      continue;
    const uint32_t L = static_cast<uint32_t>((kk * dy + jj) * dx + lo);
    const int nbits = hi - lo + 1;
    const uint32_t wi = L >> 6;
    const int sh = L & 63;
    const unsigned long long w0 = bm[wi];
    unsigned long long win = w0 >> sh;
    if (sh + nbits > 64)
      win |= bm[wi + 1] << (64 - sh);  // guard words are allocated past n_words
    win &= (nbits >= 64) ? ~0ull : ((1ull << nbits) - 1ull);
    if (!win)
      continue;
    const unsigned long long win0 = win;
    const uint32_t pre = wprefix[wi] + __popcll(w0 & ((1ull << sh) - 1ull));
    while (win)
    {
      const int t = __ffsll(static_cast<long long>(win)) - 1;
      const int pos = lo + t;
      const int adi = abs(pos - i);
      bool ok = adi <= row.r_sure;
      if (ok && pos <= covered)
      {
        win &= win - 1;  // already linked through the previous voxel of the own row
        continue;
      }
      const uint32_t nb = pre + __popcll(win0 & ((1ull << t) - 1ull));
      if (!ok && ((row.amb >> adi) & 1u))
      {
        const float4 pn = va.pts[nb];
        const float ddx = __fsub_rn(pv.x, pn.x), ddy = __fsub_rn(pv.y, pn.y), ddz = __fsub_rn(pv.z, pn.z);
        float d2 = __fmul_rn(ddx, ddx);
        d2 = __fadd_rn(d2, __fmul_rn(ddy, ddy));
        d2 = __fadd_rn(d2, __fmul_rn(ddz, ddz));
        ok = d2 < cp.r2;
      }
      if (ok)
      {
        uf_union<MODE>(va.parent, v, nb);
        // cells within the sure in-row gap of this neighbour are linked to it by their own row-(0,0) pass
        win &= ~(gapmask << t);
      }
      else
        win &= win - 1;
    }
  }
}

// Flatten the forest (label = root = smallest member) and accumulate per-cluster size and lattice AABB.
// one (dy,dz) row of hasCloseTo's stencil against the occupancy image
__device__ __forceinline__ bool close_row_hit(const MapGeom& mg, const unsigned long long* __restrict__ mapbits, const CloseRow row, int ox, int oy, int oz)
{
  const int y = oy + row.dy, z = oz + row.dz;
  if (y < 0 || y >= mg.sy || z < 0 || z >= mg.sz)
    return false;
  const int lo = max(ox + row.x_lo, 0), hi = min(ox + row.x_hi, mg.sx - 1);
  if (lo > hi)
    return false;
  const uint64_t L = (static_cast<uint64_t>(z) * mg.sy + y) * mg.sx + lo;
  const int nbits = hi - lo + 1;
  const uint64_t wi = L >> 6;
  const int sh = L & 63;
  unsigned long long win = mapbits[wi] >> sh;
  if (sh + nbits > 64)
    win |= mapbits[wi + 1] << (64 - sh);
  win &= (nbits >= 64) ? ~0ull : ((1ull << nbits) - 1ull);
  return win != 0ull;
}

// Cluster statistics are pre-aggregated per wave (shuffles) and per block (a small LDS hash keyed by root)
// so that the one giant ground cluster does not serialise thousands of global atomics on seven addresses.
constexpr int FL_SLOTS = 64;
template <int SRC>
__global__ __launch_bounds__(256) void k_flatten(const GridParams g, const FrameHdr* hdrs, VoxelArrays va_all, uint32_t* labels_all, const BrickArrays ba_all,
                                                 uint32_t bricks_cap, const MapGeom mg, const unsigned long long* __restrict__ mapclose, const unsigned long long* __restrict__ mapbits,
                                                 const CloseRow* __restrict__ crows, int n_crows)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  __shared__ uint32_t s_root[FL_SLOTS];
  __shared__ uint32_t s_cnt[FL_SLOTS];
  __shared__ int s_box[FL_SLOTS][6];
  for (int t = threadIdx.x; t < FL_SLOTS; t += blockDim.x)
  {
    s_root[t] = 0xffffffffu;
    s_cnt[t] = 0;
    for (int c = 0; c < 3; c++)
    {
      s_box[t][c] = 0x7fffffff;
      s_box[t][3 + c] = static_cast<int>(0x80000000u);
    }
  }
  __syncthreads();
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t v = BX * blockDim.x + threadIdx.x;
  const bool active = v < h.V;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  uint32_t* labels = labels_all + static_cast<size_t>(FRAME) * g.vox_cap;
  uint32_t root = 0xffffffffu;
  int ijk[3] = {0, 0, 0};
  if (active)
  {
    const int dx = h.div_b[0], dxy = h.div_b[0] * h.div_b[1];
    const uint32_t key = va.key[v];
    ijk[2] = key / dxy;
    const int rem = key - ijk[2] * dxy;
    ijk[1] = rem / dx;
    ijk[0] = rem - ijk[1] * dx;
    if (SRC == 0)
    {
      root = v;
      uint32_t p;
      while ((p = va.parent[root]) != root)
        root = p;
    }
    else
    {
      // brick path: k_brick_root left every brick pointing at its representative, whose bcmin is the component's label
      const BrickArrays ba = frame_bricks(ba_all, FRAME, bricks_cap, g.vox_cap);
      int bit;
      const uint32_t b = brick_of(h, ijk[0], ijk[1], ijk[2], bit);
      root = ba.bcmin[ba.bparent[b]];
    }
    labels[v] = root;
  }
  // hasCloseTo through the dilated occupancy image (k_dilate), when the caller has one: the voxel centre is rebuilt with
  // k_emit's expression, its map cell looked up, and one flag store per run of equal roots marks the cluster close
  if (mapclose)
  {
    uint32_t hit_root = 0xffffffffu;
    if (active)
    {
      const float cx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(ijk[0]), 0.5f), g.leaf[0]), h.offset[0]);
      const float cy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(ijk[1]), 0.5f), g.leaf[1]), h.offset[1]);
      const float cz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(ijk[2]), 0.5f), g.leaf[2]), h.offset[2]);
      const int ox = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cx, mg.off[0]), mg.vs_inv)));
      const int oy = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cy, mg.off[1]), mg.vs_inv)));
      const int oz = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cz, mg.off[2]), mg.vs_inv)));
      if (ox >= 0 && ox < mg.sx && oy >= 0 && oy < mg.sy && oz >= 0 && oz < mg.sz)
      {
        const uint64_t L = (static_cast<uint64_t>(oz) * mg.sy + oy) * mg.sx + ox;
        if ((mapclose[L >> 6] >> (L & 63)) & 1ull)
          hit_root = root;
      }
      else  // a centre outside the map (a point on the far face of the operation area): the stencil sweep, clipped to the map
        for (int r = 0; r < n_crows && hit_root == 0xffffffffu; r++)
          if (close_row_hit(mg, mapbits, crows[r], ox, oy, oz))
            hit_root = root;
    }
    int end;
    if (run_heads(hit_root, static_cast<int>(threadIdx.x & 63), end) && hit_root != 0xffffffffu)
      __hip_atomic_store(&va.cclose[hit_root], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // wave level: reduce the wave's leading cluster with shuffles
  const unsigned long long m_active = __ballot(active);
  uint32_t lead = 0xffffffffu;
  bool same = false;
  uint32_t n_same = 0;
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {static_cast<int>(0x80000000u), static_cast<int>(0x80000000u), static_cast<int>(0x80000000u)};
  int lead_lane = -1;
  if (m_active)
  {
    lead_lane = __ffsll(static_cast<long long>(m_active)) - 1;
    lead = __shfl(root, lead_lane);
    same = active && root == lead;
    n_same = __popcll(__ballot(same));
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
      mn[c] = same ? ijk[c] : 0x7fffffff;
      mx[c] = same ? ijk[c] : static_cast<int>(0x80000000u);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1)
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        mn[c] = min(mn[c], __shfl_xor(mn[c], s));
        mx[c] = max(mx[c], __shfl_xor(mx[c], s));
      }
  }
  // block level: one contribution per (wave, leading root) plus the stragglers go through the LDS hash
  const bool is_lead = m_active && static_cast<int>(threadIdx.x & 63) == lead_lane;
  const bool straggler = active && !same;
  if (is_lead || straggler)
  {
    const uint32_t rt = is_lead ? lead : root;
    const uint32_t cnt = is_lead ? n_same : 1u;
    int slot = (rt * 2654435761u) >> 26;  // 64 slots
    bool done = false;
    for (int probe = 0; probe < FL_SLOTS && !done; probe++)
    {
      const uint32_t old = atomicCAS(&s_root[slot], 0xffffffffu, rt);
      if (old == 0xffffffffu || old == rt)
      {
        atomicAdd(&s_cnt[slot], cnt);
#pragma unroll
        for (int c = 0; c < 3; c++)
        {
          atomicMin(&s_box[slot][c], is_lead ? mn[c] : ijk[c]);
          atomicMax(&s_box[slot][3 + c], is_lead ? mx[c] : ijk[c]);
        }
        done = true;
      }
      slot = (slot + 1) & (FL_SLOTS - 1);
    }
    if (!done)  // hash full (more than 64 distinct clusters in one block): go to global memory directly
    {
      atomicAdd(&va.csize[rt], cnt);
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        atomicMin(&va.cbox[6 * rt + c], is_lead ? mn[c] : ijk[c]);
        atomicMax(&va.cbox[6 * rt + 3 + c], is_lead ? mx[c] : ijk[c]);
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < FL_SLOTS; t += blockDim.x)
  {
    const uint32_t rt = s_root[t];
    if (rt == 0xffffffffu)
      continue;
    atomicAdd(&va.csize[rt], s_cnt[t]);
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
      atomicMin(&va.cbox[6 * rt + c], s_box[t][c]);
      atomicMax(&va.cbox[6 * rt + 3 + c], s_box[t][3 + c]);
    }
  }
}

// K8 + occupancy image of the voxel map: bit i = (map[i] > threshold), plus nVoxelsOver
// (voxel_map.cpp:216-222, called per scan at vofod_nodelet.cpp:715).  Streaming, HBM-bound:
// 4 B read per voxel, 1 bit written.  Each wave turns 64 consecutive voxels into one bitmap word
// with a ballot; eight independent coalesced loads are kept in flight per lane.
// Each lane loads 16 bytes (4 voxels), turns them into a nibble, and 16 lanes OR their nibbles into one
// 64-bit word with four xor-shuffles; four independent float4 loads are kept in flight per lane.
constexpr int MB_UNROLL = 4;
constexpr int MB_SLOTS = 64;  // partial counters, 64 B apart
__global__ __launch_bounds__(256) void k_mapbits(const float* __restrict__ map, uint64_t n, float threshold, unsigned long long* __restrict__ bits,
                                                 unsigned long long* __restrict__ count)
{
  const uint64_t n4 = n >> 2;  // whole float4 groups; the tail is handled by the last lanes scalar-wise
  const uint64_t n_words = (n + 63) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t nthreads = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  const float4* map4 = reinterpret_cast<const float4*>(map);
  unsigned long long local = 0;
  // group index gi covers voxels [4*gi, 4*gi+4); a wave covers 256 consecutive voxels = 4 words per load
  const uint64_t n_groups = (n + 3) >> 2;
  const uint64_t n_groups_round = (n_groups + 63) & ~63ull;
  for (uint64_t g0 = tid; g0 < n_groups_round; g0 += nthreads * MB_UNROLL)
  {
    float4 v[MB_UNROLL];
#pragma unroll
    for (int u = 0; u < MB_UNROLL; u++)
    {
      const uint64_t gi = g0 + static_cast<uint64_t>(u) * nthreads;
      if (gi < n4)
        v[u] = map4[gi];
      else
      {
        const uint64_t e = gi << 2;
        v[u].x = e + 0 < n ? map[e + 0] : -INFINITY;
        v[u].y = e + 1 < n ? map[e + 1] : -INFINITY;
        v[u].z = e + 2 < n ? map[e + 2] : -INFINITY;
        v[u].w = e + 3 < n ? map[e + 3] : -INFINITY;
      }
    }
#pragma unroll
    for (int u = 0; u < MB_UNROLL; u++)
    {
      const uint64_t gi = g0 + static_cast<uint64_t>(u) * nthreads;
      const unsigned nib = (v[u].x > threshold ? 1u : 0u) | (v[u].y > threshold ? 2u : 0u) | (v[u].z > threshold ? 4u : 0u) | (v[u].w > threshold ? 8u : 0u);
      unsigned long long w = static_cast<unsigned long long>(nib) << ((lane & 15u) * 4u);
      w |= __shfl_xor(w, 1);
      w |= __shfl_xor(w, 2);
      w |= __shfl_xor(w, 4);
      w |= __shfl_xor(w, 8);
      const uint64_t wi = gi >> 4;
      if ((lane & 15u) == 0 && wi < n_words)
      {
        bits[wi] = w;
        local += __popcll(w);
      }
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1)
    local += __shfl_xor(local, s);
  // one atomic per block, spread over MB_SLOTS counters on separate cache lines: thousands of atomics on ONE address
  // serialise at ~13 ns each and were measured to cost more than the whole sweep (tools/ubench/mapbits.hip)
  __shared__ unsigned long long s_red[4];
  if (lane == 0)
    s_red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const unsigned long long t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (t)
      atomicAdd(&count[(blockIdx.x % MB_SLOTS) * 8], t);
  }
}

// K9: hasCloseTo (voxel_map.cpp:376-400) for every voxel against the occupancy image.
// A cluster is close iff any member is (vofod_nodelet.cpp:727-748).
// Phase A: every lane tests the x-run through its own map cell (in a warm map nearly every background
// voxel hits there).  Phase B: the voxels still undecided are taken one at a time by the whole wave, each
// lane testing rows lane, lane+64, ... of the stencil, so a true negative costs n_rows/64 row tests per
// lane instead of n_rows serial ones and divergence does not hold finished lanes hostage.
// hasCloseTo for every cell of the map at once: out bit = OR of the occupancy image over the cell's stencil, i.e. the
// image dilated by the half-open cube of voxel_map.cpp:384-393.  Worth its cost (one pass of n_rows window tests per 64
// cells) when the map stays unchanged over many frames (batches): k_closefar then answers a voxel with one bit.
// One thread per chunk of 64 x-cells of a map row; `out` must be zero on entry.
__global__ __launch_bounds__(256) void k_dilate(const MapGeom mg, const CloseParams cp, const CloseRow* __restrict__ rows, const unsigned long long* __restrict__ mapbits,
                                                unsigned long long* __restrict__ out)
{
  const uint32_t cx = (mg.sx + 63) >> 6;
  const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= static_cast<uint64_t>(mg.sz) * mg.sy * cx)
    return;
  const int z = static_cast<int>(c / (static_cast<uint64_t>(mg.sy) * cx));
  const uint32_t rem = static_cast<uint32_t>(c - static_cast<uint64_t>(z) * mg.sy * cx);
  const int y = rem / cx;
  const int x0 = static_cast<int>(rem - y * cx) * 64;
  const int nx = min(64, mg.sx - x0);
  const int xb = x0 - 32;  // the 128-bit input window starts 32 cells left of the chunk (|dx| <= MAX_R = 31)
  unsigned long long acc = 0ull;
  for (int r = 0; r < cp.n_rows; r++)
  {
    const CloseRow row = rows[r];
    const int yy = y + row.dy, zz = z + row.dz;
    if (yy < 0 || yy >= mg.sy || zz < 0 || zz >= mg.sz)
      continue;
    const int xs = max(xb, 0), shiftin = xs - xb;  // 0 or 32
    const uint64_t P = (static_cast<uint64_t>(zz) * mg.sy + yy) * mg.sx + xs;
    const uint64_t wi = P >> 6;
    const int sh = P & 63;
    const unsigned long long w0 = mapbits[wi], w1 = mapbits[wi + 1], w2 = mapbits[wi + 2];
    unsigned long long lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
    unsigned long long hi = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
    // cells at or beyond the end of the map row read as empty
    const int valid = mg.sx - xs;  // bits of (hi:lo) that belong to this row
    if (valid < 64)
    {
      lo &= (1ull << valid) - 1ull;
      hi = 0ull;
    }
    else if (valid < 128)
      hi &= (valid == 64) ? 0ull : ((1ull << (valid - 64)) - 1ull);
    if (shiftin)
    {
      hi = (hi << 32) | (lo >> 32);
      lo <<= 32;
    }
    // out bit i (x = x0 + i = xb + 32 + i) |= in bit (32 + i + dx) for dx in [x_lo, x_hi]
    for (int dx = row.x_lo; dx <= row.x_hi; dx++)
    {
      const int s = 32 + dx;  // 1..63
      acc |= (lo >> s) | (hi << (64 - s));
    }
  }
  if (nx < 64)
    acc &= (1ull << nx) - 1ull;
  if (!acc)
    return;
  const uint64_t L = (static_cast<uint64_t>(z) * mg.sy + y) * mg.sx + x0;
  const int so = L & 63;
  atomicOr(&out[L >> 6], acc << so);
  if (so && (acc >> (64 - so)))
    atomicOr(&out[(L >> 6) + 1], acc >> (64 - so));
}

__global__ __launch_bounds__(256) void k_closefar(const GridParams g, const MapGeom mg, const CloseParams cp, const CloseRow* __restrict__ rows,
                                                  const FrameHdr* hdrs, const unsigned long long* __restrict__ mapbits, VoxelArrays va_all,
                                                  const uint32_t* labels_all, const unsigned long long* __restrict__ mapclose, CandMember* __restrict__ undecided_all,
                                                  FrameHdr* hdrs_w)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t v = BX * blockDim.x + threadIdx.x;
  const bool active = v < h.V;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const int lane = threadIdx.x & 63;
  uint32_t root = 0, hit_root = 0xffffffffu;
  int ox = 0, oy = 0, oz = 0;
  bool undecided = false;
  if (active)
    root = labels_all ? labels_all[static_cast<size_t>(FRAME) * g.vox_cap + v] : v;  // (no labels: every voxel for itself - kernels_far.h)
  // the cluster's "already close" flag is read once per run of equal roots and handed to the run's lanes
  uint32_t flag = 0;
  {
    const uint32_t key = active ? root : 0xffffffffu;
    const uint32_t prev = __shfl_up(key, 1);
    const bool head = lane == 0 || prev != key;
    const unsigned long long H = __ballot(head);
    if (head && active)
      flag = __hip_atomic_load(&va.cclose[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long upto = H & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int head_lane = 63 - __clzll(static_cast<long long>(upto));
    flag = __shfl(flag, head_lane);
  }
  if (active && !flag)
  {
    const float4 p = va.pts[v];
    // coordToIdx voxel_map.cpp:592-599
    ox = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.x, mg.off[0]), mg.vs_inv)));
    oy = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.y, mg.off[1]), mg.vs_inv)));
    oz = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.z, mg.off[2]), mg.vs_inv)));
    if (mapclose && ox >= 0 && ox < mg.sx && oy >= 0 && oy < mg.sy && oz >= 0 && oz < mg.sz)
    {
      // the dilated image (k_dilate) holds the whole answer for cells of the map
      const uint64_t L = (static_cast<uint64_t>(oz) * mg.sy + oy) * mg.sx + ox;
      if ((mapclose[L >> 6] >> (L & 63)) & 1ull)
        hit_root = root;
    }
    else if (cp.n_rows > 0 && close_row_hit(mg, mapbits, rows[0], ox, oy, oz))  // rows[0] is (dy,dz) = (0,0): nearest first
      hit_root = root;
    else
      undecided = true;
  }
  {
    // consecutive voxels mostly share the cluster: one flag store per run of equal roots, not one per lane
    int end;
    if (run_heads(hit_root, lane, end) && hit_root != 0xffffffffu)
      __hip_atomic_store(&va.cclose[hit_root], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // Phase B.  With a list (single frames: the launch has too few waves to hide the sweep's latency) the undecided voxels
  // are handed to k_closefar_sweep, one reservation per wave; otherwise they are swept here.
  unsigned long long todo = __ballot(undecided);
  if (undecided_all)
  {
    if (todo)
    {
      const int leader = __ffsll(static_cast<long long>(todo)) - 1;
      uint32_t base = 0;
      if (lane == leader)
        base = atomicAdd(&hdrs_w[FRAME].n_undecided, static_cast<uint32_t>(__popcll(todo)));
      base = __shfl(base, leader);
      if (undecided)
      {
        CandMember cm;
        cm.root = root;
        cm.v = v;
        undecided_all[static_cast<size_t>(FRAME) * g.vox_cap + base + __popcll(todo & ((1ull << lane) - 1ull))] = cm;
      }
    }
    return;
  }
  // four undecided voxels at a time, 16 lanes each sweeping the stencil rows
  const int grp = lane >> 4, sub = lane & 15;
  while (todo)
  {
    // pick up to four source lanes
    int src = -1;
    unsigned long long rest = todo;
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
      const int s = rest ? __ffsll(static_cast<long long>(rest)) - 1 : -1;
      if (rest)
        rest &= rest - 1;
      if (q == grp)
        src = s;
    }
    todo = rest;
    const int srcl = src < 0 ? 0 : src;
    const uint32_t r_root = __shfl(root, srcl);
    const int sx_ = __shfl(ox, srcl), sy_ = __shfl(oy, srcl), sz_ = __shfl(oz, srcl);
    // rows are ordered nearest first; a group stops as soon as any of its lanes has found an occupied cell
    const unsigned long long gmask = 0xffffull << (grp * 16);
    bool done = src < 0, found = false;
    for (int r0 = 1; r0 < cp.n_rows; r0 += 16)
    {
      bool hit = false;
      const int r = r0 + sub;
      if (!done && r < cp.n_rows)
        hit = close_row_hit(mg, mapbits, rows[r], sx_, sy_, sz_);
      if (__ballot(hit) & gmask)
      {
        done = true;
        found = true;
      }
      if (!__ballot(!done))
        break;
    }
    if (found && sub == 0)
      __hip_atomic_store(&va.cclose[r_root], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The stencil sweep of the voxels k_closefar could not decide from their own row: 16 lanes per voxel over the nearest-first
// rows, as many workgroups as the list needs (a kernel of its own so that a single frame gets thousands of waves).
__global__ __launch_bounds__(256) void k_closefar_sweep(const GridParams g, const MapGeom mg, const CloseParams cp, const CloseRow* __restrict__ rows, const FrameHdr* hdrs,
                                                        const unsigned long long* __restrict__ mapbits, VoxelArrays va_all, const CandMember* __restrict__ undecided_all,
                                                        uint32_t* __restrict__ far_list = nullptr, FrameHdr* hdrs_w = nullptr, uint32_t far_cap = 0)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const uint32_t n = hdrs[FRAME].n_undecided;
  const uint32_t u = (BX * blockDim.x + threadIdx.x) >> 4;
  if ((BX * blockDim.x >> 4) >= n)
    return;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const bool live = u < n;
  uint32_t root = 0;
  int ox = 0, oy = 0, oz = 0;
  bool done = !live;
  if (live)
  {
    const CandMember cm = undecided_all[static_cast<size_t>(FRAME) * g.vox_cap + u];
    root = cm.root;
    const float4 p = va.pts[cm.v];
    ox = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.x, mg.off[0]), mg.vs_inv)));
    oy = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.y, mg.off[1]), mg.vs_inv)));
    oz = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.z, mg.off[2]), mg.vs_inv)));
    done = __hip_atomic_load(&va.cclose[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;  // the cluster is close already
  }
  const unsigned long long gmask = 0xffffull << (grp * 16);
  bool found = false;
  for (int r0 = 1; r0 < cp.n_rows; r0 += 16)  // row 0 was tested by k_closefar
  {
    bool hit = false;
    const int r = r0 + sub;
    if (!done && r < cp.n_rows)
      hit = close_row_hit(mg, mapbits, rows[r], ox, oy, oz);
    if (__ballot(hit) & gmask)
    {
      done = true;
      found = true;
    }
    if (!__ballot(!done))
      break;
  }
  if (found && sub == 0)
    __hip_atomic_store(&va.cclose[root], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (close-first on the general path, kernels_far.h: with every voxel as its own cluster, a voxel nothing was found for is FAR)
  if (far_list && live && !found && sub == 0)
  {
    const uint32_t pos = atomicAdd(&hdrs_w[FRAME].n_far, 1u);
    if (pos < far_cap)
      far_list[pos] = undecided_all[static_cast<size_t>(FRAME) * g.vox_cap + u].v;
  }
}

// K10 + cluster table + candidate members.
//   - updateVoxel (vofod_nodelet.cpp:777-797) with score/flag chosen by the cluster's close flag (:946-948)
//   - every root appends its cluster record (unordered; the host puts the table in canonical order)
//   - voxels of far clusters that can still pass the min_points/max_size gates (:1679-1690) are
//     appended to the candidate member list the host-side classification consumes.
__global__ __launch_bounds__(256) void k_finalize(const GridParams g, const MapGeom mg, const UpdateParams up, FrameHdr* hdrs, VoxelArrays va_all,
                                                  const uint32_t* labels_all, float* __restrict__ vmap, float* __restrict__ vflags, ClusterRec* table_all,
                                                  CandMember* cand_all, unsigned long long* __restrict__ bitmaps)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  FrameHdr& h = hdrs[FRAME];
  const uint32_t v = BX * blockDim.x + threadIdx.x;
  if (v >= h.V)
    return;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const uint32_t root = labels_all[static_cast<size_t>(FRAME) * g.vox_cap + v];
  const uint32_t close = va.cclose[root];
  const uint32_t size = va.csize[root];
  const float4 p = va.pts[v];
  // last consumer of the frame's occupancy bitmap: leave it all-zero for the next call (no 2.5 MB memset per frame)
  if (bitmaps)
    bitmaps[static_cast<size_t>(FRAME) * (g.words_cap + 2) + (va.key[v] >> 6)] = 0ull;
  if (!up.no_update)
  {
    const int ox = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.x, mg.off[0]), mg.vs_inv)));
    const int oy = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.y, mg.off[1]), mg.vs_inv)));
    const int oz = static_cast<int>(floorf(__fmul_rn(__fsub_rn(p.z, mg.off[2]), mg.vs_inv)));
    if (ox < 0 || ox >= mg.sx || oy < 0 || oy >= mg.sy || oz < 0 || oz >= mg.sz)
      h.status = VOFOD_ERR_MAP_RANGE;
    else
    {
      const uint64_t li = (static_cast<uint64_t>(oz) * mg.sy + oy) * mg.sx + ox;
      const uint32_t c = min(__float_as_uint(p.w), 63u);
      const float w = __uint_as_float((127u - c) << 23);  // 1.0f / float(1lu << c), exact
      const float score = close ? up.score_point : up.score_unknown;
      const float m = vmap[li];
      vmap[li] = __fadd_rn(__fmul_rn(w, m), __fmul_rn(__fsub_rn(1.0f, w), score));
      vflags[li] = close ? 2.0f : 3.0f;  // m_vflags_point / m_vflags_unknown (:2336-2337)
    }
  }
  int ext_ok = 1;
  const int32_t* box = &va.cbox[6 * root];
#pragma unroll
  for (int c = 0; c < 3; c++)
    ext_ok &= (static_cast<float>(box[3 + c] - box[c]) * g.leaf[c] <= up.cand_max_extent);
  const bool cand = !close && static_cast<int>(size) >= up.min_points && ext_ok;
  if (root == v)
  {
    const uint32_t slot = atomicAdd(&h.C, 1u);
    ClusterRec rec;
    rec.root = root;
    rec.size = size;
    for (int c = 0; c < 3; c++)
    {
      rec.imin[c] = box[c];
      rec.imax[c] = box[3 + c];
    }
    rec.close = close;
    rec.cand = cand ? 1u : 0u;
    table_all[static_cast<size_t>(FRAME) * g.vox_cap + slot] = rec;
  }
  if (cand)
  {
    const uint32_t s = atomicAdd(&h.n_cand, 1u);
    CandMember cm;
    cm.root = root;
    cm.v = v;
    cand_all[static_cast<size_t>(FRAME) * g.vox_cap + s] = cm;
  }
}

// small helpers used by the host-side classification tail -----------------------------------------

// copy a clamped sub-box of a map into a dense staging buffer (x fastest)
__global__ void k_read_box(const float* __restrict__ map, const MapGeom mg, int x0, int y0, int z0, int nx, int ny, int nz, float* __restrict__ dst)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = static_cast<uint32_t>(nx) * ny * nz;
  if (t >= n)
    return;
  const int x = t % nx, y = (t / nx) % ny, z = t / (nx * ny);
  dst[t] = map[(static_cast<uint64_t>(z0 + z) * mg.sy + (y0 + y)) * mg.sx + (x0 + x)];
}

__global__ void k_scatter_set(float* __restrict__ map, const uint64_t* __restrict__ idx, uint32_t n, float value)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n)
    map[idx[t]] = value;
}

__global__ void k_fill(float* __restrict__ p, uint64_t n, float v)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

// initialize_apriori_map vofod_nodelet.cpp:339-341
__global__ void k_apriori(float* __restrict__ map, const MapGeom mg, const float* __restrict__ xyz, uint32_t n)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n)
    return;
  const int ox = static_cast<int>(floorf(__fmul_rn(__fsub_rn(xyz[3 * t + 0], mg.off[0]), mg.vs_inv)));
  const int oy = static_cast<int>(floorf(__fmul_rn(__fsub_rn(xyz[3 * t + 1], mg.off[1]), mg.vs_inv)));
  const int oz = static_cast<int>(floorf(__fmul_rn(__fsub_rn(xyz[3 * t + 2], mg.off[2]), mg.vs_inv)));
  if (ox < 0 || ox >= mg.sx || oy < 0 || oy >= mg.sy || oz < 0 || oz >= mg.sz)
    return;
  map[(static_cast<uint64_t>(oz) * mg.sy + oy) * mg.sx + ox] = INFINITY;
}

}  // namespace vk
