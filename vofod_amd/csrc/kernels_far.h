// Close-first clustering on the general path (round 4): a single map-updating scan - the reference's own mode.
//
// The batched frame kernel gets a voxel's close bit from the dilated map image; a sensor stream changes the map with every scan,
// so that image (140 us to build) is never valid.  Here the close bits come from hasCloseTo's stencil itself (k_closefar / k_closefar_sweep with
// every voxel as its own "cluster": the own map row first, the other rows of the stencil only for the few voxels that row leaves
// undecided), and only the FAR voxels - those the sweep finds nothing for, typically 1-2 % of a scan - are clustered:
//   k_far_edges   one thread per (far voxel, row of the clustering stencil, direction): the occupancy bitmap window of that row
//                 (as k_union), neighbours within the tolerance (sure by the lattice distance, FLANN's float expression on the
//                 boundary); a CLOSE neighbour taints the voxel, a far one is joined (lock-free union-find, forward rows only);
//   k_far_final   one workgroup: taint of every component, sizes / lattice boxes of the surviving ones (= far_clusters_indices of
//                 vofod_nodelet.cpp:746), their records and candidate members - candidates first, in the canonical order, members
//                 cluster by cluster with ascending rank: the lists k_frame_lds_far leaves, read by the same k_tail_far; a
//                 tainted voxel is close from here on;
//   k_finalize_far updateVMaps (:943-950) voxel by voxel: scores/point + flag 2 for close voxels, scores/unknown + flag 3 for the
//                 voxels of far clusters - what k_finalize does through cluster labels; the map's occupancy image (one bit per
//                 cell, k_mapbits) and its counters are patched where the update flips a cell.
// A far voxel's component is a far cluster iff it has no edge to a close voxel: the argument of kernels_frame.h, on voxels
// instead of bricks (no clique assumption: any tolerance / leaf).  More than FAR_MAX far voxels (a cold map) raise
// CF_RETRY_STATUS: the scan is run again through the full clustering.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_brick_lds.h"
#include "kernels_cluster.h"

namespace vk
{

constexpr uint32_t FAR_MAX = 4096;  // far voxels of a scan the close-first path takes

// taint flag of a far voxel / of a component's root: VoxelArrays::csize (0 from the emission; the surviving roots' sizes later)
__global__ __launch_bounds__(256) void k_far_edges(const GridParams g, const ClusterParams cp, const StencilRow* __restrict__ rows, const FrameHdr* hdrs, const unsigned long long* __restrict__ bitmaps,
                                                   const uint32_t* __restrict__ wprefix_all, VoxelArrays va, const uint32_t* __restrict__ far_list)
{
  const FrameHdr& h = hdrs[0];
  const uint32_t n_far = h.n_far;
  if (n_far > FAR_MAX)
    return;
  const uint32_t per = 2u * static_cast<uint32_t>(cp.n_rows);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t idx = t / per;
  if (idx >= n_far)
    return;
  const uint32_t sub = t - idx * per;
  const int r = static_cast<int>(sub >> 1);
  const bool back = sub & 1u;
  const uint32_t v = far_list[idx];
  const unsigned long long* bm = bitmaps;
  const uint32_t* wprefix = wprefix_all;
  const int dx = h.div_b[0], dy = h.div_b[1], dz = h.div_b[2];
  int i, j, k;
  key_to_ijk(h, va.key[v], i, j, k);
  const StencilRow row = rows[r];
  const int jj = back ? j - row.dj : j + row.dj, kk = back ? k - row.dk : k + row.dk;
  if (jj < 0 || jj >= dy || kk < 0 || kk >= dz)
    return;
  const bool own_row = row.dj == 0 && row.dk == 0;
  int lo = max(i - row.r_max, 0), hi = min(i + row.r_max, dx - 1);
  if (own_row)
  {
    if (back)
      hi = i - 1;
    else
      lo = i + 1;
  }
  if (lo > hi)
    return;
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
    return;
  const unsigned long long win0 = win;
  // rank of the window's first set bit.  With GridParams::sparse_prefix (the brick family's emission) k_emit leaves the prefix
  // entries of all-empty bitmap blocks unwritten: a window may START in the last word of such a block with all its bits in the
  // next word, whose block is not empty - that word's own entry is the rank then (wprefix[wi] was stale memory there)
  const uint32_t pre = w0 ? wprefix[wi] + __popcll(w0 & ((1ull << sh) - 1ull)) : wprefix[wi + 1];
  const float4 pv = va.pts[v];
  bool tainted = false;
  while (win)
  {
    const int tb = __ffsll(static_cast<long long>(win)) - 1;
    win &= win - 1;
    const int adi = abs(lo + tb - i);
    bool ok = adi <= row.r_sure;
    const uint32_t nb = pre + __popcll(win0 & ((1ull << tb) - 1ull));
    if (!ok && ((row.amb >> adi) & 1u))
    {
      const float4 pn = va.pts[nb];
      const float ddx = __fsub_rn(pv.x, pn.x), ddy = __fsub_rn(pv.y, pn.y), ddz = __fsub_rn(pv.z, pn.z);
      float d2 = __fmul_rn(ddx, ddx);
      d2 = __fadd_rn(d2, __fmul_rn(ddy, ddy));
      d2 = __fadd_rn(d2, __fmul_rn(ddz, ddz));
      ok = d2 < cp.r2;
    }
    if (!ok)
      continue;
    if (va.cclose[nb])
      tainted = true;  // an edge to a close voxel: the component is a close cluster
    else if (!back)
      uf_union<0>(va.parent, v, nb);  // (a pair of far voxels is seen from both ends: joined from its base)
  }
  if (tainted)
    __hip_atomic_store(&va.csize[v], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// record of a surviving component (sizes and boxes accumulated in its root's slots)
__device__ __forceinline__ ClusterRec far_record(const GridParams& g, const UpdateParams& up, const VoxelArrays& va, uint32_t root)
{
  ClusterRec rec;
  rec.root = root;
  rec.size = __hip_atomic_load(&va.csize[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int ext_ok = 1;
  for (int a = 0; a < 3; a++)
  {
    rec.imin[a] = __hip_atomic_load(&va.cbox[6 * root + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rec.imax[a] = __hip_atomic_load(&va.cbox[6 * root + 3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ext_ok &= (static_cast<float>(rec.imax[a] - rec.imin[a]) * g.leaf[a] <= up.cand_max_extent);
  }
  rec.close = 0u;
  rec.cand = (static_cast<int>(rec.size) >= up.min_points && ext_ok) ? 1u : 0u;
  return rec;
}

__global__ __launch_bounds__(1024) void k_far_final(const GridParams g, FrameHdr* hdrs, VoxelArrays va, const uint32_t* __restrict__ far_list, const UpdateParams up, ClusterRec* __restrict__ table,
                                                    CandMember* __restrict__ cands)
{
  __shared__ uint32_t s_C, s_nc, s_ncc;
  __shared__ ClusterRec s_rec[TAIL_MAXC];
  __shared__ uint32_t s_oroot[TAIL_MAXC];
  __shared__ unsigned long long s_key[TAIL_MAXM];
  FrameHdr& h = hdrs[0];
  const uint32_t n_far = h.n_far;
  const int tid = threadIdx.x;
  if (n_far > FAR_MAX)
  {
    if (tid == 0)
    {
      h.status = CF_RETRY_STATUS;  // (a cold map) the scan takes the full clustering
      h.C = h.n_cand = 0;
    }
    return;
  }
  if (tid == 0)
    s_C = s_nc = s_ncc = 0;
  constexpr int PT = FAR_MAX / 1024;
  uint32_t vv[PT], rr[PT];
  // 1: roots; a tainted voxel taints its root
#pragma unroll
  for (int q = 0; q < PT; q++)
  {
    const uint32_t idx = q * 1024 + tid;
    vv[q] = rr[q] = 0xffffffffu;
    if (idx < n_far)
    {
      vv[q] = far_list[idx];
      rr[q] = uf_find<0>(va.parent, vv[q]);
      if (rr[q] != vv[q] && __hip_atomic_load(&va.csize[vv[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_store(&va.csize[rr[q]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __threadfence();
  __syncthreads();
  // 2: the components' taint, read by everybody before the surviving roots' slots turn into their sizes
  uint32_t dead = 0;
#pragma unroll
  for (int q = 0; q < PT; q++)
    if (vv[q] != 0xffffffffu && __hip_atomic_load(&va.csize[rr[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      dead |= 1u << q;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PT; q++)
  {
    if (vv[q] == 0xffffffffu)
      continue;
    if ((dead >> q) & 1u)
    {
      va.cclose[vv[q]] = 1u;  // a voxel of a close cluster after all (:946: scores/point, flag 2)
      continue;
    }
    int i, j, k;
    key_to_ijk(h, va.key[vv[q]], i, j, k);
    atomicAdd(&va.csize[rr[q]], 1u);
    int* cb = &va.cbox[6 * rr[q]];
    atomicMin(&cb[0], i), atomicMin(&cb[1], j), atomicMin(&cb[2], k);
    atomicMax(&cb[3], i), atomicMax(&cb[4], j), atomicMax(&cb[5], k);
  }
  __threadfence();
  __syncthreads();
  // 3: records of the surviving components (the root is the smallest member: the canonical label); candidates as k_finalize.
  //    The candidates go to the head of the table in the canonical order (size descending, root ascending: counted among
  //    the <= TAIL_MAXC staged ones) and their members follow cluster by cluster with ascending rank (counted among <= TAIL_MAXM
  //    staged keys): what k_tail_far reads without sorting, as k_frame_lds_far leaves it.  Beyond those capacities the lists
  //    stay unordered: the tail raises its fallback flags on the counts.
#pragma unroll
  for (int q = 0; q < PT; q++)
  {
    if (vv[q] == 0xffffffffu || ((dead >> q) & 1u) || rr[q] != vv[q])
      continue;
    const uint32_t root = vv[q];
    const ClusterRec rec = far_record(g, up, va, root);
    if (!rec.cand)
    {
      va.bb[root] = 0u;  // (the brick codes of this array serve the brick kernels only: free on this path)
      continue;
    }
    const uint32_t slot = atomicAdd(&s_ncc, 1u);
    if (slot < static_cast<uint32_t>(TAIL_MAXC))
      s_rec[slot] = rec;
    else
    {
      table[slot] = rec;
      va.bb[root] = 0x7fffffffu;
    }
  }
  __syncthreads();
  const uint32_t ncc = s_ncc, n_staged = min(ncc, static_cast<uint32_t>(TAIL_MAXC));
  if (tid < static_cast<int>(n_staged))
  {
    const ClusterRec me = s_rec[tid];
    uint32_t ord = 0;
    for (uint32_t u = 0; u < n_staged; u++)
      ord += (s_rec[u].size > me.size || (s_rec[u].size == me.size && s_rec[u].root < me.root)) ? 1u : 0u;
    table[ord] = me;
    s_oroot[ord] = me.root;
    va.bb[me.root] = ord + 1u;
  }
#pragma unroll
  for (int q = 0; q < PT; q++)
  {
    if (vv[q] == 0xffffffffu || ((dead >> q) & 1u) || rr[q] != vv[q])
      continue;
    const ClusterRec rec = far_record(g, up, va, vv[q]);
    if (!rec.cand)
      table[ncc + atomicAdd(&s_C, 1u)] = rec;
  }
  __threadfence();
  __syncthreads();
  // 4: the candidates' members
#pragma unroll
  for (int q = 0; q < PT; q++)
  {
    if (vv[q] == 0xffffffffu || ((dead >> q) & 1u))
      continue;
    const uint32_t o = __hip_atomic_load(&va.bb[rr[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o)
    {
      const uint32_t slot = atomicAdd(&s_nc, 1u);
      if (slot < static_cast<uint32_t>(TAIL_MAXM))
        s_key[slot] = (static_cast<unsigned long long>(o - 1u) << 32) | vv[q];
      else
      {
        CandMember cm;
        cm.root = rr[q];
        cm.v = vv[q];
        cands[slot] = cm;
      }
    }
  }
  __syncthreads();
  const uint32_t n_cand = s_nc;
  if (tid < static_cast<int>(min(n_cand, static_cast<uint32_t>(TAIL_MAXM))))
  {
    const unsigned long long me = s_key[tid];
    const uint32_t o = static_cast<uint32_t>(me >> 32);
    CandMember cm;
    cm.v = static_cast<uint32_t>(me);
    uint32_t pos = tid;
    if (n_cand <= static_cast<uint32_t>(TAIL_MAXM) && ncc <= static_cast<uint32_t>(TAIL_MAXC))
    {
      pos = 0;
      for (uint32_t u = 0; u < n_cand; u++)
        pos += s_key[u] < me ? 1u : 0u;
      cm.root = s_oroot[o];
    }
    else
      cm.root = uf_find<0>(va.parent, cm.v);
    cands[pos] = cm;
  }
  if (tid == 0)
  {
    h.C = ncc + s_C;
    h.n_cand = n_cand;
    h.n_cand_clusters = ncc;
    h.far_only = 1u;
  }
}

// updateVoxel (vofod_nodelet.cpp:777-797) voxel by voxel; the frame's occupancy bitmap is left all-zero for the next call
// `mapbits` (nullable): the map's occupancy image (bit = m > thr_new, k_mapbits) and its partial nVoxelsOver counters are patched
// where this scan's update flips a voxel's bit - a sensor stream then never rebuilds the image (k_mapbits: a 78 MB sweep, 21 us per
// scan) between two scans.
__global__ __launch_bounds__(256) void k_finalize_far(const GridParams g, const MapGeom mg, const UpdateParams up, FrameHdr* hdrs, VoxelArrays va, float* __restrict__ vmap, float* __restrict__ vflags,
                                                      unsigned long long* __restrict__ bitmaps, unsigned long long* __restrict__ mapbits, unsigned long long* __restrict__ bgcount, float thr_new)
{
  FrameHdr& h = hdrs[0];
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= h.V)
    return;
  if (bitmaps)
    bitmaps[va.key[v] >> 6] = 0ull;
  if (h.status == CF_RETRY_STATUS)
    return;  // nothing of this scan is used: the full clustering runs it again (from a clean bitmap)
  const float4 p = va.pts[v];
  const uint32_t close = va.cclose[v];
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
    const float m2 = __fadd_rn(__fmul_rn(w, m), __fmul_rn(__fsub_rn(1.0f, w), score));
    vmap[li] = m2;
    vflags[li] = close ? 2.0f : 3.0f;  // m_vflags_point / m_vflags_unknown (:2336-2337)
    if (mapbits && (m > thr_new) != (m2 > thr_new))
    {
      // (one voxel of the aligned lattice per map cell: this thread is the cell's only writer)
      const unsigned long long bit = 1ull << (li & 63);
      if (m2 > thr_new)
      {
        atomicOr(&mapbits[li >> 6], bit);
        atomicAdd(&bgcount[(li & (MB_SLOTS - 1)) * 8], 1ull);
      }
      else
      {
        atomicAnd(&mapbits[li >> 6], ~bit);
        atomicAdd(&bgcount[(li & (MB_SLOTS - 1)) * 8], ~0ull);  // - 1 (the host sums the partial counters modulo 2^64)
      }
    }
  }
}

}  // namespace vk
