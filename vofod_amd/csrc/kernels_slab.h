// Occupancy bitmap built in LDS slabs (gfx950) - the fast form of K4b/K5a/K6-weights for lattices of moderate size.
//
// k_setbits marks cells with one global 64-bit atomic per point.  A 128-ring scan at 0.25 m leaves ~40 k points
// in a lattice of 11 M cells: every atomic is a read-modify-write of a cache line nobody else touches, so the kernel
// runs at the pace of random 64-byte HBM accesses (measured: 2.7 us per frame, flat in the batch size).  Here
//   k_key   computes the cell key of every surviving point once and appends it to the frame's key list
//           (block-level compaction: one global atomic per 8192 points);
//   k_slab  gives every (frame, 1 Mi-cell slab of the lattice) one workgroup: the slab's bitmap lives in 128 KB of LDS,
//           the frame's keys (a few hundred KB, L2-resident) are scanned, bits are set with LDS atomics, and the slab
//           leaves the CU as coalesced 8-byte stores together with its per-256-word popcounts (phase a of the scan).
//           A point that finds its bit already set is an "extra": only those (points - voxels, ~13 % here) are counted
//           afterwards; k_emit starts every voxel's weight at 1;
//   k_count_extras adds the extras to their voxels' weights through the rank lookup.
// The global bitmap is written densely (zeros included), so it needs no clearing before or after.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_voxelize.h"

namespace vk
{

#ifndef VOFOD_KEY_THREADS
#define VOFOD_KEY_THREADS 256  /* 256 x 8: 121 us per 256 frames; 512 x 8: 149; 1024 x 8: 160; 256 x 16: 164 */
#endif
#ifndef VOFOD_KEY_PPT
#define VOFOD_KEY_PPT 8
#endif
constexpr int KEY_THREADS = VOFOD_KEY_THREADS;
constexpr int KEY_PPT = VOFOD_KEY_PPT;  // points per thread
constexpr int SLAB_THREADS = 1024;
constexpr uint32_t SLAB_WORDS64 = 16384;              // 64-bit bitmap words per slab = 128 KB of LDS
constexpr uint32_t SLAB_CELLS = SLAB_WORDS64 * 64u;   // 1 Mi cells
constexpr int SLAB_EXTRA_CAP = 6144;                  // extras staged in LDS per workgroup

struct SlabArrays
{
  uint32_t* keys;    // [F][pt_cap] surviving points' cell keys, unordered
  uint32_t* extras;  // [F][pt_cap] keys of points that were not the first of their voxel
  uint32_t* counts;  // [F][2]: n_keys, n_extras
};

__global__ __launch_bounds__(KEY_THREADS) void k_key(const FrameArgs* args, const GridParams g, const FrameHdr* hdrs, SlabArrays sa, uint32_t pt_cap)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameArgs& a = args[FRAME];
  const FrameHdr& h = hdrs[FRAME];
  if (h.n_in == 0)
    return;
  const uint32_t base_pt = BX * KEY_THREADS * KEY_PPT;
  if (base_pt >= a.n)
    return;
  uint32_t key[KEY_PPT];
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < KEY_PPT; j++)
  {
    const uint32_t i = base_pt + j * KEY_THREADS + threadIdx.x;
    key[j] = 0xffffffffu;
    float q[3];
    if (i < a.n && fetch_point(a, g, i, q))
    {
      const uint32_t k = cell_key(h, g, q);
      if (k < h.n_cells)  // else: rounding artefact outside the lattice, dropped as in k_setbits / k_count
        key[j] = k;
    }
    cnt += key[j] != 0xffffffffu;
  }
  // block-level exclusive scan of the per-thread counts
  __shared__ uint32_t s_wsum[KEY_THREADS / 64];
  __shared__ uint32_t s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(cnt);
  if (lane == 63)
    s_wsum[wave] = incl;
  __syncthreads();
  uint32_t off = incl - cnt, total = 0;
  for (int w = 0; w < KEY_THREADS / 64; w++)
  {
    const uint32_t x = s_wsum[w];
    off += w < wave ? x : 0u;
    total += x;
  }
  if (threadIdx.x == 0)
    s_base = total ? atomicAdd(&sa.counts[2 * FRAME], total) : 0u;
  __syncthreads();
  uint32_t* out = sa.keys + static_cast<size_t>(FRAME) * pt_cap + s_base + off;
#pragma unroll
  for (int j = 0; j < KEY_PPT; j++)
    if (key[j] != 0xffffffffu)
      *out++ = key[j];
}

constexpr int SLAB_KR = 48;  // keys a lane keeps in registers across the slabs of its workgroup

// One workgroup serves the slabs GI, GI + G, ... of a frame.  The frame's keys are fetched once into registers (48 Ki
// keys; a longer list is re-read per slab), so with G = 1 (large batches: one workgroup per frame keeps every CU busy)
// the key list crosses the memory system once instead of once per slab.
__global__ __launch_bounds__(SLAB_THREADS) void k_slab(const GridParams g, const FrameHdr* hdrs, SlabArrays sa, uint32_t pt_cap, unsigned long long* __restrict__ bitmaps,
                                                      uint32_t* __restrict__ blocksums, uint32_t nblk_cap)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[SLAB_WORDS64 * 2];
  __shared__ uint32_t s_extra[SLAB_EXTRA_CAP];
  __shared__ uint32_t s_bsum[SLAB_WORDS64 / SCAN_WPB];
  __shared__ uint32_t s_ne, s_gbase;
  uint32_t FRAME, GI, G;
  if (!frame_block(g, FRAME, GI, G))
    return;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t n_slabs = (h.n_words + SLAB_WORDS64 - 1) / SLAB_WORDS64;
  if (GI >= n_slabs)
    return;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t n_keys = sa.counts[2 * FRAME];
  const uint32_t* keys = sa.keys + static_cast<size_t>(FRAME) * pt_cap;
  uint32_t* extras = sa.extras + static_cast<size_t>(FRAME) * pt_cap;
  const uint32_t n_round = (n_keys + 63u) & ~63u;  // whole waves: the extras are appended with a ballot
  uint32_t kreg[SLAB_KR];
#pragma unroll
  for (int j = 0; j < SLAB_KR; j++)
  {
    const uint32_t i = j * SLAB_THREADS + tid;
    kreg[j] = i < n_keys ? keys[i] : 0xffffffffu;
  }
  for (uint32_t SLAB = GI; SLAB < n_slabs; SLAB += G)
  {
    const uint32_t w_first = SLAB * SLAB_WORDS64;
    const uint32_t n_w = min(SLAB_WORDS64, h.n_words - w_first);
    __syncthreads();  // the previous slab has left the LDS
    for (uint32_t i = tid; i < SLAB_WORDS64 / 2; i += SLAB_THREADS)
      reinterpret_cast<uint4*>(s_bits)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < static_cast<int>(SLAB_WORDS64 / SCAN_WPB))
      s_bsum[tid] = 0u;
    if (tid == 0)
      s_ne = 0u;
    __syncthreads();
    const uint32_t cell0 = SLAB * SLAB_CELLS;
    // set the bit; true when it was set already (the point is an "extra" of its voxel)
    auto mark = [&](uint32_t kv) -> bool {
      const uint32_t local = kv - cell0;  // wraps to a huge value for keys below the slab and for the filler
      if (kv == 0xffffffffu || local >= SLAB_CELLS)
        return false;
      const uint32_t bit = 1u << (local & 31u);
      return (atomicOr(&s_bits[local >> 5], bit) & bit) != 0u;
    };
    auto stage = [&](bool extra, uint32_t kv) {
      const unsigned long long m = __ballot(extra);
      if (m)
      {
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&s_ne, static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, leader);
        if (extra)
        {
          const uint32_t p = base + __popcll(m & ((1ull << lane) - 1ull));
          if (p < SLAB_EXTRA_CAP)
            s_extra[p] = kv;
          else
            extras[atomicAdd(&sa.counts[2 * FRAME + 1], 1u)] = kv;  // staging area full: straight to the list
        }
      }
    };
    constexpr int KB = 8;  // returning LDS atomics in flight per lane
#pragma unroll
    for (int j0 = 0; j0 < SLAB_KR; j0 += KB)
    {
      if (static_cast<uint32_t>(j0) * SLAB_THREADS < n_round)  // block-uniform; no break: kreg must stay in registers
      {
        bool ex[KB];
#pragma unroll
        for (int u = 0; u < KB; u++)
          ex[u] = mark(kreg[j0 + u]);
#pragma unroll
        for (int u = 0; u < KB; u++)
          stage(ex[u], kreg[j0 + u]);
      }
    }
    for (uint32_t i0 = SLAB_KR * SLAB_THREADS + tid; i0 < n_round; i0 += SLAB_THREADS)  // lists beyond the register file
    {
      const uint32_t kv = i0 < n_keys ? keys[i0] : 0xffffffffu;
      stage(mark(kv), kv);
    }
    __syncthreads();
    // the staged extras leave with one reservation per workgroup
    const uint32_t ne = min(s_ne, static_cast<uint32_t>(SLAB_EXTRA_CAP));
    if (tid == 0)
      s_gbase = ne ? atomicAdd(&sa.counts[2 * FRAME + 1], ne) : 0u;
    // the slab: coalesced 8-byte stores + popcounts per 256-word block (phase a of the rank scan)
    unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2) + w_first;
    const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(s_bits);
    for (uint32_t w = tid; w < SLAB_WORDS64; w += SLAB_THREADS)
    {
      const unsigned long long v = s64[w];
      if (w < n_w)
        bm[w] = v;
      uint32_t c = __popcll(v);
#pragma unroll
      for (int s = 32; s > 0; s >>= 1)
        c += __shfl_xor(c, s);
      if (lane == 0 && c)
        atomicAdd(&s_bsum[w / SCAN_WPB], c);
    }
    __syncthreads();
    for (uint32_t i = tid; i < ne; i += SLAB_THREADS)
      extras[s_gbase + i] = s_extra[i];
    const uint32_t blk0 = w_first / SCAN_WPB;
    if (tid < static_cast<int>(SLAB_WORDS64 / SCAN_WPB) && static_cast<uint32_t>(tid) * SCAN_WPB < n_w)
      blocksums[static_cast<size_t>(FRAME) * nblk_cap + blk0 + tid] = s_bsum[tid];
  }
}

// ---- large batches: slab voxelisation and emission in one kernel --------------------------------------------
// With one workgroup per frame walking the slabs in order, the running voxel count is known inside the workgroup: the
// voxel records can leave straight from the LDS slab, in key order, and k_scan_b / k_emit (a second trip of the bitmap
// through global memory, 250 us per 256 frames) disappear.  Per slab: the 256-word groups of the bitmap are dealt
// round-robin to the 16 waves (a lane owns 4 consecutive words: 64-word groups cost 2.6x the rounds on a sparse lattice);
// pass 1 counts the groups, one wave scans the 64 group totals, pass 2 writes bitmap and rank prefix and emits the group's
// voxels 64 at a time: slot s finds its owner lane by a binary search over the wave's inclusive counts (register
// shuffles), its word among the owner's four, its bit by popcount descent, and the lanes store consecutive ranks.
constexpr int SE_EXTRA_CAP = 7168;  // extras staged in LDS per slab (the ground slab of an OS1-128 scan holds ~5 k)
constexpr int SE_GWORDS = 256;                      // bitmap words per group: four consecutive words per lane
constexpr int SE_GROUPS = SLAB_WORDS64 / SE_GWORDS;  // 64 groups per slab

__device__ __forceinline__ int select_bit64(unsigned long long w, uint32_t u)
{
  // position of the u-th (0-based) set bit of w, by popcount descent
  int pos = 0;
#pragma unroll
  for (int width = 32; width >= 1; width >>= 1)
  {
    const unsigned long long lowmask = (width == 32) ? 0xffffffffull : ((1ull << width) - 1ull);
    const uint32_t c = __popcll((w >> pos) & lowmask);
    if (u >= c)
    {
      u -= c;
      pos += width;
    }
  }
  return pos;
}

__global__ __launch_bounds__(SLAB_THREADS) void k_slab_emit(const GridParams g, FrameHdr* hdrs, SlabArrays sa, uint32_t pt_cap, unsigned long long* __restrict__ bitmaps,
                                                           uint32_t* __restrict__ wprefix_all, VoxelArrays va_all, uint32_t init_count, unsigned long long* __restrict__ prof)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[SLAB_WORDS64 * 2];
  __shared__ uint32_t s_extra[SE_EXTRA_CAP];
  __shared__ uint32_t s_gsum[SE_GROUPS];  // per group: voxel count, then exclusive base
  __shared__ uint32_t s_ne, s_total, s_next;
  const uint32_t FRAME = blockIdx.x;
  FrameHdr& h = hdrs[FRAME];
  const uint32_t n_slabs = (h.n_words + SLAB_WORDS64 - 1) / SLAB_WORDS64;
  if (n_slabs == 0)
    return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n_keys = sa.counts[2 * FRAME];
  const uint32_t* keys = sa.keys + static_cast<size_t>(FRAME) * pt_cap;
  uint32_t* extras = sa.extras + static_cast<size_t>(FRAME) * pt_cap;
  unsigned long long* bm_frame = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const uint32_t n_round = (n_keys + 63u) & ~63u;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(bm_frame) | reinterpret_cast<uintptr_t>(wprefix)) & 15u) == 0u;  // 16-byte stores allowed for this frame
  const int dx = h.div_b[0], dxy = h.div_b[0] * h.div_b[1];
  const float inv_dx = 1.0f / static_cast<float>(dx);
  const uint32_t nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2;
  uint32_t kreg[SLAB_KR];
#pragma unroll
  for (int j = 0; j < SLAB_KR; j++)
  {
    const uint32_t i = j * SLAB_THREADS + tid;
    kreg[j] = i < n_keys ? keys[i] : 0xffffffffu;
  }
  uint32_t run_base = 0;
  unsigned long long t_acc[5] = {0, 0, 0, 0, 0}, t_prev = prof ? wall_clock64() : 0ull;  // VOFOD_LDS_PROF diagnostics
  const unsigned long long t_start = t_prev;
#define SE_STAMP(i)                           \
  if (prof)                                   \
  {                                           \
    const unsigned long long t_now = wall_clock64(); \
    t_acc[i] += t_now - t_prev;               \
    t_prev = t_now;                           \
  }
  SE_STAMP(0);
  for (uint32_t SLAB = 0; SLAB < n_slabs; SLAB++)
  {
    const uint32_t w_first = SLAB * SLAB_WORDS64;
    const uint32_t n_w = min(SLAB_WORDS64, h.n_words - w_first);
    __syncthreads();  // the previous slab has left the LDS
    for (uint32_t i = tid; i < SLAB_WORDS64 / 2; i += SLAB_THREADS)
      reinterpret_cast<uint4*>(s_bits)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0)
    {
      s_ne = 0u;
      s_next = 0u;
    }
    __syncthreads();
    SE_STAMP(1);
    const uint32_t cell0 = SLAB * SLAB_CELLS;
    auto mark = [&](uint32_t kv) -> bool {
      const uint32_t local = kv - cell0;
      if (kv == 0xffffffffu || local >= SLAB_CELLS)
        return false;
      const uint32_t bit = 1u << (local & 31u);
      return (atomicOr(&s_bits[local >> 5], bit) & bit) != 0u;
    };
    auto stage = [&](bool extra, uint32_t kv) {
      const unsigned long long m = __ballot(extra);
      if (m)
      {
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&s_ne, static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, leader);
        if (extra)
        {
          const uint32_t p = base + __popcll(m & ((1ull << lane) - 1ull));
          if (p < SE_EXTRA_CAP)
            s_extra[p] = kv;
          else
            extras[atomicAdd(&sa.counts[2 * FRAME + 1], 1u)] = kv;
        }
      }
    };
    constexpr int KB = 8;
#pragma unroll
    for (int j0 = 0; j0 < SLAB_KR; j0 += KB)
    {
      if (static_cast<uint32_t>(j0) * SLAB_THREADS < n_round)
      {
        bool ex[KB];
#pragma unroll
        for (int u = 0; u < KB; u++)
          ex[u] = mark(kreg[j0 + u]);
#pragma unroll
        for (int u = 0; u < KB; u++)
          stage(ex[u], kreg[j0 + u]);
      }
    }
    for (uint32_t i0 = SLAB_KR * SLAB_THREADS + tid; i0 < n_round; i0 += SLAB_THREADS * KB)  // lists beyond the register file
    {
      uint32_t kv[KB];
      bool ex[KB];
#pragma unroll
      for (int u = 0; u < KB; u++)
        kv[u] = i0 + u * SLAB_THREADS < n_keys ? keys[i0 + u * SLAB_THREADS] : 0xffffffffu;
#pragma unroll
      for (int u = 0; u < KB; u++)
        ex[u] = mark(kv[u]);
#pragma unroll
      for (int u = 0; u < KB; u++)
        if (i0 + u * SLAB_THREADS - tid < n_round)  // wave-uniform
          stage(ex[u], kv[u]);
    }
    __syncthreads();
    SE_STAMP(2);
    const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(s_bits);
    // pass 1: voxel count of every group of 256 words (group gi belongs to wave gi % 16, a lane owns 4 consecutive words)
    for (int gi = wave; gi < SE_GROUPS; gi += SLAB_THREADS / 64)
    {
      const uint4* w4 = reinterpret_cast<const uint4*>(s64 + gi * SE_GWORDS + lane * 4);
      const uint4 a = w4[0], b = w4[1];
      uint32_t c = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1)
        c += __shfl_xor(c, sft);
      if (lane == 0)
        s_gsum[gi] = c;
    }
    __syncthreads();
    if (wave == 0)
    {
      // exclusive scan of the 64 group totals
      const uint32_t tot = s_gsum[lane];
      const uint32_t incl = wave_incl_scan(tot);
      s_gsum[lane] = incl - tot;
      if (lane == 63)
        s_total = incl;
    }
    const uint32_t ne = min(s_ne, static_cast<uint32_t>(SE_EXTRA_CAP));
    __syncthreads();
    const uint32_t slab_total = s_total;
    if (run_base + slab_total > g.vox_cap)
    {
      if (tid == 0)
      {
        h.status = VOFOD_ERR_CAPACITY;  // as k_scan_b
        h.V = 0;
      }
      return;
    }
    SE_STAMP(3);
    // pass 2: bitmap + rank prefix out, voxel records out.  The waves draw the groups from a counter: a group of the ground
    // sheet holds a hundred times the voxels of one in the air, a fixed deal leaves most waves waiting at the barrier.
    for (;;)
    {
      int gi = 0;
      if (lane == 0)
        gi = static_cast<int>(atomicAdd(&s_next, 1u));
      gi = __shfl(gi, 0);
      if (gi >= SE_GROUPS)
        break;
      const uint32_t w0 = gi * SE_GWORDS + lane * 4;  // first of this lane's four words
      unsigned long long word[4];
      uint32_t cw[4], c = 0;
#pragma unroll
      for (int q = 0; q < 4; q++)
      {
        word[q] = s64[w0 + q];
        cw[q] = __popcll(word[q]);
        c += cw[q];
      }
      const uint32_t incl = wave_incl_scan(c);
      const uint32_t T = __shfl(incl, 63);
      const uint32_t gbase = run_base + s_gsum[gi];
      {
        const uint32_t run = gbase + incl - c;
        if (w0 + 3 < n_w && vec_ok)  // 32 + 16 bytes per lane, contiguous across the wave
        {
          uint4* bo = reinterpret_cast<uint4*>(bm_frame + w_first + w0);
          bo[0] = make_uint4(static_cast<uint32_t>(word[0]), static_cast<uint32_t>(word[0] >> 32), static_cast<uint32_t>(word[1]), static_cast<uint32_t>(word[1] >> 32));
          bo[1] = make_uint4(static_cast<uint32_t>(word[2]), static_cast<uint32_t>(word[2] >> 32), static_cast<uint32_t>(word[3]), static_cast<uint32_t>(word[3] >> 32));
          *reinterpret_cast<uint4*>(wprefix + w_first + w0) = make_uint4(run, run + cw[0], run + cw[0] + cw[1], run + cw[0] + cw[1] + cw[2]);
        }
        else  // the lattice's last words
        {
          uint32_t r = run;
          for (int q = 0; q < 4; q++)
            if (w0 + q < n_w)
            {
              bm_frame[w_first + w0 + q] = word[q];
              wprefix[w_first + w0 + q] = r;
              r += cw[q];
            }
        }
      }
      if (T == 0)
        continue;
      // lattice coordinates of the group's first cell (wave-uniform), the voxels' follow from their offset in the group
      const uint32_t cell_first = (w_first + gi * SE_GWORDS) * 64u;
      const uint32_t gk2 = cell_first / dxy;
      const uint32_t grem = cell_first - gk2 * dxy;
      const uint32_t gk1 = grem / dx;
      const uint32_t gk0 = grem - gk1 * dx;
      const uint32_t cpack = cw[0] | (cw[1] << 8) | (cw[2] << 16);  // counts of the lane's first three words (each <= 64)
      for (uint32_t s0 = 0; s0 < T; s0 += 64)
      {
        const uint32_t sidx = s0 + lane;
        // owner lane: the first lane whose inclusive count exceeds the slot index
        int lo = 0, hi = 63;
#pragma unroll
        for (int step = 0; step < 6; step++)
        {
          const int mid = (lo + hi) >> 1;
          const uint32_t v = __shfl(incl, mid);
          if (v <= sidx)
            lo = mid + 1;
          else
            hi = mid;
        }
        const int owner = lo;
        const uint32_t oincl = __shfl(incl, owner), oc = __shfl(c, owner), opack = __shfl(cpack, owner);
        if (sidx < T)
        {
          uint32_t u = sidx - (oincl - oc);  // index among the owner's voxels
          int q = 0;
#pragma unroll
          for (int t = 0; t < 3; t++)
          {
            const uint32_t ct = (opack >> (8 * t)) & 0xffu;
            if (q == t && u >= ct)
            {
              u -= ct;
              q = t + 1;
            }
          }
          const uint32_t wi = gi * SE_GWORDS + owner * 4 + q;
          const int bit = select_bit64(s64[wi], u);
          const uint32_t off = (static_cast<uint32_t>(owner) * 4u + q) * 64u + bit;  // < 16384
          const uint32_t key = cell_first + off;
          const uint32_t rank = gbase + sidx;
          // (gk0 + off) / dx by float reciprocal: the operands are below 2^14 + dx, the correction steps make it exact
          const uint32_t x = gk0 + off;
          uint32_t qd = static_cast<uint32_t>(static_cast<float>(x) * inv_dx);
          qd -= (qd * static_cast<uint32_t>(dx) > x) ? 1u : 0u;
          qd += ((qd + 1u) * static_cast<uint32_t>(dx) <= x) ? 1u : 0u;
          const int k0 = static_cast<int>(x - qd * dx);
          uint32_t y = gk1 + qd;
          int k2 = static_cast<int>(gk2);
          while (y >= static_cast<uint32_t>(h.div_b[1]))  // a group spans a few dozen rows: rarely more than one plane boundary
          {
            y -= h.div_b[1];
            k2++;
          }
          const int k1 = static_cast<int>(y);
          float4 p;
          p.x = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k0), 0.5f), g.leaf[0]), h.offset[0]);
          p.y = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k1), 0.5f), g.leaf[1]), h.offset[1]);
          p.z = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k2), 0.5f), g.leaf[2]), h.offset[2]);
          p.w = __uint_as_float(init_count & 0x7fffffffu);
          va.pts[rank] = p;
          va.key[rank] = key;
          va.bb[rank] = (static_cast<uint32_t>(((k2 >> 2) * nby + (k1 >> 2)) * nbx + (k0 >> 2)) << 6) | static_cast<uint32_t>((k0 & 3) | ((k1 & 3) << 2) | ((k2 & 3) << 4));
          if (!(init_count & 0x80000000u))
          {
            va.parent[rank] = rank;
            va.csize[rank] = 0;
            va.cclose[rank] = 0;
            int* cb = &va.cbox[6 * rank];
            cb[0] = cb[1] = cb[2] = 0x7fffffff;
            cb[3] = cb[4] = cb[5] = static_cast<int>(0x80000000u);
          }
        }
      }
    }
    // the staged extras add to their voxels' weights right here: the records were just written by this workgroup (they sit
    // in L2), the rank comes from the prefix entry written above and the slab's word in LDS.  Only extras beyond the staging
    // area went to the global list (k_count_extras).
    __syncthreads();
    for (uint32_t i = tid; i < ne; i += SLAB_THREADS)
    {
      const uint32_t local = s_extra[i] - cell0;
      const uint32_t w = local >> 6;
      const uint32_t rank = wprefix[w_first + w] + __popcll(s64[w] & ((1ull << (local & 63u)) - 1ull));
      atomicAdd(reinterpret_cast<uint32_t*>(&va.pts[rank].w), 1u);
    }
    run_base += slab_total;
    SE_STAMP(4);
  }
  if (tid == 0)
    h.V = run_base;
  if (prof && tid == 0)
  {
    for (int i = 0; i < 5; i++)
      prof[static_cast<size_t>(FRAME) * 16 + i] = t_acc[i];
    prof[static_cast<size_t>(FRAME) * 16 + 5] = t_start;
    prof[static_cast<size_t>(FRAME) * 16 + 6] = wall_clock64();
    prof[static_cast<size_t>(FRAME) * 16 + 7] = n_keys;
  }
#undef SE_STAMP
}

// weights: every voxel starts at 1 (k_emit); each extra point adds 1 to its voxel
__global__ __launch_bounds__(256) void k_count_extras(const GridParams g, const FrameHdr* hdrs, SlabArrays sa, uint32_t pt_cap, const unsigned long long* __restrict__ bitmaps,
                                                      const uint32_t* __restrict__ wprefix_all, VoxelArrays va_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  const FrameHdr& h = hdrs[FRAME];
  if (h.V == 0)
    return;
  const uint32_t ne = sa.counts[2 * FRAME + 1];
  const uint32_t* extras = sa.extras + static_cast<size_t>(FRAME) * pt_cap;
  const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  for (uint32_t i = BX * blockDim.x + threadIdx.x; i < ne; i += GX * blockDim.x)
  {
    const uint32_t r = rank_of(bm, wprefix, extras[i]);
    atomicAdd(reinterpret_cast<uint32_t*>(&va.pts[r].w), 1u);
  }
}

}  // namespace vk
