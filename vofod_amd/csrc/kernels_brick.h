// Brick-level Euclidean clustering for gfx950 (fast path of K7).
//
// When the tolerance is large against the voxel pitch (every pair of voxels inside a 4x4x4 brick is
// closer than the tolerance: tol/leaf > 3*sqrt(3)), a brick is a clique of the Euclidean graph and can
// be merged as ONE node.  A brick's occupancy is exactly one 64-bit word (bit = x + 4y + 16z inside the
// brick), so the neighbourhood test between two bricks is a handful of AND operations against host-built
// masks: sure[o][p] = the bits q of the neighbour brick at brick offset o that are certainly within the
// tolerance of bit p, amb[o][p] = the bits sitting on the tolerance boundary, decided by the float
// expression FLANN evaluates on the actual voxel centres.  The union-find then runs over the
// few thousand occupied bricks of a frame instead of its tens of thousands of voxels, and a voxel's label
// is the smallest voxel rank of its brick component (the same canonical label as the voxel-level kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_cluster.h"

namespace vk
{

#ifndef VOFOD_BRICK_UFM
#define VOFOD_BRICK_UFM 3
#endif
constexpr int UFM = VOFOD_BRICK_UFM;  // union-find memory mode of the brick kernels (see uf_ld)

struct BrickOff
{
  int8_t dx, dy, dz;
  uint8_t has_amb;
  uint32_t pad;
  unsigned long long ua, ub;  // bits of this brick / of the neighbour brick that have any partner at this offset
};

// Is any voxel of brick (bx,by,bz) with occupancy A within the tolerance of any voxel of the brick at stencil offset o
// with occupancy B?  sure[o][p] / amb[o][p]: see the header comment; ambiguous pairs are decided by FLANN's float
// expression on the actual centres.
__device__ __forceinline__ bool brick_pair_conn(const GridParams& g, const BrickParams& bp, const FrameHdr& h, const BrickOff& off, int o,
                                                const unsigned long long* __restrict__ sure, const unsigned long long* __restrict__ amb, unsigned long long A,
                                                unsigned long long B, int bx, int by, int bz)
{
  const int nx = bx + off.dx, ny = by + off.dy, nz = bz + off.dz;
  bool conn = false;
  const unsigned long long* s = sure + static_cast<size_t>(o) * 64;
  unsigned long long a = A & off.ua;
  while (a && !conn)
  {
    const int p = __ffsll(static_cast<long long>(a)) - 1;
    a &= a - 1;
    conn = (s[p] & B) != 0ull;
  }
  if (!conn && off.has_amb)
  {
    const unsigned long long* m = amb + static_cast<size_t>(o) * 64;
    a = A & off.ua;
    while (a && !conn)
    {
      const int p = __ffsll(static_cast<long long>(a)) - 1;
      a &= a - 1;
      unsigned long long cand = m[p] & B;
      if (!cand)
        continue;
      const float px = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bx + (p & 3)), 0.5f), g.leaf[0]), h.offset[0]);
      const float py = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * by + ((p >> 2) & 3)), 0.5f), g.leaf[1]), h.offset[1]);
      const float pz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bz + (p >> 4)), 0.5f), g.leaf[2]), h.offset[2]);
      while (cand && !conn)
      {
        const int q = __ffsll(static_cast<long long>(cand)) - 1;
        cand &= cand - 1;
        const float qx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * nx + (q & 3)), 0.5f), g.leaf[0]), h.offset[0]);
        const float qy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * ny + ((q >> 2) & 3)), 0.5f), g.leaf[1]), h.offset[1]);
        const float qz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * nz + (q >> 4)), 0.5f), g.leaf[2]), h.offset[2]);
        const float ddx = __fsub_rn(px, qx), ddy = __fsub_rn(py, qy), ddz = __fsub_rn(pz, qz);
        float d2 = __fmul_rn(ddx, ddx);
        d2 = __fadd_rn(d2, __fmul_rn(ddy, ddy));
        d2 = __fadd_rn(d2, __fmul_rn(ddz, ddz));
        conn = d2 < bp.r2;
      }
    }
  }
  return conn;
}

// mark every voxel in its brick word; the first voxel of a brick registers it
__global__ __launch_bounds__(256) void k_brick_set(const GridParams g, const BrickParams bp, FrameHdr* hdrs, VoxelArrays va_all, BrickArrays ba_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  FrameHdr& h = hdrs[FRAME];
  const uint32_t v = BX * blockDim.x + threadIdx.x;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  uint32_t b = 0;
  bool first = false;
  if (v < h.V)
  {
    int i, j, k;
    key_to_ijk(h, va.key[v], i, j, k);
    first = brick_mark(h, ba, i, j, k, v, b);
  }
  brick_append(h, ba, first, b);
}

template <int BRICK_LANES>
__global__ __launch_bounds__(256) void k_brick_union(const GridParams g, const BrickParams bp, const BrickOff* __restrict__ offs,
                                                     const unsigned long long* __restrict__ sure, const unsigned long long* __restrict__ amb, const FrameHdr* hdrs,
                                                     BrickArrays ba_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  // BRICK_LANES lanes share one brick and split its neighbour offsets: short dependent chains, 8x the waves
  const uint32_t t = (BX * blockDim.x + threadIdx.x) / BRICK_LANES;
  const int sub = threadIdx.x % BRICK_LANES;
  if (t >= h.n_bricks)
    return;
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  const int nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2, nbz = (h.div_b[2] + 3) >> 2;
  const uint32_t b = ba.blist[t];
  const unsigned long long A = ba.bricks[b];
  const int bz = b / (nbx * nby);
  const int brem = b - bz * nbx * nby;
  const int by = brem / nbx;
  const int bx = brem - by * nbx;
  uint32_t rv = b;  // current representative of this brick's component (refreshed lazily)
  constexpr int CH = 4;  // neighbour words fetched per round: independent loads in flight
  for (int o0 = sub * CH; o0 < bp.n_off; o0 += CH * BRICK_LANES)
  {
    unsigned long long Bw[CH];
    uint32_t nbi[CH];
#pragma unroll
    for (int c = 0; c < CH; c++)
    {
      Bw[c] = 0ull;
      nbi[c] = 0;
      const int o = o0 + c;
      if (o < bp.n_off)
      {
        const BrickOff off = offs[o];
        const int nx = bx + off.dx, ny = by + off.dy, nz = bz + off.dz;
        if (nx >= 0 && nx < nbx && ny >= 0 && ny < nby && nz < nbz)
        {
          nbi[c] = static_cast<uint32_t>((nz * nby + ny) * nbx + nx);
          Bw[c] = ba.bricks[nbi[c]];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; c++)
    {
      const unsigned long long B = Bw[c];
      if (!B)
        continue;
      const int o = o0 + c;
      const BrickOff off = offs[o];
      if (!(A & off.ua) || !(B & off.ub))
        continue;  // no bit of either brick can reach the other at this offset
      const uint32_t nb = nbi[c];
      const int nx = bx + off.dx, ny = by + off.dy, nz = bz + off.dz;
      bool conn = false;
      const unsigned long long* s = sure + static_cast<size_t>(o) * 64;
      unsigned long long a = A;
      while (a && !conn)
      {
        const int p = __ffsll(static_cast<long long>(a)) - 1;
        a &= a - 1;
        conn = (s[p] & B) != 0ull;
      }
      if (!conn && off.has_amb)
      {
        const unsigned long long* m = amb + static_cast<size_t>(o) * 64;
        a = A;
        while (a && !conn)
        {
          const int p = __ffsll(static_cast<long long>(a)) - 1;
          a &= a - 1;
          unsigned long long cand = m[p] & B;
          if (!cand)
            continue;
          const float px = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bx + (p & 3)), 0.5f), g.leaf[0]), h.offset[0]);
          const float py = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * by + ((p >> 2) & 3)), 0.5f), g.leaf[1]), h.offset[1]);
          const float pz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bz + (p >> 4)), 0.5f), g.leaf[2]), h.offset[2]);
          while (cand && !conn)
          {
            const int q = __ffsll(static_cast<long long>(cand)) - 1;
            cand &= cand - 1;
            const float qx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * nx + (q & 3)), 0.5f), g.leaf[0]), h.offset[0]);
            const float qy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * ny + ((q >> 2) & 3)), 0.5f), g.leaf[1]), h.offset[1]);
            const float qz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * nz + (q >> 4)), 0.5f), g.leaf[2]), h.offset[2]);
            const float ddx = __fsub_rn(px, qx), ddy = __fsub_rn(py, qy), ddz = __fsub_rn(pz, qz);
            float d2 = __fmul_rn(ddx, ddx);
            d2 = __fadd_rn(d2, __fmul_rn(ddy, ddy));
            d2 = __fadd_rn(d2, __fmul_rn(ddz, ddz));
            conn = d2 < bp.r2;
          }
        }
      }
      if (conn)
      {
        // union(b, nb) with the representative of b kept in a register.  Cheap exit first: once the forest has
        // settled most neighbours hang directly under our representative (one load instead of two chases).
        const uint32_t pn = uf_ld<UFM>(&ba.bparent[nb]);
        if (pn == rv)
          continue;
        uint32_t ra = uf_find<UFM>(ba.bparent, rv), rb = (pn == nb) ? nb : uf_find<UFM>(ba.bparent, pn);
        while (ra != rb)
        {
          if (ra < rb)
          {
            const uint32_t tmp = ra;
            ra = rb;
            rb = tmp;
          }
          const uint32_t old = atomicCAS(&ba.bparent[ra], ra, rb);
          if (old == ra)
            break;
          ra = old;
        }
        rv = min(ra, rb);
      }
    }
  }
}

// ---- two-phase variant for stencils of at most 64 brick offsets -------------------------------------------
// Phase 1 (k_brick_conn): CONN_LANES lanes share a brick, split its neighbour offsets and OR their findings into one
// 64-bit connectivity mask per brick (bit o: the occupied neighbour at offset o is within the tolerance).  No atomics,
// no dependent chains: pure probing, spread over 8x the lanes.
constexpr int CONN_LANES = 8;
__global__ __launch_bounds__(256) void k_brick_conn(const GridParams g, const BrickParams bp, const BrickOff* __restrict__ offs,
                                                    const unsigned long long* __restrict__ sure, const unsigned long long* __restrict__ amb, const FrameHdr* hdrs,
                                                    BrickArrays ba_all, unsigned long long* __restrict__ conn_all, unsigned long long* __restrict__ bconn_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t t = (BX * blockDim.x + threadIdx.x) / CONN_LANES;
  const int sub = threadIdx.x % CONN_LANES;
  // whole groups of CONN_LANES lanes share t, so the shuffles below see uniform participation per group
  const bool live = t < h.n_bricks;
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  const int nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2, nbz = (h.div_b[2] + 3) >> 2;
  unsigned long long mask = 0ull;
  if (live)
  {
    const uint32_t b = ba.blist[t];
    const unsigned long long A = ba.bricks[b];
    const int bz = b / (nbx * nby);
    const int brem = b - bz * nbx * nby;
    const int by = brem / nbx;
    const int bx = brem - by * nbx;
    for (int o = sub; o < bp.n_off; o += CONN_LANES)
    {
      const BrickOff off = offs[o];
      const int nx = bx + off.dx, ny = by + off.dy, nz = bz + off.dz;
      if (nx < 0 || nx >= nbx || ny < 0 || ny >= nby || nz >= nbz)
        continue;
      if (!(A & off.ua))
        continue;
      const unsigned long long B = ba.bricks[static_cast<uint32_t>((nz * nby + ny) * nbx + nx)];
      if (!(B & off.ub))
        continue;
      const bool conn = brick_pair_conn(g, bp, h, off, o, sure, amb, A, B, bx, by, bz);
      if (conn)
        mask |= 1ull << o;
    }
  }
#pragma unroll
  for (int s2 = 1; s2 < CONN_LANES; s2 <<= 1)
    mask |= __shfl_xor(mask, s2);
  if (live && sub == 0)
  {
    conn_all[static_cast<size_t>(FRAME) * g.vox_cap + t] = mask;
    if (bconn_all)
      bconn_all[static_cast<size_t>(FRAME) * bp.bricks_cap + ba.blist[t]] = mask;
  }
}

// Phase 2 alternative (k_brick_link_tr): transitive reduction of the brick graph before any union.  The edge (b,n)
// is redundant when some brick c with b < c < n is connected to both: (b,c) is among b's own edges and (c,n) is read
// from c's connectivity mask through pair_idx[o1][o2] = the stencil index of offset(o2) - offset(o1) (or -1).  By
// induction over n - b every dropped edge is implied by kept ones, so the components are unchanged while the number
// of union operations (dependent pointer chases + compare-and-swaps, the measured cost of this stage) falls from
// all neighbours to a few per brick.  Restricting the intermediates to adjacent bricks was measured 2.7x slower.
__global__ __launch_bounds__(256) void k_brick_link_tr(const GridParams g, const BrickParams bp, const BrickOff* __restrict__ offs, const int8_t* __restrict__ pair_idx,
                                                       const FrameHdr* hdrs, BrickArrays ba_all, const unsigned long long* __restrict__ conn_all,
                                                       const unsigned long long* __restrict__ bconn_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t t = BX * blockDim.x + threadIdx.x;
  if (t >= h.n_bricks)
    return;
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  const unsigned long long* bconn = bconn_all + static_cast<size_t>(FRAME) * bp.bricks_cap;
  const unsigned long long mask = conn_all[static_cast<size_t>(FRAME) * g.vox_cap + t];
  if (!mask)
    return;
  const int nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2;
  const uint32_t b = ba.blist[t];
  // neighbour masks of up to TR_K neighbours, fetched with independent loads
  constexpr int TR_K = 16;
  unsigned long long cm[TR_K];
  int oi[TR_K];
  int k = 0;
  {
    unsigned long long m = mask;
#pragma unroll
    for (int i = 0; i < TR_K; i++)
    {
      cm[i] = 0ull;
      oi[i] = -1;
      if (m)
      {
        const int o = __ffsll(static_cast<long long>(m)) - 1;
        m &= m - 1;
        const BrickOff off = offs[o];
        oi[i] = o;
        cm[i] = bconn[b + static_cast<uint32_t>((off.dz * nby + off.dy) * nbx + off.dx)];
        k = i + 1;
      }
    }
  }
  unsigned long long keep = mask;
#pragma unroll
  for (int i2 = 0; i2 < TR_K; i2++)  // candidate edge to drop: o2 = oi[i2]
  {
    if (i2 >= k)
      break;
    bool drop = false;
#pragma unroll
    for (int i1 = 0; i1 < TR_K; i1++)  // intermediate c = b + offset(oi[i1])
    {
      if (i1 >= k || i1 == i2)
        continue;
      const int o3 = pair_idx[oi[i1] * 64 + oi[i2]];
      if (o3 >= 0 && ((cm[i1] >> o3) & 1ull))
        drop = true;
    }
    if (drop)
      keep &= ~(1ull << oi[i2]);
  }
  uint32_t rv = b;
  while (keep)
  {
    const int o = __ffsll(static_cast<long long>(keep)) - 1;
    keep &= keep - 1;
    const BrickOff off = offs[o];
    const uint32_t nb = b + static_cast<uint32_t>((off.dz * nby + off.dy) * nbx + off.dx);
    const uint32_t pn = uf_ld<UFM>(&ba.bparent[nb]);
    if (pn == rv)
      continue;
    uint32_t ra = uf_find<UFM>(ba.bparent, rv), rb = (pn == nb) ? nb : uf_find<UFM>(ba.bparent, pn);
    while (ra != rb)
    {
      if (ra < rb)
      {
        const uint32_t tmp = ra;
        ra = rb;
        rb = tmp;
      }
      const uint32_t old = atomicCAS(&ba.bparent[ra], ra, rb);
      if (old == ra)
        break;
      ra = old;
    }
    rv = min(ra, rb);
  }
}

// per occupied brick: point it straight at its representative and fold its smallest voxel rank into the component's.
// A brick's smallest rank belongs to its lowest set bit (bit order inside a brick is the key order), and that voxel's
// rank comes from the occupancy bitmap's prefix array: no per-voxel atomics.  Consecutive list entries mostly share
// the component, so the atomicMin on the representative is issued once per run of equal representatives in the wave.
__global__ __launch_bounds__(256) void k_brick_root(const GridParams g, const BrickParams bp, const FrameHdr* hdrs, BrickArrays ba_all,
                                                    const unsigned long long* __restrict__ bitmaps, const uint32_t* __restrict__ wprefix_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t t = BX * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  uint32_t R = 0xffffffffu, vmin = 0xffffffffu;
  if (t < h.n_bricks)
  {
    const uint32_t b = ba.blist[t];
    R = b;
    uint32_t p;
    while ((p = uf_ld<UFM>(&ba.bparent[R])) != R)
      R = p;
    if (R != b)
      ba.bparent[b] = R;  // still an ancestor for every concurrent chase through b
    const int nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2;
    const int bz = b / (nbx * nby);
    const int brem = b - bz * nbx * nby;
    const int by = brem / nbx;
    const int bx = brem - by * nbx;
    const int bit = __ffsll(static_cast<long long>(ba.bricks[b])) - 1;
    const uint32_t key = static_cast<uint32_t>(((4 * bz + (bit >> 4)) * h.div_b[1] + (4 * by + ((bit >> 2) & 3))) * h.div_b[0] + 4 * bx + (bit & 3));
    const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
    const uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
    vmin = rank_of(bm, wprefix, key);
  }
  int end;
  const bool head = run_heads(R, lane, end);
#pragma unroll
  for (int s2 = 1; s2 < 64; s2 <<= 1)
  {
    const uint32_t o = __shfl_down(vmin, s2);
    if (lane + s2 < end)
      vmin = min(vmin, o);
  }
  if (head && R != 0xffffffffu)
    atomicMin(&ba.bcmin[R], vmin);
}

__global__ __launch_bounds__(256) void k_brick_clear(const GridParams g, const BrickParams bp, const FrameHdr* hdrs, BrickArrays ba_all)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const FrameHdr& h = hdrs[FRAME];
  const uint32_t t = BX * blockDim.x + threadIdx.x;
  if (t >= h.n_bricks)
    return;
  const BrickArrays ba = frame_bricks(ba_all, FRAME, bp.bricks_cap, g.vox_cap);
  const uint32_t b = ba.blist[t];
  ba.bricks[b] = 0ull;
  ba.bmin[b] = 0xffffffffu;
  ba.bcmin[b] = 0xffffffffu;
}

}  // namespace vk
