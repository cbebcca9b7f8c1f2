// Brick-level Euclidean clustering of one frame inside one workgroup's LDS (gfx950: 160 KB per CU).
//
// The global-memory brick kernels (kernels_brick.h) spend their time on dependent L2/HBM round trips: probing the
// dense brick lattice (one 64-byte sector per probe), chasing union-find parents, compare-and-swap hooks.  A frame of
// a 128-ring LiDAR at 0.25 m holds ~34 k voxels in ~5 k occupied bricks: that graph fits a CU's LDS.  One 1024-thread
// workgroup per frame
//   A  marks the occupied bricks in an LDS bitmap of the frame's brick lattice (bit = linear brick id),
//   B  prefix-sums the bitmap's word popcounts: a brick's node index is its rank among the occupied bricks, so the
//      "is the neighbour occupied, and which node is it" lookup of the clustering is one or two LDS reads,
//   C  ORs every voxel into its brick's 64-bit occupancy word,
//   D  walks the half stencil of brick offsets row by row (all x-offsets of a (dy,dz) row come out of one window of the
//      brick bitmap), decides each occupied pair first on 2x2x2 octants (host-built 8x8 bit matrices: "every voxel pair
//      of these two octants is within the tolerance" / "no pair can be"), only then with the per-voxel sure/ambiguous
//      masks and FLANN float expression of k_brick_conn, and merges connected bricks in an LDS union-find,
//   E  labels every voxel with the smallest voxel rank of its component (the oracle's canonical label).
// Frames are independent, so a batch keeps as many CUs busy as it has frames and the kernel's duration does not grow
// with the batch until every CU holds a frame.  A frame with more than LB_MAX occupied bricks, or a brick lattice
// beyond the LDS bitmap, raises CCL_RETRY_STATUS in its header: the host re-runs the batch with the global kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_brick.h"

namespace vk
{

constexpr int LB_MAX = 7168;          // occupied bricks per frame
constexpr int LB_BITWORDS = 9984;     // 32-bit words of the brick-lattice bitmap: 319 488 bricks = 20 M cells
#ifndef VOFOD_LB_THREADS
#define VOFOD_LB_THREADS 1024  /* 512 (an experiment: half the waves per CU) makes the kernel this much slower: see DESIGN 5.3 */
#endif
constexpr int LB_THREADS = VOFOD_LB_THREADS;
constexpr int LB_LANES = 8;           // lanes sharing one brick in phase D
constexpr int LB_MAX_ROWS = 16;       // (dy,dz) rows of the half stencil (13 for a reach of 2 bricks)
constexpr int LB_MAX_OFF = 64;
constexpr int LB_WIN = 7;             // widest x-window: reach of 3 bricks
#ifndef VOFOD_LB_ST_ROWS
#define VOFOD_LB_ST_ROWS 384
#endif
constexpr int LB_ST_ROWS = VOFOD_LB_ST_ROWS;  // components per frame whose statistics are gathered in LDS (36 B each); the rest use global atomics
constexpr int32_t CCL_RETRY_STATUS = 1000;  // internal FrameHdr::status, never returned through the C-ABI

struct LbRow
{
  int8_t dy, dz;
  uint8_t valid;      // bit s: the offset dx = s - R exists in the half stencil
  int8_t o[LB_WIN];   // stencil index of dx = s - R
  uint8_t pad[6];
};

constexpr int LB_RV = 7;  // largest voxel offset per axis the ball tables cover

struct LbTables
{
  int32_t R;       // reach in bricks along x
  int32_t n_rows;
  unsigned long long near_mask;  // bit o: stencil offset o reaches the adjacent brick only (max(|dx|,|dy|,|dz|) <= 1)
  unsigned long long axis_mask;  // bit o: face neighbour (|dx| + |dy| + |dz| == 1)
  LbRow rows[LB_MAX_ROWS];
  unsigned long long oct[2 * LB_MAX_OFF];  // per offset: sure8, maybe8 (bit po*8+qo; octant = (x>>1) | (y>>1)<<1 | (z>>1)<<2)
  // The Euclidean predicate on relative voxel offsets (the EdgeClassifier of the host): [dz + RV][dy + RV] bit (dx + RV)
  uint16_t ball_sure[2 * LB_RV + 1][2 * LB_RV + 2];  // certainly within the tolerance
  uint16_t ball_amb[2 * LB_RV + 1][2 * LB_RV + 2];   // on the boundary: FLANN's float expression decides
};

// octant occupancy of a brick word (bit p = x + 4y + 16z)
__device__ __forceinline__ uint32_t lb_oct8(unsigned long long W)
{
  unsigned long long t = W | (W >> 1);
  t |= t >> 4;
  t |= t >> 16;
  const uint32_t lo = static_cast<uint32_t>(t), hi = static_cast<uint32_t>(t >> 32);
  return (lo & 1u) | ((lo >> 1) & 2u) | ((lo >> 6) & 4u) | ((lo >> 7) & 8u) | ((hi & 1u) << 4) | (((hi >> 2) & 1u) << 5) | (((hi >> 8) & 1u) << 6) | (((hi >> 10) & 1u) << 7);
}

// any bit (po, qo) of the 8x8 matrix m with po in A8 and qo in B8?  Branch-free on the two 32-bit halves: the rows of A8
// become all-ones bytes, B8 is replicated into every byte.
__device__ __forceinline__ uint32_t lb_rowsel(uint32_t A8x01010101, uint32_t pick)
{
  const uint32_t x = A8x01010101 & pick;               // byte r holds bit (r or r + 4) of A8, or nothing
  const uint32_t t = ((x + 0x7f7f7f7fu) | x) & 0x80808080u;  // high bit of every non-zero byte
  return (t >> 7) * 0xffu;
}
__device__ __forceinline__ bool lb_octtest(unsigned long long m, uint32_t A8, uint32_t B8)
{
  const uint32_t a = A8 * 0x01010101u, cols = B8 * 0x01010101u;
  const uint32_t lo = static_cast<uint32_t>(m) & lb_rowsel(a, 0x08040201u) & cols;
  const uint32_t hi = static_cast<uint32_t>(m >> 32) & lb_rowsel(a, 0x80402010u) & cols;
  return (lo | hi) != 0u;
}

__device__ __forceinline__ uint32_t lb_shift(uint32_t m, int sh) { return sh >= 0 ? (sh < 32 ? m >> sh : 0u) : m << (-sh); }

// Exact test of two occupied bricks, LDS and registers only: x-rows of A against x-rows of B through the ball tables.
// (ddx,ddy,ddz): brick offset of B from A = (bx,by,bz).
__device__ __forceinline__ bool lb_pair_conn(const LbTables& tab, const GridParams& g, const BrickParams& bp, const FrameHdr& h, unsigned long long A, unsigned long long B, int bx,
                                             int by, int bz, int ddx, int ddy, int ddz)
{
  unsigned long long a = A;
  while (a)
  {
    const int ra = (__ffsll(static_cast<long long>(a)) - 1) >> 2;  // row = py + 4 pz
    const uint32_t a4 = static_cast<uint32_t>(A >> (4 * ra)) & 0xfu;
    a &= ~(0xfull << (4 * ra));
    const int py = ra & 3, pz = ra >> 2;
    unsigned long long b = B;
    while (b)
    {
      const int rb = (__ffsll(static_cast<long long>(b)) - 1) >> 2;
      const uint32_t b4 = static_cast<uint32_t>(B >> (4 * rb)) & 0xfu;
      b &= ~(0xfull << (4 * rb));
      const int qy = rb & 3, qz = rb >> 2;
      const int dyr = 4 * ddy + qy - py, dzr = 4 * ddz + qz - pz;
      if (dyr < -LB_RV || dyr > LB_RV || dzr < -LB_RV || dzr > LB_RV)
        continue;
      const uint32_t Ms = tab.ball_sure[dzr + LB_RV][dyr + LB_RV], Ma = tab.ball_amb[dzr + LB_RV][dyr + LB_RV];
      if (!(Ms | Ma))
        continue;
      // bit qx of T: some px of the row has dx = 4 ddx + qx - px inside the mask
      uint32_t Ts = 0, Ta = 0;
#pragma unroll
      for (int px = 0; px < 4; px++)
        if ((a4 >> px) & 1u)
        {
          const int sh = 4 * ddx - px + LB_RV;
          Ts |= lb_shift(Ms, sh);
          Ta |= lb_shift(Ma, sh);
        }
      if (Ts & b4)
        return true;
      if (Ta & b4)
      {
        const float pyc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * by + py), 0.5f), g.leaf[1]), h.offset[1]);
        const float pzc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bz + pz), 0.5f), g.leaf[2]), h.offset[2]);
        const float qyc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * (by + ddy) + qy), 0.5f), g.leaf[1]), h.offset[1]);
        const float qzc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * (bz + ddz) + qz), 0.5f), g.leaf[2]), h.offset[2]);
        for (int px = 0; px < 4; px++)
          for (int qx = 0; qx < 4; qx++)
          {
            if (!((a4 >> px) & 1u) || !((b4 >> qx) & 1u))
              continue;
            const int bit = 4 * ddx + qx - px + LB_RV;
            if (bit < 0 || bit > 2 * LB_RV || !((Ma >> bit) & 1u))
              continue;
            const float pxc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bx + px), 0.5f), g.leaf[0]), h.offset[0]);
            const float qxc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * (bx + ddx) + qx), 0.5f), g.leaf[0]), h.offset[0]);
            const float ex = __fsub_rn(pxc, qxc), ey = __fsub_rn(pyc, qyc), ez = __fsub_rn(pzc, qzc);
            float d2 = __fmul_rn(ex, ex);
            d2 = __fadd_rn(d2, __fmul_rn(ey, ey));
            d2 = __fadd_rn(d2, __fmul_rn(ez, ez));
            if (d2 < bp.r2)
              return true;
          }
      }
    }
  }
  return false;
}

// 16-bit union-find in LDS: parents only ever decrease, stale reads are ancestors
__device__ __forceinline__ uint32_t lb_ld16(const uint16_t* par, uint32_t i) { return *reinterpret_cast<const volatile uint16_t*>(par + i); }
__device__ __forceinline__ void lb_st16(uint16_t* par, uint32_t i, uint32_t v) { *reinterpret_cast<volatile uint16_t*>(par + i) = static_cast<uint16_t>(v); }

__device__ __forceinline__ uint32_t lb_find(uint16_t* par, uint32_t v)
{
  uint32_t curr = lb_ld16(par, v);
  if (curr != v)
  {
    uint32_t prev = v, next;
    while (curr > (next = lb_ld16(par, curr)))
    {
      lb_st16(par, prev, next);
      prev = curr;
      curr = next;
    }
  }
  return curr;
}

// compare-and-swap of one 16-bit parent through the 32-bit word that holds it; returns the old value of the half
__device__ __forceinline__ uint32_t lb_cas16(uint16_t* par, uint32_t i, uint32_t expect, uint32_t val)
{
  uint32_t* w = reinterpret_cast<uint32_t*>(par) + (i >> 1);
  const int sh = (i & 1u) * 16;
  uint32_t cur = *reinterpret_cast<volatile uint32_t*>(w);
  for (;;)
  {
    const uint32_t half = (cur >> sh) & 0xffffu;
    if (half != expect)
      return half;
    const uint32_t nw = (cur & ~(0xffffu << sh)) | (val << sh);
    const uint32_t old = atomicCAS(w, cur, nw);
    if (old == cur)
      return expect;
    cur = old;
  }
}

__device__ __forceinline__ uint32_t lb_node(const uint32_t* bits, const uint16_t* pre, uint32_t b)
{
  return pre[b >> 5] + __popc(bits[b >> 5] & ((1u << (b & 31u)) - 1u));
}

__global__ __launch_bounds__(LB_THREADS) void k_brick_ccl_lds(const GridParams g, const BrickParams bp, const LbTables* __restrict__ tab, FrameHdr* hdrs, VoxelArrays va_all,
                                                             uint32_t* __restrict__ labels_all, const unsigned long long* __restrict__ bitmaps,
                                                             const uint32_t* __restrict__ wprefix_all, uint32_t lb_limit, uint32_t* __restrict__ scratch_all, const MapGeom mg, const unsigned long long* __restrict__ mapclose,
                                                             const unsigned long long* __restrict__ mapbits, const CloseRow* __restrict__ crows, int n_crows,
                                                             const UpdateParams up, ClusterRec* __restrict__ table_all, CandMember* __restrict__ cand_all, int write_tables,
                                                             unsigned long long* __restrict__ prof)
{
  __shared__ uint32_t s_bits[LB_BITWORDS + 2];      // brick-lattice bitmap
  __shared__ uint16_t s_pre[LB_BITWORDS];           // exclusive popcount prefix per word = node index of the word's first brick
  __shared__ unsigned long long s_word[LB_MAX];     // node -> occupancy word; phase E: component minima
  __shared__ uint32_t s_xyz[LB_MAX];                // node -> brick coordinates, 10 bits each
  __shared__ uint16_t s_par[LB_MAX];                // union-find
  __shared__ LbTables s_tab;
  __shared__ uint32_t s_wsum[LB_THREADS / 64];
  __shared__ uint32_t s_n, s_nh, s_no;
  const uint32_t FRAME = blockIdx.x;
  FrameHdr& h = hdrs[FRAME];
  const uint32_t V = h.V;
  if (V == 0)
    return;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  uint32_t* labels = labels_all + static_cast<size_t>(FRAME) * g.vox_cap;
  uint32_t* s_cmin = reinterpret_cast<uint32_t*>(s_word);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbx = (h.div_b[0] + 3) >> 2, nby = (h.div_b[1] + 3) >> 2, nbz = (h.div_b[2] + 3) >> 2;
#define LB_STAMP(i)     \
  if (prof && tid == 0) \
  prof[static_cast<size_t>(FRAME) * 16 + (i)] = wall_clock64()
  LB_STAMP(0);
  if (static_cast<long long>(nbx) * nby * nbz > static_cast<long long>(LB_BITWORDS) * 32 || nbx > 1023 || nby > 1023 || nbz > 1023)
  {
    if (tid == 0)
    {
      h.status = CCL_RETRY_STATUS;
      h.V = 0;  // the frame is empty for the rest of the chain; the host re-runs the batch
    }
    return;
  }
  for (int s = tid; s < LB_BITWORDS + 2; s += LB_THREADS)
    s_bits[s] = 0u;
  for (int s = tid; s < static_cast<int>(sizeof(LbTables) / 4); s += LB_THREADS)
    reinterpret_cast<uint32_t*>(&s_tab)[s] = reinterpret_cast<const uint32_t*>(tab)[s];
  __syncthreads();
  constexpr int VU = 8;  // voxel records fetched per lane and round: independent loads in flight
  const uint32_t Vround = (V + 63u) & ~63u;  // whole waves stay in the voxel loops: they shuffle
  // ---- A: mark the occupied bricks, one atomic per run of lanes in the same brick (ranks ascend along x)
  for (uint32_t v0 = tid; v0 < Vround; v0 += LB_THREADS * VU)
  {
    uint32_t bbv[VU];
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      const uint32_t v = v0 + u * LB_THREADS;
      bbv[u] = v < V ? va.bb[v] : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      if (v0 + u * LB_THREADS >= Vround)  // wave-uniform
        break;
      const uint32_t b = bbv[u] == 0xffffffffu ? 0xffffffffu : bbv[u] >> 6;
      const uint32_t prev = __shfl_up(b, 1);
      if (b != 0xffffffffu && (lane == 0 || prev != b))
        atomicOr(&s_bits[b >> 5], 1u << (b & 31u));
    }
  }
  __syncthreads();
  LB_STAMP(1);
  // ---- B: node indices = ranks of the set bits
  constexpr int WPT = (LB_BITWORDS + LB_THREADS - 1) / LB_THREADS;  // consecutive words per thread
  uint32_t cnt = 0;
#pragma unroll
  for (int r = 0; r < WPT; r++)
    if (tid * WPT + r < LB_BITWORDS)
      cnt += __popc(s_bits[tid * WPT + r]);
  {
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane == 63)
      s_wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < LB_THREADS / 64; w++)
    {
      const uint32_t x = s_wsum[w];
      base += w < wave ? x : 0u;
      total += x;
    }
    if (tid == 0)
    {
      s_n = total;
      s_nh = 0;
      s_no = 0;
    }
    if (total > lb_limit)  // <= LB_MAX (lower only in the tests of the fallback)
    {
      if (tid == 0)
      {
        h.status = CCL_RETRY_STATUS;
        h.n_bricks = total;
        h.V = 0;
      }
      return;
    }
    uint32_t run = base + incl - cnt;
#pragma unroll
    for (int r = 0; r < WPT; r++)
      if (tid * WPT + r < LB_BITWORDS)
      {
        s_pre[tid * WPT + r] = static_cast<uint16_t>(run);
        run += __popc(s_bits[tid * WPT + r]);
      }
    for (uint32_t i = tid; i < total; i += LB_THREADS)
    {
      s_par[i] = static_cast<uint16_t>(i);
      s_word[i] = 0ull;
    }
  }
  __syncthreads();
  const uint32_t n = s_n;
  LB_STAMP(2);
  // ---- C: occupancy words and brick coordinates, one atomic per run of lanes in the same brick
  for (uint32_t v0 = tid; v0 < Vround; v0 += LB_THREADS * VU)
  {
    uint32_t bbv[VU];
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      const uint32_t v = v0 + u * LB_THREADS;
      bbv[u] = v < V ? va.bb[v] : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      if (v0 + u * LB_THREADS >= Vround)
        break;
      const uint32_t b = bbv[u] == 0xffffffffu ? 0xffffffffu : bbv[u] >> 6;
      unsigned long long bits = bbv[u] == 0xffffffffu ? 0ull : 1ull << (bbv[u] & 63u);
      // runs of lanes in one brick are cut at 4-lane boundaries (a brick row holds 4 voxels): three shuffles reduce a run
      const uint32_t prev = __shfl_up(b, 1);
      const bool head = (lane & 3) == 0 || prev != b;
#pragma unroll
      for (int s = 1; s < 4; s++)
      {
        const unsigned long long t = __shfl_down(bits, s);
        const uint32_t bs = __shfl_down(b, s);
        const bool hs = __shfl_down(static_cast<int>(head), s) != 0;
        // lane + s belongs to this head's run iff no head sits in (lane, lane + s]
        if (head && (lane & 3) + s < 4 && bs == b && !hs)
          bits |= t;
      }
      if (head && b != 0xffffffffu)
      {
        const uint32_t node = lb_node(s_bits, s_pre, b);
        atomicOr(&s_word[node], bits);
        const uint32_t bz = b / (nbx * nby);
        const uint32_t brem = b - bz * nbx * nby;
        const uint32_t by = brem / nbx;
        s_xyz[node] = (brem - by * nbx) | (by << 10) | (bz << 20);  // every run of the brick writes the same value
      }
    }
  }
  __syncthreads();
  LB_STAMP(3);
  // ---- D: probe, test, union.
  // D-a: every (brick, stencil row) reads one window of the brick bitmap; the occupied neighbours go to a hit list
  //      (node, neighbour node, stencil index) in global scratch.  Neighbours of one window are consecutive bricks of a
  //      lattice row, hence consecutive nodes: one prefix lookup per window.
  // D-b: one hit per lane, flat and balanced: octant matrices; accepted pairs are merged, open pairs go to a second list.
  // D-c: the open pairs that still sit in different components get the exact test.
  uint32_t* hits = scratch_all + static_cast<size_t>(FRAME) * g.vox_cap * 10u;
  const uint32_t hcap = g.vox_cap * 5u;
  uint32_t* opens = hits + hcap;
  {
    const int R = s_tab.R, n_rows = s_tab.n_rows;
    const int sub = tid % LB_LANES;
    const uint32_t n_round = (n + LB_THREADS / LB_LANES - 1) / (LB_THREADS / LB_LANES) * (LB_THREADS / LB_LANES);
    for (uint32_t t = tid / LB_LANES; t < n_round; t += LB_THREADS / LB_LANES)  // wave-uniform trip counts: the reservation below shuffles
    {
      const bool live = t < n;
      const uint32_t xyz = live ? s_xyz[t] : 0u;
      const int bx = xyz & 1023u, by = (xyz >> 10) & 1023u, bz = xyz >> 20;
      const int lo = max(bx - R, 0), hi = min(bx + R, nbx - 1);
      // both rows of the lane are probed before either is consumed: their LDS reads overlap
      constexpr int RPL = LB_MAX_ROWS / LB_LANES;  // rows per lane
      static_assert(RPL == 2, "the reservation below adds up two rows per lane");
      uint32_t winv[RPL], rawv[RPL], nbv[RPL];
      unsigned long long ovv[RPL];
      int shv[RPL];
#pragma unroll
      for (int rr = 0; rr < RPL; rr++)
      {
        const int row = rr * LB_LANES + sub;
        winv[rr] = rawv[rr] = nbv[rr] = 0;
        ovv[rr] = 0;
        shv[rr] = 0;
        if (live && row < n_rows)
        {
          const unsigned long long q0 = reinterpret_cast<const unsigned long long*>(&s_tab.rows[row])[0];
          const unsigned long long q1 = reinterpret_cast<const unsigned long long*>(&s_tab.rows[row])[1];
          const int ddy = static_cast<int8_t>(q0 & 0xffu), ddz = static_cast<int8_t>((q0 >> 8) & 0xffu);
          const uint32_t rw_valid = static_cast<uint32_t>(q0 >> 16) & 0xffu;
          ovv[rr] = (q0 >> 24) | (q1 << 40);  // byte s: stencil index of dx = s - R
          const int ny = by + ddy, nz = bz + ddz;
          if (ny >= 0 && ny < nby && nz < nbz)
          {
            const uint32_t first = static_cast<uint32_t>((nz * nby + ny) * nbx) + lo;
            const uint32_t wi = first >> 5, sh = first & 31u;
            const uint32_t w_lo = s_bits[wi], w_hi = s_bits[wi + 1];
            const uint32_t pre = s_pre[wi];
            const unsigned long long two = static_cast<unsigned long long>(w_lo) | (static_cast<unsigned long long>(w_hi) << 32);
            rawv[rr] = static_cast<uint32_t>(two >> sh) & ((1u << (hi - lo + 1)) - 1u);  // bit j: brick first + j
            shv[rr] = lo - (bx - R);
            winv[rr] = (rawv[rr] << shv[rr]) & rw_valid;  // bit s: the brick at dx = s - R is occupied and in the half stencil
            nbv[rr] = pre + __popc(w_lo & ((1u << sh) - 1u));  // node of the first occupied brick at or after `first`
          }
        }
      }
      // one reservation per wave for both rows
      const uint32_t k = __popc(winv[0]) + (RPL > 1 ? __popc(winv[RPL - 1]) : 0u);
      const uint32_t incl = wave_incl_scan(k);
      uint32_t base = 0;
      if (lane == 63 && incl)
        base = atomicAdd(&s_nh, incl);
      uint32_t pos = __shfl(base, 63) + incl - k;
#pragma unroll
      for (int rr = 0; rr < RPL; rr++)
      {
        uint32_t win = winv[rr];
        while (win)
        {
          const int s = __ffs(static_cast<int>(win)) - 1;
          win &= win - 1;
          const uint32_t o = static_cast<uint32_t>(ovv[rr] >> (8 * s)) & 0xffu;
          const uint32_t t2 = nbv[rr] + __popc(rawv[rr] & ((1u << (s - shv[rr])) - 1u));
          if (pos < hcap)
            hits[pos] = t | (t2 << 13) | (o << 26);
          pos++;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t nh = s_nh;
  if (nh > hcap)
  {
    if (tid == 0)
    {
      h.status = CCL_RETRY_STATUS;
      h.V = 0;
    }
    return;
  }
  if (prof && tid == 0)
    prof[static_cast<size_t>(FRAME) * 16 + 7] = wall_clock64();
  constexpr int HU = 8;
  // a lane takes HU consecutive hits: they mostly share the brick t (the list is in D-a's order), whose root is then found once
  for (uint32_t i0 = tid * HU; i0 < nh; i0 += LB_THREADS * HU)
  {
    uint32_t hv[HU];
#pragma unroll
    for (int u = 0; u < HU; u++)
    {
      const uint32_t i = i0 + u;
      hv[u] = i < nh ? __builtin_nontemporal_load(&hits[i]) : 0xffffffffu;
    }
    // staged so that the LDS reads of the round's hits are issued together (16 waves per CU hide little latency)
    unsigned long long Aw[HU], Bw[HU], Ms[HU], Mm[HU];
    uint32_t pa[HU], pb[HU];
#pragma unroll
    for (int u = 0; u < HU; u++)
    {
      const bool ok = hv[u] != 0xffffffffu;
      const uint32_t t = ok ? hv[u] & 8191u : 0u, t2 = ok ? (hv[u] >> 13) & 8191u : 0u, o = ok ? hv[u] >> 26 : 0u;
      Aw[u] = s_word[t];
      Bw[u] = s_word[t2];
      Ms[u] = s_tab.oct[2 * o];
      Mm[u] = s_tab.oct[2 * o + 1];
      pa[u] = lb_ld16(s_par, t);
      pb[u] = lb_ld16(s_par, t2);
    }
    uint32_t kind[HU];  // 0 nothing, 1 accepted by the octant matrices, 2 open
#pragma unroll
    for (int u = 0; u < HU; u++)
    {
      const uint32_t A8 = lb_oct8(Aw[u]), B8 = lb_oct8(Bw[u]);
      kind[u] = hv[u] == 0xffffffffu ? 0u : lb_octtest(Ms[u], A8, B8) ? 1u : lb_octtest(Mm[u], A8, B8) ? 2u : 0u;
    }
#pragma unroll
    for (int u = 0; u < HU; u++)
    {
      // the open pairs of the round: one reservation per wave
      const unsigned long long m = __ballot(kind[u] == 2u);
      if (m)
      {
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&s_no, static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, leader);
        if (kind[u] == 2u)
          opens[base + __popcll(m & ((1ull << lane) - 1ull))] = hv[u];
      }
    }
    uint32_t cur_t = 0xffffffffu, cur_root = 0;
#pragma unroll
    for (int u = 0; u < HU; u++)
    {
      if (kind[u] != 1u || pa[u] == pb[u])
        continue;
      const uint32_t t = hv[u] & 8191u, t2 = (hv[u] >> 13) & 8191u;
      cur_root = lb_find(s_par, t == cur_t ? cur_root : t);
      cur_t = t;
      uint32_t ra = cur_root, rb = lb_find(s_par, t2);
      if (ra == rb)
        continue;
      cur_root = min(ra, rb);  // whichever way the hooks below go, the smaller root ends above both
      while (ra != rb)
      {
        if (ra < rb)
        {
          const uint32_t tmp = ra;
          ra = rb;
          rb = tmp;
        }
        const uint32_t old = lb_cas16(s_par, ra, ra, rb);
        if (old == ra)
          break;
        ra = old;
      }
    }
  }
  __syncthreads();
  {
    // flatten, then the open pairs
    uint32_t roots[LB_MAX / LB_THREADS];
#pragma unroll
    for (int r = 0; r < LB_MAX / LB_THREADS; r++)
    {
      const uint32_t i = r * LB_THREADS + tid;
      uint32_t root = i < n ? i : 0u, p;
      if (i < n)
        while ((p = lb_ld16(s_par, root)) != root)
          root = p;
      roots[r] = root;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LB_MAX / LB_THREADS; r++)
      if (r * LB_THREADS + tid < n)
        s_par[r * LB_THREADS + tid] = static_cast<uint16_t>(roots[r]);
    __syncthreads();
  }
  const uint32_t no = s_no;
  if (prof && tid == 0)
  {
    prof[static_cast<size_t>(FRAME) * 16 + 8] = wall_clock64();
    prof[static_cast<size_t>(FRAME) * 16 + 9] = nh;
    prof[static_cast<size_t>(FRAME) * 16 + 10] = no;
    prof[static_cast<size_t>(FRAME) * 16 + 11] = n;
  }
  constexpr int OU = 8;  // open pairs fetched per lane and round: independent loads in flight
  for (uint32_t i0 = tid; i0 < no; i0 += LB_THREADS * OU)
  {
    uint32_t ov[OU];
#pragma unroll
    for (int u = 0; u < OU; u++)
    {
      const uint32_t i = i0 + u * LB_THREADS;
      ov[u] = i < no ? opens[i] : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < OU; u++)
    {
    if (ov[u] == 0xffffffffu)
      continue;
    const uint32_t hv = ov[u];
    const uint32_t t = hv & 8191u, t2 = (hv >> 13) & 8191u;
    uint32_t ra = lb_find(s_par, t), rb = lb_find(s_par, t2);
    if (ra == rb)
      continue;
    const uint32_t xa = s_xyz[t], xb = s_xyz[t2];
    const int bx = xa & 1023u, by = (xa >> 10) & 1023u, bz = xa >> 20;
    const int ddx = static_cast<int>(xb & 1023u) - bx, ddy = static_cast<int>((xb >> 10) & 1023u) - by, ddz = static_cast<int>(xb >> 20) - bz;
    if (!lb_pair_conn(s_tab, g, bp, h, s_word[t], s_word[t2], bx, by, bz, ddx, ddy, ddz))
      continue;
    ra = lb_find(s_par, ra);
    rb = lb_find(s_par, rb);
    while (ra != rb)
    {
      if (ra < rb)
      {
        const uint32_t tmp = ra;
        ra = rb;
        rb = tmp;
      }
      const uint32_t old = lb_cas16(s_par, ra, ra, rb);
      if (old == ra)
        break;
      ra = old;
    }
    }
  }
  __syncthreads();
  LB_STAMP(4);
  // ---- E: component minima.  A brick's smallest rank belongs to its lowest set bit (bit order inside a brick is the key
  // order); its rank comes from the occupancy bitmap's prefix array.
  const unsigned long long* bm = bitmaps + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  const uint32_t* wprefix = wprefix_all + static_cast<size_t>(FRAME) * (g.words_cap + 2);
  uint32_t my_root[LB_MAX / LB_THREADS], my_min[LB_MAX / LB_THREADS];
  unsigned long long my_w[LB_MAX / LB_THREADS];
#pragma unroll
  for (int r = 0; r < LB_MAX / LB_THREADS; r++)
  {
    const uint32_t i = r * LB_THREADS + tid;
    my_root[r] = 0xffffffffu;
    my_min[r] = 0xffffffffu;
    my_w[r] = 0ull;
    if (i < n)
    {
      uint32_t root = i, p;
      while ((p = lb_ld16(s_par, root)) != root)
        root = p;
      my_root[r] = root;
      const uint32_t xyz = s_xyz[i];
      const int bx = xyz & 1023u, by = (xyz >> 10) & 1023u, bz = xyz >> 20;
      my_w[r] = s_word[i];
      const int bit = __ffsll(static_cast<long long>(my_w[r])) - 1;
      my_min[r] = static_cast<uint32_t>(((4 * bz + (bit >> 4)) * h.div_b[1] + (4 * by + ((bit >> 2) & 3))) * h.div_b[0] + 4 * bx + (bit & 3));  // key for now
    }
  }
  {
    // the ranks of those keys: all of a lane's bitmap / prefix words are fetched together
    unsigned long long bw[LB_MAX / LB_THREADS];
    uint32_t pw[LB_MAX / LB_THREADS];
#pragma unroll
    for (int r = 0; r < LB_MAX / LB_THREADS; r++)
    {
      const bool ok = my_root[r] != 0xffffffffu;
      bw[r] = ok ? bm[my_min[r] >> 6] : 0ull;
      pw[r] = ok ? wprefix[my_min[r] >> 6] : 0u;
    }
#pragma unroll
    for (int r = 0; r < LB_MAX / LB_THREADS; r++)
      if (my_root[r] != 0xffffffffu)
        my_min[r] = pw[r] + __popcll(bw[r] & ((1ull << (my_min[r] & 63u)) - 1ull));
  }
  __syncthreads();  // roots and words are in registers: flatten the forest, turn the words into the minima
#pragma unroll
  for (int r = 0; r < LB_MAX / LB_THREADS; r++)
  {
    const uint32_t i = r * LB_THREADS + tid;
    if (i < n)
    {
      s_par[i] = static_cast<uint16_t>(my_root[r]);
      s_cmin[i] = 0xffffffffu;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LB_MAX / LB_THREADS; r++)
    if (my_root[r] != 0xffffffffu)
      atomicMin(&s_cmin[my_root[r]], my_min[r]);
  __syncthreads();
  // ---- cluster statistics (size, lattice box, close flag) gathered brick by brick: what k_flatten does voxel by voxel for
  // the other clustering paths.  The whole frame is in this workgroup: every component gets an index and a row in an LDS
  // table (upper half of s_word, free by now); components beyond the table go through global atomics on their slots.
  uint16_t* s_cidx = reinterpret_cast<uint16_t*>(s_word + LB_MAX / 2);               // node (root) -> component index
  uint32_t* st_label = reinterpret_cast<uint32_t*>(s_cidx + LB_MAX);                 // LB_ST_ROWS x {label, count, close, box[6]}
  uint32_t* st_cnt = st_label + LB_ST_ROWS;
  uint32_t* st_close = st_cnt + LB_ST_ROWS;
  int* st_box = reinterpret_cast<int*>(st_close + LB_ST_ROWS);
  if (tid == 0)
    s_nh = 0;  // reused: number of components
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LB_MAX / LB_THREADS; r++)
  {
    const bool is_root = my_root[r] == static_cast<uint32_t>(r * LB_THREADS + tid);
    const unsigned long long m = __ballot(is_root);
    if (!m)
      continue;
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    uint32_t base = 0;
    if (lane == leader)
      base = atomicAdd(&s_nh, static_cast<uint32_t>(__popcll(m)));
    base = __shfl(base, leader);
    if (is_root)
    {
      const uint32_t c = base + __popcll(m & ((1ull << lane) - 1ull));
      const uint32_t label = s_cmin[my_root[r]];
      s_cidx[my_root[r]] = static_cast<uint16_t>(min(c, static_cast<uint32_t>(LB_ST_ROWS)));
      if (c < LB_ST_ROWS)
      {
        st_label[c] = label;
        st_cnt[c] = 0;
        st_close[c] = 0;
        for (int a = 0; a < 3; a++)
        {
          st_box[6 * c + a] = 0x7fffffff;
          st_box[6 * c + 3 + a] = static_cast<int>(0x80000000u);
        }
      }
      else
      {
        // components beyond the table accumulate in their global slots (k_emit left those to this kernel: lean emission);
        // initialised with atomics, which are ordered at L2 with the accumulating atomics of the other lanes
        atomicExch(&va.csize[label], 0u);
        atomicExch(&va.cclose[label], 0u);
        for (int a = 0; a < 3; a++)
        {
          atomicExch(&va.cbox[6 * label + a], 0x7fffffff);
          atomicExch(&va.cbox[6 * label + 3 + a], static_cast<int>(0x80000000u));
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LB_MAX / LB_THREADS; r++)
  {
    const uint32_t i = r * LB_THREADS + tid;
    if (r * LB_THREADS + (tid & ~63) >= n)  // the whole wave is past the last brick
      continue;
    const bool live = i < n;  // lanes past the last brick stay in the shuffles below with neutral values
    const unsigned long long W = live ? my_w[r] : 1ull;
    const uint32_t c = live ? s_cidx[my_root[r]] : 0xffffffffu;
    const uint32_t xyz = live ? s_xyz[i] : 0u;
    const int bx = xyz & 1023u, by = (xyz >> 10) & 1023u, bz = xyz >> 20;
    // extents of the set bits along x, y, z (bit p = x + 4y + 16z)
    unsigned long long t = W | (W >> 16) | (W >> 32) | (W >> 48);
    uint32_t ox = static_cast<uint32_t>(t) & 0xffffu;
    ox = (ox | (ox >> 4) | (ox >> 8) | (ox >> 12)) & 0xfu;
    t = W | (W >> 1);
    t |= t >> 2;  // bit 4y + 16z: row (y,z) is occupied
    unsigned long long ty = t | (t >> 16) | (t >> 32) | (t >> 48);
    const uint32_t oy = (static_cast<uint32_t>(ty) & 1u) | ((static_cast<uint32_t>(ty) >> 3) & 2u) | ((static_cast<uint32_t>(ty) >> 6) & 4u) | ((static_cast<uint32_t>(ty) >> 9) & 8u);
    const uint32_t oz = ((W & 0xffffull) ? 1u : 0u) | ((W & 0xffff0000ull) ? 2u : 0u) | ((W & 0xffff00000000ull) ? 4u : 0u) | ((W >> 48) ? 8u : 0u);
    const int lo[3] = {4 * bx + __ffs(static_cast<int>(ox)) - 1, 4 * by + __ffs(static_cast<int>(oy)) - 1, 4 * bz + __ffs(static_cast<int>(oz)) - 1};
    const int hi[3] = {4 * bx + 31 - __clz(static_cast<int>(ox)), 4 * by + 31 - __clz(static_cast<int>(oy)), 4 * bz + 31 - __clz(static_cast<int>(oz))};
    const uint32_t cnt = __popcll(W);
    // hasCloseTo through the dilated map image, until the component is known to be close
    bool hit = false;
    if (live && mapclose && !(c < LB_ST_ROWS ? st_close[c] : 0u))
    {
      unsigned long long a = W;
      while (a && !hit)
      {
        // eight voxels per round: their image words are fetched together, not one dependent load after the other
        constexpr int CB = 8;
        uint64_t Lq[CB];
        unsigned long long wq[CB];
#pragma unroll
        for (int q = 0; q < CB; q++)
        {
          Lq[q] = ~0ull;
          if (!a)
            continue;
          const int p = __ffsll(static_cast<long long>(a)) - 1;
          a &= a - 1;
          const float cx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bx + (p & 3)), 0.5f), g.leaf[0]), h.offset[0]);
          const float cy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * by + ((p >> 2) & 3)), 0.5f), g.leaf[1]), h.offset[1]);
          const float cz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bz + (p >> 4)), 0.5f), g.leaf[2]), h.offset[2]);
          const int mx_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cx, mg.off[0]), mg.vs_inv)));
          const int my_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cy, mg.off[1]), mg.vs_inv)));
          const int mz_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cz, mg.off[2]), mg.vs_inv)));
          if (mx_ >= 0 && mx_ < mg.sx && my_ >= 0 && my_ < mg.sy && mz_ >= 0 && mz_ < mg.sz)
            Lq[q] = (static_cast<uint64_t>(mz_) * mg.sy + my_) * mg.sx + mx_;
          else  // a centre outside the map (a point on the far face of the operation area): the clipped stencil sweep
            for (int rr = 0; rr < n_crows && !hit; rr++)
              hit = close_row_hit(mg, mapbits, crows[rr], mx_, my_, mz_);
        }
#pragma unroll
        for (int q = 0; q < CB; q++)
          wq[q] = Lq[q] != ~0ull ? mapclose[Lq[q] >> 6] : 0ull;
#pragma unroll
        for (int q = 0; q < CB; q++)
          hit |= Lq[q] != ~0ull && ((wq[q] >> (Lq[q] & 63)) & 1ull);
      }
    }
    // wave level: the lanes hold consecutive bricks, mostly of one component (the ground sheet): its lanes are reduced with
    // shuffles and one lane adds the aggregate; LDS atomics on one row would otherwise serialise (4 cycles each)
    {
      const uint32_t lead = __shfl(c, __ffsll(static_cast<long long>(__ballot(1))) - 1);
      const bool same = c == lead && c < LB_ST_ROWS;
      const unsigned long long m_same = __ballot(same);
      if (__popcll(m_same) >= 8)
      {
        uint32_t rc = same ? cnt : 0u;
        int rlo[3], rhi[3];
        for (int a = 0; a < 3; a++)
        {
          rlo[a] = same ? lo[a] : 0x7fffffff;
          rhi[a] = same ? hi[a] : static_cast<int>(0x80000000u);
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1)
        {
          rc += __shfl_xor(rc, sft);
#pragma unroll
          for (int a = 0; a < 3; a++)
          {
            rlo[a] = min(rlo[a], __shfl_xor(rlo[a], sft));
            rhi[a] = max(rhi[a], __shfl_xor(rhi[a], sft));
          }
        }
        const bool any_hit = __ballot(same && hit) != 0ull;
        if (lane == __ffsll(static_cast<long long>(m_same)) - 1)
        {
          atomicAdd(&st_cnt[lead], rc);
          if (any_hit)
            st_close[lead] = 1u;
          for (int a = 0; a < 3; a++)
          {
            atomicMin(&st_box[6 * lead + a], rlo[a]);
            atomicMax(&st_box[6 * lead + 3 + a], rhi[a]);
          }
        }
        if (same)
          continue;  // folded into the aggregate
      }
    }
    if (!live)
      continue;
    if (c < LB_ST_ROWS)
    {
      atomicAdd(&st_cnt[c], cnt);
      if (hit)
        st_close[c] = 1u;
      for (int a = 0; a < 3; a++)
      {
        atomicMin(&st_box[6 * c + a], lo[a]);
        atomicMax(&st_box[6 * c + 3 + a], hi[a]);
      }
    }
    else
    {
      const uint32_t label = s_cmin[my_root[r]];
      atomicAdd(&va.csize[label], cnt);
      if (hit)
        atomicOr(&va.cclose[label], 1u);
      for (int a = 0; a < 3; a++)
      {
        atomicMin(&va.cbox[6 * label + a], lo[a]);
        atomicMax(&va.cbox[6 * label + 3 + a], hi[a]);
      }
    }
  }
  __syncthreads();
  // the rows leave for the global slots; with write_tables (read-only batches) this kernel also does k_finalize's part:
  // one cluster record per component and, below, the member list of the candidate (far, small enough) clusters
  uint8_t* st_cand = reinterpret_cast<uint8_t*>(st_box + 6 * LB_ST_ROWS);
  ClusterRec* table = table_all + static_cast<size_t>(FRAME) * g.vox_cap;  // (the hit list that lived here is dead)
  CandMember* cands = cand_all + static_cast<size_t>(FRAME) * g.vox_cap;
  auto is_cand = [&](uint32_t close, uint32_t size, const int* box) {
    int ext_ok = 1;
    for (int a = 0; a < 3; a++)
      ext_ok &= (static_cast<float>(box[3 + a] - box[a]) * g.leaf[a] <= up.cand_max_extent);
    return !close && static_cast<int>(size) >= up.min_points && ext_ok;
  };
  {
    const uint32_t nc = min(s_nh, static_cast<uint32_t>(LB_ST_ROWS));
    for (uint32_t c = tid; c < nc; c += LB_THREADS)
    {
      const uint32_t label = st_label[c];
      va.csize[label] = st_cnt[c];
      va.cclose[label] = st_close[c];
      for (int a = 0; a < 6; a++)
        va.cbox[6 * label + a] = st_box[6 * c + a];
      if (write_tables)
      {
        const bool cand = is_cand(st_close[c], st_cnt[c], &st_box[6 * c]);
        st_cand[c] = cand ? 1 : 0;
        ClusterRec rec;
        rec.root = label;
        rec.size = st_cnt[c];
        for (int a = 0; a < 3; a++)
        {
          rec.imin[a] = st_box[6 * c + a];
          rec.imax[a] = st_box[6 * c + 3 + a];
        }
        rec.close = st_close[c];
        rec.cand = cand ? 1u : 0u;
        table[atomicAdd(&h.C, 1u)] = rec;
      }
    }
  }
  __syncthreads();
  LB_STAMP(5);
  for (uint32_t v0 = tid; v0 < Vround; v0 += LB_THREADS * VU)
  {
    uint32_t bbv[VU];
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      const uint32_t v = v0 + u * LB_THREADS;
      bbv[u] = v < V ? va.bb[v] : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      const uint32_t v = v0 + u * LB_THREADS;
      if (v0 + u * LB_THREADS - tid >= Vround)  // block-uniform
        break;
      bool cand = false;
      uint32_t label = 0;
      if (v < V)
      {
        const uint32_t root = s_par[lb_node(s_bits, s_pre, bbv[u] >> 6)];
        label = s_cmin[root];
        labels[v] = label;
        if (write_tables)
        {
          const uint32_t c = s_cidx[root];
          if (c < LB_ST_ROWS)
            cand = st_cand[c] != 0;
          else
          {
            // a component beyond the LDS table: its statistics sit in the global slots (written with atomics above)
            int box[6];
            for (int a = 0; a < 6; a++)
              box[a] = __hip_atomic_load(&va.cbox[6 * label + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t close = __hip_atomic_load(&va.cclose[label], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t size = __hip_atomic_load(&va.csize[label], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cand = is_cand(close, size, box);
            if (v == label)  // the component's first voxel writes its record
            {
              ClusterRec rec;
              rec.root = label;
              rec.size = size;
              for (int a = 0; a < 3; a++)
              {
                rec.imin[a] = box[a];
                rec.imax[a] = box[3 + a];
              }
              rec.close = close;
              rec.cand = cand ? 1u : 0u;
              table[atomicAdd(&h.C, 1u)] = rec;
            }
          }
        }
      }
      if (write_tables)
      {
        const unsigned long long m = __ballot(cand);
        if (m)
        {
          const int leader = __ffsll(static_cast<long long>(m)) - 1;
          uint32_t base = 0;
          if (lane == leader)
            base = atomicAdd(&h.n_cand, static_cast<uint32_t>(__popcll(m)));
          base = __shfl(base, leader);
          if (cand)
          {
            CandMember cm;
            cm.root = label;
            cm.v = v;
            cands[base + __popcll(m & ((1ull << lane) - 1ull))] = cm;
          }
        }
      }
    }
  }
  LB_STAMP(6);
  if (tid == 0)
    h.n_bricks = n;
#undef LB_STAMP
}

}  // namespace vk
