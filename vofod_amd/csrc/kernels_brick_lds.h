// Tables and LDS helpers of the brick-level Euclidean clustering that runs inside one workgroup's LDS (gfx950: 160 KB per
// CU): capacities, the stencil rows / octant matrices / ball tables the host builds (LbTables), the octant test, the exact
// pair test with FLANN's float expression, and the 16-bit union-find.  The kernel itself is k_frame_lds (kernels_frame.h),
// which voxelises and clusters a frame on one LDS image; round 1's stand-alone k_brick_ccl_lds (clustering only, behind a
// separate slab voxeliser) was retired in round 3.  A frame with more than LB_MAX occupied bricks, or a brick lattice
// beyond the LDS bitmap, raises CCL_RETRY_STATUS in its header: the host re-runs the batch with the global kernels
// (kernels_brick.h).
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
constexpr int32_t CF_RETRY_STATUS = 1001;   // internal: more pure-far bricks than the close-first frame kernel takes (a cold map): the batch runs again with the full clustering

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
// (cx,cy,cz): cell of brick A's corner in the frame's own lattice (round 5: the bricks are anchored in the batch's reference
// lattice, a whole-cell shift away); (ddx,ddy,ddz): brick offset of B from A.
__device__ __forceinline__ bool lb_pair_conn(const LbTables& tab, const GridParams& g, const BrickParams& bp, const FrameHdr& h, unsigned long long A, unsigned long long B, int cx,
                                             int cy, int cz, int ddx, int ddy, int ddz)
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
        const float pyc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cy + py), 0.5f), g.leaf[1]), h.offset[1]);
        const float pzc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cz + pz), 0.5f), g.leaf[2]), h.offset[2]);
        const float qyc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cy + 4 * ddy + qy), 0.5f), g.leaf[1]), h.offset[1]);
        const float qzc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cz + 4 * ddz + qz), 0.5f), g.leaf[2]), h.offset[2]);
        // (rare path - boundary cases of the tolerance only: kept rolled, it set the kernel's register peak when unrolled)
#pragma unroll 1
        for (int px = 0; px < 4; px++)
#pragma unroll 1
          for (int qx = 0; qx < 4; qx++)
          {
            if (!((a4 >> px) & 1u) || !((b4 >> qx) & 1u))
              continue;
            const int bit = 4 * ddx + qx - px + LB_RV;
            if (bit < 0 || bit > 2 * LB_RV || !((Ma >> bit) & 1u))
              continue;
            const float pxc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cx + px), 0.5f), g.leaf[0]), h.offset[0]);
            const float qxc = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cx + 4 * ddx + qx), 0.5f), g.leaf[0]), h.offset[0]);
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

}  // namespace vk
