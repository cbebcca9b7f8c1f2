// One frame = one workgroup, bricks first (gfx950: 160 KB of LDS per CU).
//
// Round 1's batched path walked the frame's 11 M-cell lattice in 1 Mi-cell LDS slabs (k_slab_emit: 11 x clear + mark +
// scan of 128 KB, 1400 emission groups, 99.7 % of them empty) and then re-derived the brick graph from the voxel
// records in a second LDS kernel (retired in round 3).  The lattice is sparse, the *brick* lattice (4x4x4 cells
// per brick, 64 times fewer bits) is not large: a whole frame's brick bitmap fits LDS next to its occupied bricks.
// So voxelisation (voxel_grid_weighted.cpp:122-188) runs brick-first and the clustering (vofod_nodelet.cpp:689-698)
// continues on the very same LDS image:
//   k_key2      one pass over the input columns: crops + transform (vofod_nodelet.cpp:625-655), cell of every
//               surviving point (voxel_grid_weighted.cpp:131-136) as a *brick code* (brick coordinates 9 + 9 + 6 bits, then
//               the bit inside the brick: 6 bits), appended in point order (8 consecutive points per thread: points of one ring that fall into
//               one voxel / brick stay neighbours in the list);
//   k_frame_lds 1  brick bitmap of the frame (one LDS atomic per run of codes in the same brick),
//               2  popcount prefix: a brick's node index = its rank among the occupied bricks,
//               3  occupancy words: OR of the codes into the node's 64-bit word; a code whose bit was set already is
//                  an extra point of its voxel (weights, voxel_grid_weighted.cpp:181) and goes to a small record list,
//               4  voxel ranks in the reference's key order (z, y, x) without ever sorting: nodes are ordered
//                  (bz, by, bx); a brick row (bz, by) holds 16 lattice rows (yy, zz) = 16 channels; a segmented wave
//                  scan over the nodes gives every (node, channel) its offset inside the lattice row, the row totals
//                  go through a small per-frame table in global memory (L2) and one block scan in key order gives
//                  every lattice row its base rank; then the voxel records (centre, weight 1, key, node) are stored
//                  at their ranks and the extras add to their voxels' weights,
//               D-E the clustering phases on the same bitmap / words (probe, octant test, exact test
//                  across components only, LDS union-find, component minima, statistics, labels, cluster table).
// A frame beyond the LDS capacities raises CCL_RETRY_STATUS: the host re-runs that batch on the general kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_brick_lds.h"
#include "kernels_slab.h"

namespace vk
{

constexpr int FR_THREADS = 1024;
#ifndef FR_WRITE_BACK_DEF
#define FR_WRITE_BACK_DEF 1
#endif
constexpr bool FR_WRITE_BACK = FR_WRITE_BACK_DEF != 0;  // pass 1 writes the converted codes back (0: pass 3a converts again)
constexpr int FR_BW64 = LB_BITWORDS / 2;  // 64-bit words of the brick-lattice bitmap
#ifndef FR_WAVE_PRIO
#define FR_WAVE_PRIO 3
#endif
constexpr int FR_EREC = LB_MAX / 2;       // extras records kept in LDS (they share the union-find's storage)
constexpr int FR_CHUNKS = LB_MAX / 64;    // 64-node chunks of the rank scans
constexpr int FR_MAX_NBZ = 64;            // brick layers along z
constexpr uint32_t FR_ROWS_MAX = 8192;    // brick rows (nby * nbz) the per-frame row tables cover
constexpr uint32_t FR_CODE_NONE = 0xffffffffu;
constexpr int FR_BB64 = FR_BW64 + 2 + FR_BW64 / 4;  // 64-bit words of the bitmap (+ 2 guard words) followed by its 16-bit prefix array
constexpr uint32_t FR_CNT_CAP = FR_BB64 * 8;       // per-voxel byte counters that fit the same storage
#ifndef FR_VGPR_CAP
#define FR_VGPR_ATTR
#else
#define FR_VGPR_ATTR __attribute__((amdgpu_num_vgpr(FR_VGPR_CAP)))
#endif
#ifndef FR_REG_ROUNDS_DEF
#define FR_REG_ROUNDS_DEF 3
#endif
constexpr int FR_REG_ROUNDS = FR_REG_ROUNDS_DEF;  // rounds (1 024 codes) of a wave's code segment that stay in registers between the passes 3a and 3b
constexpr int CF_MAX = FR_THREADS;                 // close-first clustering: pure-far bricks per frame (one per thread); a frame with more takes the full clustering
#ifndef CF_G_DEF
#define CF_G_DEF 4
#endif
constexpr int CF_G = CF_G_DEF;                     // bricks per thread whose map lookups are in flight together (close-first, first part)
constexpr uint32_t CF_LABEL_NONE = 0xffffffffu;    // label of a voxel outside the far clusters in the far-only debug view

// per-frame scratch in global memory (L2-resident: touched sparsely)
struct FrameScratch
{
  unsigned long long* rowT;  // [F][FR_ROWS_MAX][4]: per brick row, per zz: the four yy channel totals, 16 bits each
  uint32_t* rowQ;            // [F][FR_ROWS_MAX][4]: per brick row and zz: voxels of layer zz in the brick rows before it in node order (the rank of the
                             //   row group's first voxel = this + the slab's constant, k_frame_lds pass c)
  uint32_t* bmin;            // [F][LB_MAX]: per node, the rank of the brick's first voxel
  unsigned long long* bbsave; // [F][FR_BB64]: the brick bitmap + prefix, parked while their LDS holds the per-voxel point counters
  unsigned long long* nodeA; // [F][LB_MAX][4]: per node, per zz: voxels of the earlier bricks of its brick row, per yy channel (16 bits each);
                             //   bit 63 of [3]: the row began inside the node's 64-node chunk (no carry to add)
  float4* frag;              // [F][pt_cap]: the fragile points of the input pass: transformed coordinates, w: their brick code (written once the frame's lattice is known)
  uint32_t keys_cap;         // words of a frame's code list (SlabArrays::keys): the point capacity rounded up to IN_SEG_ALIGN - 16 per-wave segments
};

// 16-byte load through the global address space (the column pointers come out of a struct in memory: the compiler would
// otherwise emit flat loads, which also probe the LDS aperture)
__device__ __forceinline__ float4 ldg_f4(const char* p)
{
  typedef float gf4 __attribute__((ext_vector_type(4)));
  const gf4 v = *reinterpret_cast<const __attribute__((address_space(1))) gf4*>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---- one pass over the input ---------------------------------------------------------------------------------------
// The lattice of a frame hangs on its own bounding box (voxel_grid_weighted.cpp:72-106: offset = floor(min * inv) * leaf -
// align offset), so the cell of a point is only known once every point has been seen.  All such lattices are translates of each
// other by whole cells - up to float rounding.  The input is therefore read ONCE (round 5: by the frame kernel itself, as its
// first phase; rounds 2-4 had a streaming kernel k_key1 in front): bounding box, and every survivor's cell in a *reference*
// lattice anchored at the operation area's corner (the same expressions with the reference offset).  The 4x4x4 bricks the
// frame kernel works on are bricks of THAT lattice (round 5), so a survivor's brick is known the moment the point is read and the
// brick bitmap is built while the input streams in; the frame's own lattice only enters as a whole-cell shift where cells leave
// the kernel (centres, keys, lattice boxes, map cells).  A point whose reference position lies within `eps` cells of a cell
// boundary could land in another cell under the frame's own offset (the float subtraction / product round differently): such
// *fragile* points (a fraction of a percent) are kept aside with their transformed coordinates and encoded with the exact
// expression of the reference once the bounding box is known.  eps is a bound on all rounding differences (see
// fill_ref_lattice), far above them.
struct RefLattice
{
  float off[3];     // reference offset: fl(fl(min_b_ref * leaf) - aco), as voxel_grid_weighted.cpp:80-100 would compute it
  float eps;        // fragile band around cell boundaries, in cells
  int32_t dims[3];  // reference cells per axis
  int32_t nb[3];    // bricks per axis: ceil(dims / 4), at most 512 x 512 x 64 and LB_BITWORDS * 32 in all (the LDS bitmap)
  int32_t on;
};

constexpr int IN_PPT = 8;                                  // consecutive points per thread and round of the input pass
constexpr uint32_t IN_ROUND = FR_THREADS * IN_PPT;         // points per round and workgroup
constexpr uint32_t IN_SEG_ALIGN = 16u * 64u * IN_PPT;      // the code list of a frame: 16 per-wave segments, each a whole number of rounds

// v_min3_f32 / v_max3_f32: two new points per instruction.  A quiet NaN operand is ignored (the other operands decide), which
// makes qNaN the neutral element of both: dropped points are replaced by it once and need no second select.
__device__ __forceinline__ float min3_raw(float a, float b, float c)
{
  float d;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c)
{
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max(|a|, |b|, |c|) in one instruction (source modifiers)
__device__ __forceinline__ float max3_abs(float a, float b, float c)
{
  float d;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// wave-wide float min / max, every step one VALU instruction with a DPP source (lanes without a source keep their value)
__device__ __forceinline__ float wave_fmin(float v)
{
  asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_fmax(float v)
{
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// the six wave reductions of a bounding box in one block: the steps of the six independent chains are interleaved, so that no
// DPP instruction reads a register the instruction before it wrote (the hazard costs two wait states: 42 s_nop when the chains
// run one after the other)
__device__ __forceinline__ void wave_bbox(float (&mn)[3], float (&mx)[3])
{
#define VOFOD_BBOX_STEP(ctl)                          \
  "v_min_f32_dpp %0, %0, %0 " ctl "\n\t"              \
  "v_min_f32_dpp %1, %1, %1 " ctl "\n\t"              \
  "v_min_f32_dpp %2, %2, %2 " ctl "\n\t"              \
  "v_max_f32_dpp %3, %3, %3 " ctl "\n\t"              \
  "v_max_f32_dpp %4, %4, %4 " ctl "\n\t"              \
  "v_max_f32_dpp %5, %5, %5 " ctl "\n\t"
  asm("s_nop 1\n\t" VOFOD_BBOX_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") VOFOD_BBOX_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
          VOFOD_BBOX_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") VOFOD_BBOX_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
              VOFOD_BBOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") VOFOD_BBOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
      : "+v"(mn[0]), "+v"(mn[1]), "+v"(mn[2]), "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]));
#undef VOFOD_BBOX_STEP
#pragma unroll
  for (int c = 0; c < 3; c++)
  {
    mn[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mn[c]), 63));
    mx[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx[c]), 63));
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed f32 operation of a wave-uniform scalar with a pair of points: op_sel_hi:[0,1] makes the upper lane read the LOW dword
// of the scalar operand too, so the constant needs no pair of vector registers (round 3's kernel spent 72 of its 778 vector
// instructions per 8 points on v_mov re-building such pairs: the register budget of a guest wave leaves no room to keep 18 of
// them).  Every lane rounds like the scalar instruction, as before.
__device__ __forceinline__ f32x2 pk_mul_s(float c, f32x2 v)
{
  f32x2 d;
  const unsigned long long c64 = __float_as_uint(c);
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "s"(c64), "v"(v));
  return d;
}
__device__ __forceinline__ f32x2 pk_add_s(float c, f32x2 v)
{
  f32x2 d;
  const unsigned long long c64 = __float_as_uint(c);
  asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "s"(c64), "v"(v));
  return d;
}

// ---- helpers of k_frame_lds ----------------------------------------------------------------------------------------
// node descriptor: brick coordinates (9 + 9 + 6 bits) and, from pass a of phase 4 on, the brick's 2x2x2 octant occupancy
__device__ __forceinline__ uint32_t fr_pack(uint32_t bx, uint32_t by, uint32_t bz) { return bx | (by << 9) | (bz << 18); }
__device__ __forceinline__ int fr_bx(uint32_t d) { return static_cast<int>(d & 511u); }
__device__ __forceinline__ int fr_by(uint32_t d) { return static_cast<int>((d >> 9) & 511u); }
__device__ __forceinline__ int fr_bz(uint32_t d) { return static_cast<int>((d >> 18) & 63u); }
__device__ __forceinline__ uint32_t fr_row(uint32_t d, int nby) { return ((d >> 18) & 63u) * static_cast<uint32_t>(nby) + ((d >> 9) & 511u); }

__device__ __forceinline__ uint32_t fr_node(const unsigned long long* bits64, const uint16_t* pre, uint32_t b)
{
  return pre[b >> 6] + __popcll(bits64[b >> 6] & ((1ull << (b & 63u)) - 1ull));
}

// the four yy channel counts of z layer zz of a brick word, 16 bits each (bit p of W = x + 4y + 16z)
__device__ __forceinline__ unsigned long long fr_chan(unsigned long long W, int zz)
{
  uint32_t x = static_cast<uint32_t>(W >> (16 * zz)) & 0xffffu;
  x = x - ((x >> 1) & 0x5555u);
  x = (x & 0x3333u) + ((x >> 2) & 0x3333u);  // every nibble holds its popcount
  return static_cast<unsigned long long>(x & 0xfu) | (static_cast<unsigned long long>((x >> 4) & 0xfu) << 16) | (static_cast<unsigned long long>((x >> 8) & 0xfu) << 32) |
         (static_cast<unsigned long long>((x >> 12) & 0xfu) << 48);
}

// field-wise exclusive prefix of four 16-bit fields
__device__ __forceinline__ unsigned long long fr_excl16(unsigned long long x) { return (x << 16) + (x << 32) + (x << 48); }
__device__ __forceinline__ uint32_t fr_hsum16(unsigned long long x)
{
  return static_cast<uint32_t>(x & 0xffffu) + static_cast<uint32_t>((x >> 16) & 0xffffu) + static_cast<uint32_t>((x >> 32) & 0xffffu) + static_cast<uint32_t>(x >> 48);
}

__device__ __forceinline__ unsigned long long fr_shfl64(unsigned long long v, int src)
{
  const uint32_t lo = __shfl(static_cast<uint32_t>(v), src), hi = __shfl(static_cast<uint32_t>(v >> 32), src);
  return static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_mov0_64(unsigned long long v)
{
  return static_cast<unsigned long long>(dpp_mov0<CTRL, ROW_MASK>(static_cast<uint32_t>(v))) | (static_cast<unsigned long long>(dpp_mov0<CTRL, ROW_MASK>(static_cast<uint32_t>(v >> 32))) << 32);
}

// Segmented inclusive scan over the wave's lanes: four 64-bit values per lane, a segment starts at every lane whose
// `head` is set.  Afterwards `head` tells whether a head sits at or before the lane (its segment began inside the wave).
// Hillis-Steele steps as DPP moves (VALU only): 1, 2, 4, 8 lanes inside the rows of 16, then lane 15 -> next row, lane 31
// -> upper half; a lane adds what arrives only while no head lies between (its flag is still clear).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void fr_segstep(unsigned long long v[4], uint32_t& f)
{
  unsigned long long t[4];
#pragma unroll
  for (int q = 0; q < 4; q++)
    t[q] = dpp_mov0_64<CTRL, ROW_MASK>(v[q]);
  const uint32_t ft = dpp_mov0<CTRL, ROW_MASK>(f);
  if (!f)
  {
#pragma unroll
    for (int q = 0; q < 4; q++)
      v[q] += t[q];
  }
  f |= ft;
}

__device__ __forceinline__ void fr_segscan(unsigned long long v[4], bool& head)
{
  uint32_t f = head ? 1u : 0u;
  fr_segstep<0x111, 0xf>(v, f);
  fr_segstep<0x112, 0xf>(v, f);
  fr_segstep<0x114, 0xf>(v, f);
  fr_segstep<0x118, 0xf>(v, f);
  fr_segstep<0x142, 0xa>(v, f);
  fr_segstep<0x143, 0xc>(v, f);
  head = f != 0u;
}

// The wave's view of 64 consecutive nodes for the rank scans: channel counts P, their segmented (per brick row)
// inclusive prefix A (carry of the rows that began in earlier chunks added), the node's brick row.
struct FrNodes
{
  unsigned long long W;
  unsigned long long P[4], A[4];
  uint32_t xyz, row;
  bool live, began_here, tail;
};

__device__ __forceinline__ FrNodes fr_load_nodes(const unsigned long long* s_word, const uint32_t* s_xyz, uint32_t i, uint32_t n, int nby, int lane, const unsigned long long (*s_cin)[4],
                                                 uint32_t chunk, bool with_carry)
{
  FrNodes o;
  o.live = i < n;
  o.W = o.live ? s_word[i] : 0ull;
  o.xyz = o.live ? s_xyz[i] : 0u;
  o.row = o.live ? fr_row(o.xyz, nby) : 0xffffffffu;
  uint32_t prev = __shfl_up(o.row, 1);
  if (lane == 0)
  {
    prev = 0xfffffffeu;
    if (o.live && i > 0)
    {
      prev = fr_row(s_xyz[i - 1], nby);
    }
  }
  bool head = o.row != prev;
  uint32_t next = __shfl_down(o.row, 1);
  if (lane == 63)
  {
    next = 0xfffffffeu;
    if (i + 1 < n)
    {
      next = fr_row(s_xyz[i + 1], nby);
    }
  }
  o.tail = o.live && next != o.row;
#pragma unroll
  for (int zz = 0; zz < 4; zz++)
    o.A[zz] = o.P[zz] = fr_chan(o.W, zz);
  fr_segscan(o.A, head);
  o.began_here = head;
  if (with_carry && !head)
  {
#pragma unroll
    for (int zz = 0; zz < 4; zz++)
      o.A[zz] += s_cin[chunk][zz];
  }
  return o;
}

// CFM (round 4): 1 = the close-first instantiation - the clustering phases D-E are replaced by the few steps around the
// pure-far bricks (see behind phase 3a) and the full clustering's code is not in the kernel at all (it set the register
// budget); a frame beyond its capacity raises CF_RETRY_STATUS and the host runs the batch again with CFM = 0, the kernel of
// rounds 2-3.
// Round 5: the kernel reads the input columns itself (phase "in"); PACKED: every frame of the batch has packed float
// columns, 16-byte aligned, a multiple of 4 points (the host checks) - 16-byte loads; otherwise strided 4-byte loads.
template <int CFM, bool PACKED>
__global__ FR_VGPR_ATTR __launch_bounds__(FR_THREADS) void k_frame_lds(const GridParams g, const BrickParams bp, const LbTables* __restrict__ tab, FrameHdr* hdrs, SlabArrays sa, uint32_t pt_cap,
                                                         VoxelArrays va_all, uint32_t* __restrict__ labels_all, uint32_t lb_limit, uint32_t* __restrict__ scratch_all, FrameScratch fs,
                                                         const MapGeom mg, const unsigned long long* __restrict__ mapclose, const unsigned long long* __restrict__ mapbits,
                                                         const CloseRow* __restrict__ crows, int n_crows, const UpdateParams up, ClusterRec* __restrict__ table_all,
                                                         CandMember* __restrict__ cand_all, int write_tables, unsigned long long* __restrict__ prof, const RefLattice rl, const FrameArgs* __restrict__ args,
                                                         int close_first)
{
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) unsigned long long s_bb[FR_BB64];  // brick-lattice bitmap (bit = linear brick id) + exclusive popcount prefix per
                                                                            // 64-bit word; during the counting / rank phases: one byte counter per voxel
  __shared__ unsigned long long s_word[LB_MAX];                                      // node -> occupancy word; phase E: component minima / statistics
  __shared__ uint32_t s_xyz[LB_MAX];                                                 // node -> brick coordinates, 10 bits each
  __shared__ __attribute__((aligned(8))) uint32_t s_x2[FR_EREC];                     // extras records (phases 3-4), then the 16-bit union-find
  __shared__ LbTables s_tab;
  __shared__ unsigned long long s_cin[FR_CHUNKS + 1][4];  // per chunk: tail sums of its last brick row, then the carry into the chunk
  __shared__ uint32_t s_cflag[FR_CHUNKS + 1];
  __shared__ uint32_t s_wsum[FR_THREADS / 64];
  __shared__ uint32_t s_n, s_nh, s_nn, s_nf, s_no, s_ne;
  __shared__ uint32_t s_nhn;  // both adjacent-hit counters of the probe: face neighbours << 17 | others (<= 3 and 10 per brick)
  // close-first clustering (round 4, see the block behind phase 3a): the frame's pure-far bricks
  __shared__ uint16_t s_pf[CF_MAX];     // their nodes
  __shared__ __attribute__((aligned(16))) uint16_t s_pfpar[CF_MAX];  // union-find over the list's indices (set up behind the emission; the rank pass c borrows the storage)
  __shared__ uint32_t s_taint[CF_MAX / 32], s_troot[CF_MAX / 32];  // bit k: list entry k has an edge to a brick with a close voxel / the component rooted at k holds such an entry
  __shared__ uint32_t s_npf, s_nc, s_no2, s_ncand;
  __shared__ __attribute__((aligned(16))) uint32_t s_cfd[32][4];  // (stencil row, direction) descriptors of the close-first edge passes
  __shared__ uint8_t s_cfl[2][32];
  __shared__ uint32_t s_cfn[2];
  // input pass: per-wave results (bounding box as ordered ints, codes appended), the fragile points' counter, the frame's shift
  __shared__ int s_red[FR_THREADS / 64][6];
  __shared__ uint32_t s_wcnt[FR_THREADS / 64];
  __shared__ uint32_t s_nfrag;
  __shared__ __attribute__((aligned(16))) uint32_t s_qc[FR_MAX_NBZ][4];      // ... and the slabs' constants: row-group base rank = rowQ + s_qc[bz]
  __shared__ int s_lat[4];      // the frame's own lattice: div_b, n_cells
  __shared__ float s_latf[3];   // ... and its offset
  __shared__ int s_shift[4];  // whole cells between the reference lattice and the frame's own: frame cell = reference cell - shift; [3]: the frame is empty / failed
  // rank pass c (before the emission, while the close-first union-find is not yet in use): G at the first row of every slab (~0: no
  // brick in the slab; [FR_MAX_NBZ]: totals), and the waves' sums of the block scan
  static_assert(sizeof(unsigned long long) * (FR_MAX_NBZ + 1 + FR_THREADS / 64) <= sizeof(uint16_t) * CF_MAX, "rank pass c borrows the union-find's storage");
  unsigned long long* s_gs = reinterpret_cast<unsigned long long*>(s_pfpar);  // (four 16-bit fields: one per lattice layer zz of a slab)
  unsigned long long* s_w4 = s_gs + FR_MAX_NBZ + 1;
  unsigned long long* s_bits64 = s_bb;
  uint16_t* s_pre = reinterpret_cast<uint16_t*>(s_bb + FR_BW64 + 2);
  uint32_t* s_cnt32 = reinterpret_cast<uint32_t*>(s_bb);
  uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_bits64);
  uint16_t* s_vbase = reinterpret_cast<uint16_t*>(s_x2);  // node -> index of its first voxel in brick order (counting / rank phases)
  uint16_t* s_par = reinterpret_cast<uint16_t*>(s_x2);
  // the classification tails of the batches in front run beside this kernel (process_frames' pipeline): this kernel's waves go
  // first wherever both want to issue
  __builtin_amdgcn_s_setprio(FR_WAVE_PRIO);
  const uint32_t FRAME = blockIdx.x;
  FrameHdr& h = hdrs[FRAME];
  __shared__ int s_mapk[4];  // REFERENCE cell -> map cell offsets, [3]: valid (see below)
  __shared__ __attribute__((aligned(16))) uint32_t s_near[5][4];  // descriptors of the stencil rows that hold adjacent bricks (phase D)
  __shared__ uint32_t s_near_n;
  const VoxelArrays va = frame_voxels(va_all, FRAME, g.vox_cap);
  uint32_t* labels = labels_all + static_cast<size_t>(FRAME) * g.vox_cap;
  uint32_t* s_cmin = reinterpret_cast<uint32_t*>(s_word);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the brick lattice: bricks of the reference lattice - the same for every frame of the batch
  const int nbx = rl.nb[0], nby = rl.nb[1], nbz = rl.nb[2];
#define FR_STAMP(i)     \
  if (prof && tid == 0) \
  prof[static_cast<size_t>(FRAME) * 32 + (i)] = wall_clock64()
  FR_STAMP(0);
  // (diagnostics: the CU this frame runs on - HW_ID: cu 11:8, sh 12, se 15:13; XCC_ID 3:0 - so that the stamps of the frames that
  // followed each other on one CU can be laid end to end: tools/cu_gaps.py)
  if (prof && tid == 0)
    prof[static_cast<size_t>(FRAME) * 32 + 21] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) | static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 4));
  for (int s = tid; s < FR_BW64 + 2; s += FR_THREADS)
    s_bits64[s] = 0ull;
  for (int s = tid; s < static_cast<int>(sizeof(LbTables) / 4); s += FR_THREADS)
    reinterpret_cast<uint32_t*>(&s_tab)[s] = reinterpret_cast<const uint32_t*>(tab)[s];
  if (tid == 0)
  {
    s_ne = 0;
    s_nh = 0;
    s_nhn = 0;
    s_nn = 0;
    s_nf = 0;
    s_no = 0;
    s_npf = 0;
    s_nc = 0;
    s_no2 = 0;
    s_ncand = 0;
    s_nfrag = 0;
  }
  if (tid < CF_MAX / 32)
    s_taint[tid] = s_troot[tid] = 0u;
  __syncthreads();
  // bit of a packed brick in the brick bitmap
  auto brick_lin = [&](uint32_t p) -> uint32_t { return (((p >> 18) & 63u) * static_cast<uint32_t>(nby) + ((p >> 9) & 511u)) * static_cast<uint32_t>(nbx) + (p & 511u); };
  // brick code of a reference cell: packed brick coordinates (fr_pack), then the bit inside the brick
  auto ref_cell_code = [](uint32_t k0, uint32_t k1, uint32_t k2) -> uint32_t {
    return (fr_pack(k0 >> 2, k1 >> 2, k2 >> 2) << 6) | (k0 & 3u) | ((k1 & 3u) << 2) | ((k2 & 3u) << 4);
  };
  // ---- in: the input columns, once (vofod_nodelet.cpp:625-655 CropBox, transformPointCloud, CropBox; voxel_grid_weighted.cpp:58
  // getMinMax3D, :131-136 the cell of a point).  Round r: thread t takes the IN_PPT consecutive points from (r * FR_THREADS + t) *
  // IN_PPT on; the next round's loads are in flight while a round is worked on.  Survivors away from cell boundaries ("solid")
  // set their brick's bit in the LDS bitmap at once (one LDS atomic per run of a thread's consecutive points in one brick) and
  // leave as brick codes in the wave's own segment of the frame's code list, in point order (points of a ring that share a voxel
  // / brick stay neighbours: the later passes merge such runs in registers); no block-wide step inside the loop.
  const FrameArgs a = args[FRAME];  // (a copy: pointers and transform stay in scalar registers)
  const uint32_t n_pts = a.n;
  const uint32_t keys_cap = fs.keys_cap, seg_cap = fs.keys_cap / (FR_THREADS / 64);
  uint32_t* seg = sa.keys + static_cast<size_t>(FRAME) * keys_cap + static_cast<size_t>(wave) * seg_cap;
  float4* fragl = fs.frag + static_cast<size_t>(FRAME) * pt_cap;  // the fragile points: transformed coordinates, then (w) their brick code
  uint32_t wcnt = 0;  // codes in this wave's segment (wave-uniform)
  {
    const uint32_t in_rounds = (n_pts + IN_ROUND - 1u) / IN_ROUND;
    if (static_cast<uint64_t>(in_rounds) * 64u * IN_PPT > seg_cap)
    {
      if (tid == 0)
      {
        h.status = VOFOD_ERR_CAPACITY;  // (a cloud beyond the workspace's code list: the host sizes it for the sensor)
        h.V = 0;
      }
      return;
    }
    float fmn[3] = {INFINITY, INFINITY, INFINITY}, fmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // Branch-free per point: a value lies inside a closed interval iff the median of (value, low, high) is the value itself -
    // one v_med3 + one compare per axis, exact, false for NaN.  A non-finite input can only give a non-finite transformed
    // point, which fails the operation-area test: the explicit isfinite() of the first crop is implied.
    auto inside = [](float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi) == v; };
    // (v_med3 reads one scalar register at most: the upper bounds live in vector registers for the whole loop)
    float ex_hi[3] = {g.ex_max[0], g.ex_max[1], g.ex_max[2]}, op_hi[3] = {g.op_max[0], g.op_max[1], g.op_max[2]};
#pragma unroll
    for (int c = 0; c < 3; c++)
      asm volatile("" : "+v"(ex_hi[c]), "+v"(op_hi[c]));
    const float solid_lim = 0.5f - rl.eps;
    const float qnan = __int_as_float(0x7fc00000);
    // (the brick lattice's dimensions in VECTOR registers for this loop: as scalars they were re-loaded from the kernel arguments
    // inside every branch of the run merging below - the loop runs out of scalar registers -, each load a stall of the wave)
    uint32_t v_nbx = static_cast<uint32_t>(nbx), v_nby = static_cast<uint32_t>(nby);
    asm volatile("" : "+v"(v_nbx), "+v"(v_nby));
    auto brick_lin_v = [&](uint32_t p) -> uint32_t { return (((p >> 18) & 63u) * v_nby + ((p >> 9) & 511u)) * v_nbx + (p & 511u); };
    // a pixel without a return is exactly (0, 0, 0) (ouster_ros: range 0); where the exclude box holds the origin (it is the box around
    // the sensor: vofod_nodelet.cpp:626-629) a wave whose 512 points of a round are all zero - a ring that looks at the sky - has
    // nothing to do
    const bool zero_dropped = g.ex_min[0] <= 0.0f && g.ex_min[1] <= 0.0f && g.ex_min[2] <= 0.0f && g.ex_max[0] >= 0.0f && g.ex_max[1] >= 0.0f && g.ex_max[2] >= 0.0f;
    const uint32_t cell_lim = 2048u;  // (k0 | k1 | k2 << 3) < 2048: the cell fits the code's fields (9 + 2, 9 + 2, 6 + 2 bits)
    auto load_round = [&](uint32_t r, float (&X)[IN_PPT], float (&Y)[IN_PPT], float (&Z)[IN_PPT]) {
      const uint32_t i0 = r * IN_ROUND + ((static_cast<uint32_t>(wave) + r) & (FR_THREADS / 64 - 1)) * (64u * IN_PPT) + static_cast<uint32_t>(lane) * IN_PPT;
      if constexpr (PACKED)
      {
#pragma unroll
        for (int q = 0; q < IN_PPT / 4; q++)
        {
          // (a quad behind the cloud's end reads the last quad instead - the number of points is a multiple of 4, and at least 4
          // here; the index test below drops its points: no conditional load, no registers to clear first)
          const uint64_t o = static_cast<uint64_t>(min(i0 + 4u * q, n_pts - 4u)) * 4;
          const float4 x0 = ldg_f4(a.x + o), y0 = ldg_f4(a.y + o), z0 = ldg_f4(a.z + o);
          X[4 * q] = x0.x, X[4 * q + 1] = x0.y, X[4 * q + 2] = x0.z, X[4 * q + 3] = x0.w;
          Y[4 * q] = y0.x, Y[4 * q + 1] = y0.y, Y[4 * q + 2] = y0.z, Y[4 * q + 3] = y0.w;
          Z[4 * q] = z0.x, Z[4 * q + 1] = z0.y, Z[4 * q + 2] = z0.z, Z[4 * q + 3] = z0.w;
        }
      }
      else
      {
#pragma unroll
        for (int j = 0; j < IN_PPT; j++)
        {
          const uint32_t i = min(i0 + j, n_pts - 1u);  // (clamped: the index test drops it)
          X[j] = ldf(a.x, a.stride, i);
          Y[j] = ldf(a.y, a.stride, i);
          Z[j] = ldf(a.z, a.stride, i);
        }
      }
    };
    for (uint32_t r = 0; r < in_rounds; r++)
    {
      // (No software prefetch of the next round.  Round 4: with the codes' stores between a prefetch and its use every round ended
      // with a full drain of its own stores - vmcnt counts loads and stores alike.  Round 5: the clean form - wait at the top of the
      // iteration, where this round's loads are the youngest operations in flight, then the previous round's stores, held in
      // registers, then the next round's loads, then the work - compiles as intended (one s_waitcnt vmcnt(0) per round, at the
      // top) and measures the same: 274.1 / 277.1 / 278.2 us against 276.1 / 279.3 / 279.1 us per 256 frames, interleaved on one box.
      // The pass is not waiting for its loads: ~60 branches and ~540 vector instructions per wave and round are what it costs.)
      float px[IN_PPT], py[IN_PPT], pz[IN_PPT];
      load_round(r, px, py, pz);
      // (the waves take the round's 512-point pieces in turn: a wave sees every azimuth sector and ring parity - even load)
      const uint32_t i0 = r * IN_ROUND + ((static_cast<uint32_t>(wave) + r) & (FR_THREADS / 64 - 1)) * (64u * IN_PPT) + static_cast<uint32_t>(lane) * IN_PPT;
      if (zero_dropped)
      {
        uint32_t nz_bits = 0;
#pragma unroll
        for (int j = 0; j < IN_PPT; j++)
          nz_bits |= (__float_as_uint(px[j]) | __float_as_uint(py[j]) | __float_as_uint(pz[j])) & 0x7fffffffu;  // (-0.0 is zero too)
        if (!__any(nz_bits != 0u))
          continue;
      }
      uint32_t code[IN_PPT];
      uint32_t cnt = 0, frag_mask = 0;
      float sq0 = 0.0f, sq1 = 0.0f, sq2 = 0.0f;  // transformed coordinates of the thread's FIRST fragile point of the round (8 % of the threads have one, 0.3 % a second)
      // the transform and the cell expression work on PAIRS of consecutive points as packed-f32 operations (v_pk_mul_f32 /
      // v_pk_add_f32: every lane of a packed instruction rounds like the scalar one, so the separately rounded se3 association
      // is kept); the bounding box is two v_min3 / v_max3 per axis and pair with qNaN standing in for dropped points; a point is
      // "solid" when max(|fr - 0.5|) over the axes stays below 0.5 - eps (one v_max3 with |.| modifiers)
#pragma unroll
      for (int jp = 0; jp < IN_PPT / 2; jp++)
      {
        const int j0 = 2 * jp, j1 = 2 * jp + 1;
        const f32x2 X = {px[j0], px[j1]}, Y = {py[j0], py[j1]}, Z = {pz[j0], pz[j1]};
        bool in_ex[2], in_op[2], keep[2];
#pragma unroll
        for (int e = 0; e < 2; e++)
          in_ex[e] = static_cast<int>(inside(X[e], g.ex_min[0], ex_hi[0])) & inside(Y[e], g.ex_min[1], ex_hi[1]) & inside(Z[e], g.ex_min[2], ex_hi[2]);
        f32x2 q[3];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)  // pcl::detail::Transformer<float>::se3: c0*x + (c1*y + (c2*z + c3)), every op rounded
          q[rr] = pk_mul_s(a.tf[4 * rr + 0], X) + (pk_mul_s(a.tf[4 * rr + 1], Y) + pk_add_s(a.tf[4 * rr + 3], pk_mul_s(a.tf[4 * rr + 2], Z)));
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
          in_op[e] = static_cast<int>(inside(q[0][e], g.op_min[0], op_hi[0])) & inside(q[1][e], g.op_min[1], op_hi[1]) & inside(q[2][e], g.op_min[2], op_hi[2]);
          keep[e] = (i0 + j0 + e < n_pts) & !in_ex[e] & in_op[e];
        }
        code[j0] = code[j1] = FR_CODE_NONE;
        if (!__any(keep[0] | keep[1]))
          continue;  // (wave-uniform) 128 dropped points: whole rings look at the sky
        // pcl::getMinMax3D (voxel_grid_weighted.cpp:58) over the kept points
#pragma unroll
        for (int c = 0; c < 3; c++)
        {
          const float m0 = keep[0] ? q[c][0] : qnan, m1 = keep[1] ? q[c][1] : qnan;
          fmn[c] = min3_raw(fmn[c], m0, m1);
          fmx[c] = max3_raw(fmx[c], m0, m1);
        }
        // reference cell (voxel_grid_weighted.cpp:131-136 with the reference offset) and the distance from the cell's middle
        f32x2 fl[3], gmid[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
        {
          const f32x2 t = pk_mul_s(g.inv[c], pk_add_s(-rl.off[c], q[c]));  // (q - off) * inv: adding the negated offset rounds as the subtraction does
          fl[c][0] = floorf(t[0]);
          fl[c][1] = floorf(t[1]);
          const f32x2 half = {0.5f, 0.5f};
          gmid[c] = t - (fl[c] + half);
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
          const uint32_t k0 = static_cast<uint32_t>(static_cast<int>(fl[0][e])), k1 = static_cast<uint32_t>(static_cast<int>(fl[1][e])), k2 = static_cast<uint32_t>(static_cast<int>(fl[2][e]));
          // (negative or huge cells set bits above the fields: one test for all three; a kept point lies inside the operation
          // area, whose cells the reference lattice covers - the test only guards the packing)
          const bool fits = (k0 | k1 | (k2 << 3)) < cell_lim;
          const bool solid = keep[e] & fits & (max3_abs(gmid[0][e], gmid[1][e], gmid[2][e]) <= solid_lim);
          cnt += solid ? 1u : 0u;
          code[j0 + e] = solid ? ref_cell_code(k0, k1, k2) : FR_CODE_NONE;
          const bool fragile = keep[e] & !solid;  // kept aside with its transformed coordinates: encoded with the frame's own offset below
          const bool first_fr = fragile & (frag_mask == 0u);
          sq0 = first_fr ? q[0][e] : sq0;
          sq1 = first_fr ? q[1][e] : sq1;
          sq2 = first_fr ? q[2][e] : sq2;
          frag_mask |= fragile ? (1u << (j0 + e)) : 0u;
        }
      }
      if (__any(cnt != 0u))
      {
        // the occupied bricks: one LDS atomic per run of the thread's consecutive codes in one brick
        uint32_t cur = FR_CODE_NONE;
#pragma unroll
        for (int u = 0; u < IN_PPT; u++)
        {
          if (code[u] == FR_CODE_NONE)
            continue;
          const uint32_t b = code[u] >> 6;
          if (b != cur)
          {
            if (cur != FR_CODE_NONE)
            {
              const uint32_t L = brick_lin_v(cur);
              atomicOr(&s_bits[L >> 5], 1u << (L & 31u));
            }
            cur = b;
          }
        }
        if (cur != FR_CODE_NONE)
        {
          const uint32_t L = brick_lin_v(cur);
          atomicOr(&s_bits[L >> 5], 1u << (L & 31u));
        }
        // the codes, appended to the wave's segment in point order (a wave-level scan: no barrier, no atomic)
        const uint32_t incl = wave_incl_scan(cnt);
        uint32_t* out = seg + wcnt + (incl - cnt);
#pragma unroll
        for (int j = 0; j < IN_PPT; j++)
          if (code[j] != FR_CODE_NONE)
            *out++ = code[j];
        wcnt += __builtin_amdgcn_readlane(incl, 63);
      }
      if (__any(frag_mask != 0u))
      {
        // A fragile point leaves as its transformed coordinates (16 bytes with the slot of its code).  The first fragile point of
        // a thread was kept in registers; a further one (0.3 % of the threads) is loaded again - its line is still in the cache -
        // and transformed with the very expression of the loop above (packed or not, every operation rounds the same).
        const uint32_t fcnt = __popc(frag_mask), fincl = wave_incl_scan(fcnt);
        uint32_t fbase = 0;
        if (lane == 63)
          fbase = atomicAdd(&s_nfrag, fincl);
        fbase = __builtin_amdgcn_readlane(fbase, 63);
        if (frag_mask)
        {
          float4* fout = fragl + fbase + (fincl - fcnt);  // (at most one entry per point: the list holds pt_cap of them)
          *fout++ = make_float4(sq0, sq1, sq2, 0.0f);
          uint32_t rest = frag_mask & (frag_mask - 1u);
          while (rest)
          {
            const uint32_t pi = i0 + static_cast<uint32_t>(__ffs(static_cast<int>(rest)) - 1);
            rest &= rest - 1u;
            const uint64_t st = PACKED ? 4u : a.stride;
            const float p0 = ldf(a.x, st, pi), p1 = ldf(a.y, st, pi), p2 = ldf(a.z, st, pi);
            float t3[3];
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
              t3[rr] = __fadd_rn(__fmul_rn(a.tf[4 * rr + 0], p0), __fadd_rn(__fmul_rn(a.tf[4 * rr + 1], p1), __fadd_rn(__fmul_rn(a.tf[4 * rr + 2], p2), a.tf[4 * rr + 3])));
            *fout++ = make_float4(t3[0], t3[1], t3[2], 0.0f);
          }
        }
      }
    }
    // bounding box of the frame (ordered ints: +-inf where a wave kept nothing - the identity of the min / max below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's codes and fragile points have left (other lanes / waves read them below)
    wave_bbox(fmn, fmx);
    if (lane == 0)
    {
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        s_red[wave][c] = f2ord(fmn[c]);
        s_red[wave][3 + c] = f2ord(fmx[c]);
      }
      s_wcnt[wave] = wcnt;
    }
  }
  FR_STAMP(16);
  __syncthreads();
  FR_STAMP(17);
  // ---- the frame's own lattice (voxel_grid_weighted.cpp:61-113) from the bounding box; whole cells between it and the reference
  // lattice: both offsets are floats, their difference times inv lies within eps of an integer
  if (tid == 0)
  {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {static_cast<int>(0x80000000u), static_cast<int>(0x80000000u), static_cast<int>(0x80000000u)};
    uint32_t tot = s_nfrag;
    for (int w = 0; w < FR_THREADS / 64; w++)
    {
      tot += s_wcnt[w];
      for (int c = 0; c < 3; c++)
      {
        mn[c] = min(mn[c], s_red[w][c]);
        mx[c] = max(mx[c], s_red[w][3 + c]);
      }
    }
    // (the lattice is worked out on a copy in registers and leaves for the header in one go: the header lives in global memory,
    // and reading fields back right after writing them cost this single thread - and the 1023 waiting for it - ~20 us of round trips)
    FrameHdr hl;
    for (int c = 0; c < 3; c++)
    {
      hl.bb_min[c] = mn[c];
      hl.bb_max[c] = mx[c];
      hl.offset[c] = 0.0f;
      hl.min_b[c] = hl.div_b[c] = 0;
    }
    hl.n_in = tot;
    hl.status = VOFOD_OK;
    hl.n_cells = hl.n_words = hl.need_words = 0;
    grid_of_frame(g, hl);  // (an empty frame or a lattice beyond the index range: n_in = 0, status set)
    for (int c = 0; c < 3; c++)
    {
      h.bb_min[c] = hl.bb_min[c];
      h.bb_max[c] = hl.bb_max[c];
      h.offset[c] = hl.offset[c];
      h.min_b[c] = hl.min_b[c];
      h.div_b[c] = hl.div_b[c];
      s_lat[c] = hl.div_b[c];
      s_latf[c] = hl.offset[c];
    }
    h.n_in = hl.n_in;
    if (hl.status != VOFOD_OK)
      h.status = hl.status;
    h.n_cells = hl.n_cells;
    h.n_words = hl.n_words;
    h.need_words = hl.need_words;
    s_lat[3] = static_cast<int>(hl.n_cells);
    s_shift[3] = hl.n_in == 0 ? 1 : 0;
    for (int c = 0; c < 3; c++)
      s_shift[c] = hl.n_in ? static_cast<int>(rint((static_cast<double>(hl.offset[c]) - static_cast<double>(rl.off[c])) * static_cast<double>(g.inv[c]))) : 0;
  }
  FR_STAMP(18);
  __syncthreads();
  FR_STAMP(19);
  if (s_shift[3])
    return;  // k_init_hdr left V = C = 0
  const uint32_t n_frag = s_nfrag;
  uint32_t n_keys = n_frag;  // (codes of the frame: diagnostics only)
  for (int w = 0; w < FR_THREADS / 64; w++)
    n_keys += s_wcnt[w];
  const int sh0 = s_shift[0], sh1 = s_shift[1], sh2 = s_shift[2];
  // frame cell of reference cell 0 (frame cell = 4 * brick + bit + o)
  const int o0 = -sh0, o1 = -sh1, o2 = -sh2;
  const int dv0 = s_lat[0], dv1 = s_lat[1], dv2 = s_lat[2];  // the frame's own lattice
  const int dx = dv0, dxy = dv0 * dv1;
  const uint32_t n_cells = static_cast<uint32_t>(s_lat[3]);
  const float hoff0 = s_latf[0], hoff1 = s_latf[1], hoff2 = s_latf[2];
  uint32_t* extras_g = sa.extras + static_cast<size_t>(FRAME) * pt_cap;
  constexpr int KPT = 16;  // consecutive codes per thread and round: points of one ring that share a voxel / brick are merged in registers
  // a round of the wave's own segment: KPT consecutive codes per lane (the segment starts 16-byte aligned)
  auto load_codes = [&](uint32_t base, uint32_t c[KPT]) {
    if (base + KPT <= wcnt)
    {
#pragma unroll
      for (int q = 0; q < KPT / 4; q++)
      {
        const uint4 v = *reinterpret_cast<const uint4*>(seg + base + 4 * q);
        c[4 * q] = v.x, c[4 * q + 1] = v.y, c[4 * q + 2] = v.z, c[4 * q + 3] = v.w;
      }
    }
    else
    {
#pragma unroll
      for (int u = 0; u < KPT; u++)
        c[u] = base + u < wcnt ? seg[base + u] : FR_CODE_NONE;
    }
  };
  if (tid == 0)
  {
    // Is the frame's lattice the map's lattice shifted by whole voxels?  (It is whenever the grid is aligned to the map, which
    // vofod_nodelet.cpp:664 always does.)  The float expression hasCloseTo's caller evaluates - floor((centre - map offset) / vs) -
    // is then cell + K for every cell: checked at both ends of every axis with the expression itself, the fractional part
    // far from a cell boundary (the expression's rounding error grows by ~1e-7 per cell).  Stored for REFERENCE cells: K - shift.
    const int dvs[3] = {dv0, dv1, dv2};
    const float hoffs[3] = {hoff0, hoff1, hoff2};
    const int shs[3] = {sh0, sh1, sh2};
    int ok = 1;
    for (int ax = 0; ax < 3; ax++)
    {
      int K = 0;
      for (int e = 0; e < 2; e++)
      {
        const int k = e == 0 ? 0 : dvs[ax] - 1;
        const float c = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k), 0.5f), g.leaf[ax]), hoffs[ax]);
        const float t = __fmul_rn(__fsub_rn(c, mg.off[ax]), mg.vs_inv);
        const float fl = floorf(t), fr = t - fl;
        const int Ke = static_cast<int>(fl) - k;
        if (e == 0)
          K = Ke;
        ok &= (Ke == K) && fr > 0.25f && fr < 0.75f;
      }
      s_mapk[ax] = K - shs[ax];
    }
    s_mapk[3] = ok;
  }
  // ---- the fragile points: exact expression with the frame's own offset (voxel_grid_weighted.cpp:131-136).  A cell outside the
  // frame's lattice (a rounding artefact) is aliased through the linear index as the reference does (:137) and dropped when it
  // leaves the index range.  Their codes stay in the side list (w of the entry): passes 3a / 3b read them there.
  for (uint32_t i = tid; i < n_frag; i += FR_THREADS)
  {
    const float4 q = fragl[i];
    int k0 = static_cast<int>(floorf(__fmul_rn(__fsub_rn(q.x, hoff0), g.inv[0])));
    int k1 = static_cast<int>(floorf(__fmul_rn(__fsub_rn(q.y, hoff1), g.inv[1])));
    int k2 = static_cast<int>(floorf(__fmul_rn(__fsub_rn(q.z, hoff2), g.inv[2])));
    uint32_t cd = FR_CODE_NONE;
    bool ok = true;
    // one unsigned compare per axis: negative values wrap above every lattice size
    if (static_cast<uint32_t>(k0) >= static_cast<uint32_t>(dv0) || static_cast<uint32_t>(k1) >= static_cast<uint32_t>(dv1) || static_cast<uint32_t>(k2) >= static_cast<uint32_t>(dv2))
    {
      const uint32_t key = static_cast<uint32_t>(k0 + k1 * dx + k2 * dxy);
      ok = key < n_cells;
      if (ok)
      {
        k2 = static_cast<int>(key / static_cast<uint32_t>(dxy));
        const uint32_t rem = key - static_cast<uint32_t>(k2) * dxy;
        k1 = static_cast<int>(rem / static_cast<uint32_t>(dx));
        k0 = static_cast<int>(rem - static_cast<uint32_t>(k1) * dx);
      }
    }
    if (ok)
    {
      const uint32_t r0 = static_cast<uint32_t>(k0 + sh0), r1 = static_cast<uint32_t>(k1 + sh1), r2 = static_cast<uint32_t>(k2 + sh2);
      if (r0 < 4u * static_cast<uint32_t>(nbx) && r1 < 4u * static_cast<uint32_t>(nby) && r2 < 4u * static_cast<uint32_t>(nbz))  // (the frame's lattice lies inside the reference lattice)
      {
        cd = ref_cell_code(r0, r1, r2);
        const uint32_t L = brick_lin(cd >> 6);
        atomicOr(&s_bits[L >> 5], 1u << (L & 31u));
      }
    }
    reinterpret_cast<uint32_t*>(fragl + i)[3] = cd;
  }
  FR_STAMP(20);
  __syncthreads();
  FR_STAMP(1);
  // ---- 2: node indices = ranks of the set bits
  constexpr int WPT = (FR_BW64 + FR_THREADS - 1) / FR_THREADS;  // consecutive 64-bit words per thread
  {
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < WPT; r++)
      if (tid * WPT + r < FR_BW64)
        cnt += __popcll(s_bits64[tid * WPT + r]);
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane == 63)
      s_wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < FR_THREADS / 64; w++)
    {
      const uint32_t x = s_wsum[w];
      base += w < wave ? x : 0u;
      total += x;
    }
    if (tid == 0)
      s_n = total;
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
      if (tid * WPT + r < FR_BW64)
      {
        s_pre[tid * WPT + r] = static_cast<uint16_t>(run);
        run += __popcll(s_bits64[tid * WPT + r]);
      }
    for (uint32_t i = tid; i < total; i += FR_THREADS)
      s_word[i] = 0ull;
  }
  __syncthreads();
  const uint32_t n = s_n;
  const uint32_t n_chunks = (n + 63u) >> 6;
  FR_STAMP(2);
  // ---- 3a: occupancy words.  A lane ORs the run of its consecutive codes that share a brick with one LDS atomic per
  // 32-bit half, and rewrites its codes as node * 64 + bit: the counting pass needs no bitmap lookup.  Every wave walks its own
  // segment of the code list (written by itself in the input pass: no other wave's stores are read here).  The first FR_REG_ROUNDS
  // rounds of the segment (1 024 codes each) are fetched ONCE and stay in registers through both passes (rounds 2-4 read the
  // list three times and wrote it twice: 5 x 47 MB per batch); a segment beyond them (a frame of more than ~49 k survivors) goes
  // through the list in global memory as before.
  constexpr int RR = FR_REG_ROUNDS;
  uint32_t creg[RR > 0 ? RR : 1][KPT];
#pragma unroll
  for (int r = 0; r < RR; r++)
  {
    if (static_cast<uint32_t>(r) * 64u * KPT < wcnt)  // (wave-uniform)
      load_codes(static_cast<uint32_t>(r) * 64u * KPT + lane * KPT, creg[r]);
    else
    {
#pragma unroll
      for (int u = 0; u < KPT; u++)
        creg[r][u] = FR_CODE_NONE;
    }
  }
  {
    auto or_word = [&](uint32_t node, unsigned long long acc) {
      uint32_t* w32 = reinterpret_cast<uint32_t*>(&s_word[node]);
      const uint32_t lo = static_cast<uint32_t>(acc), hi = static_cast<uint32_t>(acc >> 32);
      if (lo)
        atomicOr(&w32[0], lo);
      if (hi)
        atomicOr(&w32[1], hi);
    };
    auto words_round = [&](uint32_t (&c)[KPT]) {
      uint32_t cur = FR_CODE_NONE, cur_node = 0;
      unsigned long long acc = 0ull;
#pragma unroll
      for (int u = 0; u < KPT; u++)
      {
        if (c[u] == FR_CODE_NONE)
          continue;
        const uint32_t b = c[u] >> 6;
        if (b != cur)
        {
          if (cur != FR_CODE_NONE)
            or_word(cur_node, acc);
          cur = b;
          cur_node = fr_node(s_bits64, s_pre, brick_lin(b));
          s_xyz[cur_node] = b;  // the code carries the brick's packed coordinates; every run of the brick writes the same value
          acc = 0ull;
        }
        acc |= 1ull << (c[u] & 63u);
        c[u] = (cur_node << 6) | (c[u] & 63u);
      }
      if (cur != FR_CODE_NONE)
        or_word(cur_node, acc);
    };
#pragma unroll
    for (int r = 0; r < RR; r++)
      if (static_cast<uint32_t>(r) * 64u * KPT < wcnt)
        words_round(creg[r]);
    for (uint32_t rb = static_cast<uint32_t>(RR) * 64u * KPT; rb < wcnt; rb += 64u * KPT)
    {
      const uint32_t base = rb + lane * KPT;
      uint32_t c[KPT];
      load_codes(base, c);
      words_round(c);
      if (base + KPT <= wcnt)
      {
#pragma unroll
        for (int q = 0; q < KPT / 4; q++)
          *reinterpret_cast<uint4*>(seg + base + 4 * q) = make_uint4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
      }
      else
      {
#pragma unroll
        for (int u = 0; u < KPT; u++)
          if (base + u < wcnt)
            seg[base + u] = c[u];
      }
    }
    // the fragile points' codes (the side list: a fraction of a percent of the frame), one per thread and round
    for (uint32_t i = tid; i < n_frag; i += FR_THREADS)
    {
      uint32_t* cw = reinterpret_cast<uint32_t*>(fragl + i) + 3;
      const uint32_t cd = *cw;
      if (cd == FR_CODE_NONE)
        continue;
      const uint32_t node = fr_node(s_bits64, s_pre, brick_lin(cd >> 6));
      s_xyz[node] = cd >> 6;
      or_word(node, 1ull << (cd & 63u));
      *cw = (node << 6) | (cd & 63u);
    }
  }
  __syncthreads();
  FR_STAMP(3);
  // ---- 3b: weights (voxel_grid_weighted.cpp:181).  The bitmap is parked in global memory; its LDS becomes one byte
  // counter per voxel, indexed in brick order: first voxel of the node + set bits below.  Consecutive equal codes of a
  // thread add once.  A counter that would pass 255 is undone and the points go to the frame's record list
  // (node * 64 + bit | (points - 1) << 25), added to the stored weights after the emission.
  uint32_t V = 0;
  {
    ulonglong2* bsave = reinterpret_cast<ulonglong2*>(fs.bbsave + static_cast<size_t>(FRAME) * FR_BB64);
    for (int i = tid; i < FR_BB64 / 2; i += FR_THREADS)
      bsave[i] = reinterpret_cast<const ulonglong2*>(s_bb)[i];
    constexpr int NPT = LB_MAX / FR_THREADS;  // consecutive nodes per thread
    uint32_t pc[NPT], sum = 0;
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
      const uint32_t i = tid * NPT + r;
      pc[r] = i < n ? __popcll(s_word[i]) : 0u;
      sum += pc[r];
    }
    const uint32_t incl = wave_incl_scan(sum);
    if (lane == 63)
      s_wsum[wave] = incl;
    __syncthreads();  // (also: every thread has parked its part of the bitmap)
    uint32_t run = incl - sum;
    for (int w = 0; w < FR_THREADS / 64; w++)
    {
      const uint32_t x = s_wsum[w];
      run += w < wave ? x : 0u;
      V += x;
    }
    if (V > g.vox_cap)
    {
      if (tid == 0)
      {
        h.status = VOFOD_ERR_CAPACITY;  // as k_scan_b
        h.V = 0;
      }
      return;
    }
    if (V > FR_CNT_CAP || V > 65535u)
    {
      if (tid == 0)
      {
        h.status = CCL_RETRY_STATUS;  // more voxels than byte counters: the batch takes the general kernels
        h.V = 0;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
      const uint32_t i = tid * NPT + r;
      if (i < n)
        s_vbase[i] = static_cast<uint16_t>(run);
      run += pc[r];
    }
    for (uint32_t i = tid; i < (V + 15u) / 16u; i += FR_THREADS)
      reinterpret_cast<uint4*>(s_bb)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  {
    auto count = [&](uint32_t code, uint32_t add) {
      const uint32_t node = code >> 6, bit = code & 63u;
      const uint32_t idx = s_vbase[node] + __popcll(s_word[node] & ((1ull << bit) - 1ull));
      const uint32_t sh = 8u * (idx & 3u);
      const uint32_t old = atomicAdd(&s_cnt32[idx >> 2], add << sh);
      if (((old >> sh) & 0xffu) + add > 255u)
      {
        atomicSub(&s_cnt32[idx >> 2], add << sh);
        extras_g[atomicAdd(&s_ne, 1u)] = code | ((add - 1u) << 25);  // at most one record per point: the list holds pt_cap entries
      }
    };
    auto count_round = [&](const uint32_t (&c)[KPT]) {
      uint32_t last = FR_CODE_NONE, cntl = 0;
#pragma unroll
      for (int u = 0; u < KPT; u++)
      {
        if (c[u] == FR_CODE_NONE)
          continue;
        if (c[u] == last)
        {
          cntl++;
          continue;
        }
        if (last != FR_CODE_NONE)
          count(last, cntl);
        last = c[u];
        cntl = 1;
      }
      if (last != FR_CODE_NONE)
        count(last, cntl);
    };
#pragma unroll
    for (int r = 0; r < RR; r++)
      if (static_cast<uint32_t>(r) * 64u * KPT < wcnt)
        count_round(creg[r]);
    for (uint32_t rb = static_cast<uint32_t>(RR) * 64u * KPT; rb < wcnt; rb += 64u * KPT)
    {
      uint32_t c[KPT];
      load_codes(rb + lane * KPT, c);
      count_round(c);
    }
    for (uint32_t i = tid; i < n_frag; i += FR_THREADS)
    {
      const uint32_t cd = reinterpret_cast<const uint32_t*>(fragl + i)[3];
      if (cd != FR_CODE_NONE)
        count(cd, 1u);
    }
  }
  __syncthreads();
  FR_STAMP(15);
  // ---- close first (round 4).  The reference only ever uses the FAR clusters (findCloseFarClusters, vofod_nodelet.cpp:727-748:
  // a cluster is close as soon as ONE member has a background voxel within hasCloseTo's stencil; close clusters feed nothing
  // but a per-voxel map update, :946) - and on a warmed map the ground sheet and the buildings, one giant close component, are
  // > 95 % of a frame.  The dilated map image answers hasCloseTo for a voxel with one bit, and a 4x4x4 brick is a clique under
  // the tolerance (brick-level clustering is only planned then), so:
  //   * a brick that holds a close voxel belongs to a close cluster as a whole;
  //   * a far cluster is a connected component of bricks WITHOUT any close voxel ("pure-far" bricks) that has no edge to a
  //     brick with one: walk from any of its voxels towards a close voxel - the last far voxel on the way has that edge.
  // Here: which bricks are pure far (one lookup of the dilated image per occupied lattice row of a brick, until the first hit).
  // Behind the emission: edges and unions around those few bricks only (tens to hundreds per frame instead of ~5 000).
  // Same member lists, sizes, smallest members, hence the same candidates and detections as the full clustering; that one
  // stays for the debug view of ALL clusters and for a frame with more than CF_MAX pure-far bricks (a cold map).
  constexpr bool cf = CFM != 0;
  if constexpr (cf)
  {
    {
      constexpr int NPT = LB_MAX / FR_THREADS;
      const bool mapk_ok = s_mapk[3] != 0;
      const int K0 = s_mapk[0], K1 = s_mapk[1], K2 = s_mapk[2];
      // one voxel by itself (a brick that leaves the map, or a lattice that is no translate of the map's): as phase E; (k0, k1, k2): REFERENCE cell
      auto voxel_close = [&](int k0, int k1, int k2) -> bool {
        int mx_, my_, mz_;
        if (mapk_ok)
        {
          mx_ = k0 + K0;
          my_ = k1 + K1;
          mz_ = k2 + K2;
        }
        else
        {
          const float cx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k0 + o0), 0.5f), g.leaf[0]), hoff0);
          const float cy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k1 + o1), 0.5f), g.leaf[1]), hoff1);
          const float cz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k2 + o2), 0.5f), g.leaf[2]), hoff2);
          mx_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cx, mg.off[0]), mg.vs_inv)));
          my_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cy, mg.off[1]), mg.vs_inv)));
          mz_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cz, mg.off[2]), mg.vs_inv)));
        }
        if (mx_ >= 0 && mx_ < mg.sx && my_ >= 0 && my_ < mg.sy && mz_ >= 0 && mz_ < mg.sz)
        {
          const uint64_t L = (static_cast<uint64_t>(mz_) * mg.sy + my_) * mg.sx + mx_;
          return ((mapclose[L >> 6] >> (L & 63)) & 1ull) != 0ull;
        }
        bool hit = false;  // a centre outside the map (a point on the far face of the operation area): the clipped stencil sweep
        for (int rr = 0; rr < n_crows && !hit; rr++)
          hit = close_row_hit(mg, mapbits, crows[rr], mx_, my_, mz_);
        return hit;
      };
      // Rounds: every round looks up ONE occupied lattice row (4 cells along x: 4 bits of the image, fetched as the two bytes
      // that hold them) of each of the thread's bricks that is still undecided - the lookups of a round are in flight
      // together, CF_G bricks per thread at a time.  Consecutive lanes hold consecutive nodes, i.e. neighbouring bricks of a
      // brick row: their lookups fall into the same lines of the image.
      const unsigned char* mc8 = reinterpret_cast<const unsigned char*>(mapclose);
#pragma unroll
      for (int g0 = 0; g0 < NPT; g0 += CF_G)
      {
        uint32_t rem[CF_G];  // bit r: lattice row r = yy + 4 zz of the brick is occupied and not looked up yet
        uint32_t far_m = 0;  // bit j: the group's j-th brick has shown no close voxel so far
#pragma unroll
        for (int j = 0; j < CF_G; j++)
        {
          const uint32_t i = (g0 + j) * FR_THREADS + tid;
          rem[j] = 0u;
          if (g0 + j < NPT && i < n)
          {
            unsigned long long t = s_word[i];
            t |= t >> 1;
            t |= t >> 2;
            uint32_t lo = static_cast<uint32_t>(t) & 0x11111111u, hi = static_cast<uint32_t>(t >> 32) & 0x11111111u;
            lo = (lo | (lo >> 3) | (lo >> 6) | (lo >> 9)) & 0x000f000fu;
            hi = (hi | (hi >> 3) | (hi >> 6) | (hi >> 9)) & 0x000f000fu;
            rem[j] = (lo & 0xfu) | (lo >> 12) | ((hi & 0xfu) << 8) | ((hi >> 4) & 0xf000u);
            far_m |= 1u << j;
          }
        }
        for (;;)
        {
          uint32_t val[CF_G], meta[CF_G];  // meta: nibble | shift << 4 | looked up << 8
          bool any = false;
#pragma unroll
          for (int j = 0; j < CF_G; j++)
          {
            val[j] = meta[j] = 0u;
            if (!rem[j])
              continue;
            any = true;
            const uint32_t i = (g0 + j) * FR_THREADS + tid;
            const int row = __ffs(static_cast<int>(rem[j])) - 1;
            rem[j] &= rem[j] - 1u;
            const uint32_t xyz = s_xyz[i];
            const uint32_t nib = static_cast<uint32_t>(s_word[i] >> (4 * row)) & 0xfu;
            const int k0 = 4 * fr_bx(xyz), k1 = 4 * fr_by(xyz) + (row & 3), k2 = 4 * fr_bz(xyz) + (row >> 2);
            const int mx0 = k0 + K0, my_ = k1 + K1, mz_ = k2 + K2;
            if (mapk_ok && mx0 >= 0 && mx0 + 3 < mg.sx && my_ >= 0 && my_ < mg.sy && mz_ >= 0 && mz_ < mg.sz)
            {
              const uint64_t L = (static_cast<uint64_t>(mz_) * mg.sy + my_) * mg.sx + mx0;
              meta[j] = nib | (static_cast<uint32_t>(L & 7u) << 4) | 0x100u;
              const unsigned char* q = mc8 + (L >> 3);  // (the image ends with two guard words)
              val[j] = static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8);
            }
            else
            {
              bool hit = false;
              for (int xx = 0; xx < 4 && !hit; xx++)
                if ((nib >> xx) & 1u)
                  hit = voxel_close(k0 + xx, k1, k2);
              if (hit)
              {
                rem[j] = 0u;
                far_m &= ~(1u << j);
              }
            }
          }
          if (!any)
            break;
#pragma unroll
          for (int j = 0; j < CF_G; j++)
            if ((meta[j] & 0x100u) && ((val[j] >> ((meta[j] >> 4) & 7u)) & meta[j] & 0xfu))
            {
              rem[j] = 0u;
              far_m &= ~(1u << j);
            }
        }
#pragma unroll
        for (int j = 0; j < CF_G; j++)
        {
          const bool pf = (far_m >> j) & 1u;
          const unsigned long long m = __ballot(pf);
          if (!m)
            continue;
          const int leader = __ffsll(static_cast<long long>(m)) - 1;
          uint32_t base = 0;
          if (lane == leader)
            base = atomicAdd(&s_npf, static_cast<uint32_t>(__popcll(m)));
          base = __builtin_amdgcn_readlane(base, leader);
          const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
          if (pf && pos < static_cast<uint32_t>(CF_MAX))
            s_pf[pos] = static_cast<uint16_t>((g0 + j) * FR_THREADS + tid);
        }
      }
      __syncthreads();
      if (s_npf > static_cast<uint32_t>(CF_MAX))
      {
        if (tid == 0)
        {
          h.status = CF_RETRY_STATUS;  // (a cold map: nearly every brick is pure far) the batch takes the full clustering
          h.n_bricks = s_npf;
          h.V = 0;
        }
        return;
      }
    }
  }
  FR_STAMP(14);
  // ---- 4: ranks in key order.  Pass a: one segmented scan per 64-node chunk; the per-node channel prefixes go to the
  // frame's scratch in global memory (L2), the sums of the chunk's last brick row to LDS (rows may span chunks).
  unsigned long long* rowT = fs.rowT + static_cast<size_t>(FRAME) * FR_ROWS_MAX * 4;
  uint32_t* rowQ = fs.rowQ + static_cast<size_t>(FRAME) * FR_ROWS_MAX * 4;
  uint32_t* bmin_g = fs.bmin + static_cast<size_t>(FRAME) * LB_MAX;
  ulonglong2* nodeA = reinterpret_cast<ulonglong2*>(fs.nodeA + static_cast<size_t>(FRAME) * LB_MAX * 4);
  constexpr unsigned long long FR_BEGAN = 1ull << 63;
  for (uint32_t ch = wave; ch < n_chunks; ch += FR_THREADS / 64)
  {
    const uint32_t i = ch * 64u + lane;
    const FrNodes nd = fr_load_nodes(s_word, s_xyz, i, n, nby, lane, s_cin, ch, false);
    if (nd.live)
    {
      s_xyz[i] = nd.xyz | (lb_oct8(nd.W) << 24);  // the word is final: its octant occupancy rides along for phase D
      nodeA[2 * i] = make_ulonglong2(nd.A[0] - nd.P[0], nd.A[1] - nd.P[1]);
      nodeA[2 * i + 1] = make_ulonglong2(nd.A[2] - nd.P[2], (nd.A[3] - nd.P[3]) | (nd.began_here ? FR_BEGAN : 0ull));
    }
    if (lane == 63)
    {
#pragma unroll
      for (int zz = 0; zz < 4; zz++)
        s_cin[ch + 1][zz] = nd.A[zz];  // staged one slot up: slot k + 1 becomes the carry into chunk k + 1
      s_cflag[ch + 1] = nd.began_here ? 1u : 0u;
    }
  }
  __syncthreads();
  if (wave == 0)
  {
    // carry into chunk k + 1 = S(k) = tail(k) + (the tail row began inside chunk k ? 0 : S(k - 1)): a segmented scan again
    unsigned long long carry[4] = {0ull, 0ull, 0ull, 0ull};
    for (uint32_t k0 = 0; k0 < n_chunks; k0 += 64)
    {
      const uint32_t k = k0 + lane;
      unsigned long long v[4];
      bool hd = true;
      if (k < n_chunks)
      {
#pragma unroll
        for (int zz = 0; zz < 4; zz++)
          v[zz] = s_cin[k + 1][zz];
        hd = s_cflag[k + 1] != 0u;
      }
      else
      {
#pragma unroll
        for (int zz = 0; zz < 4; zz++)
          v[zz] = 0ull;
      }
      fr_segscan(v, hd);
      if (!hd)
      {
#pragma unroll
        for (int zz = 0; zz < 4; zz++)
          v[zz] += carry[zz];
      }
      if (k < n_chunks)
      {
#pragma unroll
        for (int zz = 0; zz < 4; zz++)
          s_cin[k + 1][zz] = v[zz];
      }
#pragma unroll
      for (int zz = 0; zz < 4; zz++)
        carry[zz] = fr_shfl64(v[zz], 63);
    }
    if (lane < 4)
      s_cin[0][lane] = 0ull;
  }
  __syncthreads();
  // Pass b: the last node of every brick row writes the row's channel totals (dense index brick row = bz * nby + by; only
  // occupied rows are written / read)
  for (uint32_t i = tid; i < n; i += FR_THREADS)
  {
    const uint32_t row = fr_row(s_xyz[i], nby);
    const bool tail = i + 1 == n || fr_row(s_xyz[i + 1], nby) != row;
    if (tail)
    {
      const ulonglong2 a0 = nodeA[2 * i], a1 = nodeA[2 * i + 1];
      const unsigned long long W = s_word[i];
      const bool began = (a1.y & FR_BEGAN) != 0ull;
      unsigned long long T[4] = {a0.x + fr_chan(W, 0), a0.y + fr_chan(W, 1), a1.x + fr_chan(W, 2), (a1.y & ~FR_BEGAN) + fr_chan(W, 3)};
      if (!began)
      {
#pragma unroll
        for (int zz = 0; zz < 4; zz++)
          T[zz] += s_cin[i >> 6][zz];
      }
      ulonglong2* dst = reinterpret_cast<ulonglong2*>(rowT + static_cast<size_t>(row) * 4);
      dst[0] = make_ulonglong2(T[0], T[1]);
      dst[1] = make_ulonglong2(T[2], T[3]);
    }
  }
  __syncthreads();
  FR_STAMP(4);
  // Pass c: base rank of every lattice row group in key order.  The groups of slab bz (4 lattice layers zz) come out zz-major,
  // then by: Q(bz, zz, by) = S(bz) + sum_{zz' < zz} P(bz, zz') + sum_{by' < by} t(bz, by', zz), t = voxels of a brick row in layer zz.
  // Round 5: ONE unsegmented block scan over the nodes (7 consecutive nodes per thread; a brick row's last node contributes the
  // row's four layer totals): G_zz(row) = sum of t_zz over the rows before it in node order.  With Gs(bz) = G at the slab's first
  // row:  Q = G_zz(row) + C(bz, zz),  C = sum_zz' Gs_zz'(bz) + sum_{zz' < zz} (Gs_zz'(next slab) - Gs_zz'(bz)) - Gs_zz(bz) - a
  // table of 64 x 4 constants in LDS.  Only occupied rows are touched (rounds 2-4 walked all 4 * nby * nbz row groups, ten per thread,
  // with two look-ups of the parked bitmap in global memory each: 8-16 us).
  // (every prefix is below 65 536 - V is, checked in phase 3b -, so the four layers ride in the 16-bit fields of one 64-bit word)
  {
    constexpr int NPT = LB_MAX / FR_THREADS;
    const uint32_t i0 = static_cast<uint32_t>(tid) * NPT;
    for (int s = tid; s < FR_MAX_NBZ + 1; s += FR_THREADS)
      s_gs[s] = ~0ull;
    uint32_t rows[NPT];  // the brick row a node is the last one of, or ~0
    unsigned long long tz[NPT];
    unsigned long long sum = 0ull;
    uint32_t xyz_next = i0 < n ? s_xyz[i0] : 0u;
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
      const uint32_t i = i0 + r;
      rows[r] = 0xffffffffu;
      if (i < n)
      {
        const uint32_t xyz = xyz_next;
        xyz_next = i + 1 < n ? s_xyz[i + 1] : 0u;
        const uint32_t row = fr_row(xyz, nby);
        if (i + 1 == n || fr_row(xyz_next, nby) != row)
          rows[r] = row;
      }
    }
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
      tz[r] = 0ull;
      if (rows[r] != 0xffffffffu)
      {
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(rowT + static_cast<size_t>(rows[r]) * 4);
        const ulonglong2 t0 = src[0], t1 = src[1];
        tz[r] = static_cast<unsigned long long>(fr_hsum16(t0.x)) | (static_cast<unsigned long long>(fr_hsum16(t0.y)) << 16) | (static_cast<unsigned long long>(fr_hsum16(t1.x)) << 32) |
                (static_cast<unsigned long long>(fr_hsum16(t1.y)) << 48);
        sum += tz[r];
      }
    }
    unsigned long long incl = sum;
    incl += dpp_mov0_64<0x111, 0xf>(incl);
    incl += dpp_mov0_64<0x112, 0xf>(incl);
    incl += dpp_mov0_64<0x114, 0xf>(incl);
    incl += dpp_mov0_64<0x118, 0xf>(incl);
    incl += dpp_mov0_64<0x142, 0xa>(incl);
    incl += dpp_mov0_64<0x143, 0xc>(incl);
    if (lane == 63)
      s_w4[wave] = incl;
    __syncthreads();
    unsigned long long G = incl - sum, tot = 0ull;
    for (int w = 0; w < FR_THREADS / 64; w++)
    {
      const unsigned long long x = s_w4[w];
      G += w < wave ? x : 0ull;
      tot += x;
    }
    if (tid == 0)
      s_gs[FR_MAX_NBZ] = tot;  // the sentinel behind the last slab
    uint32_t bz_prev = (i0 > 0 && i0 < n) ? static_cast<uint32_t>(fr_bz(s_xyz[i0 - 1])) : 0xffffffffu;
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
      const uint32_t i = i0 + r;
      if (i >= n)
        break;
      const uint32_t bz = static_cast<uint32_t>(fr_bz(s_xyz[i]));
      if (bz != bz_prev)  // the slab's first node: every row of the slabs before has been added
        s_gs[bz] = G;
      bz_prev = bz;
      if (rows[r] != 0xffffffffu)
      {
        // G of the row: the slab's constant is added where it is read
        *reinterpret_cast<uint4*>(rowQ + static_cast<size_t>(rows[r]) * 4) =
            make_uint4(static_cast<uint32_t>(G) & 0xffffu, static_cast<uint32_t>(G >> 16) & 0xffffu, static_cast<uint32_t>(G >> 32) & 0xffffu, static_cast<uint32_t>(G >> 48));
        G += tz[r];
      }
    }
    __syncthreads();
    if (tid < nbz && s_gs[tid] != ~0ull)
    {
      int nxt = tid + 1;
      while (nxt < nbz && s_gs[nxt] == ~0ull)
        nxt++;
      if (nxt >= nbz)
        nxt = FR_MAX_NBZ;
      const unsigned long long ga = s_gs[tid], gb = s_gs[nxt];
      const uint32_t g0 = static_cast<uint32_t>(ga) & 0xffffu, g1 = static_cast<uint32_t>(ga >> 16) & 0xffffu, g2 = static_cast<uint32_t>(ga >> 32) & 0xffffu, g3 = static_cast<uint32_t>(ga >> 48);
      const uint32_t p0 = (static_cast<uint32_t>(gb) & 0xffffu) - g0, p1 = (static_cast<uint32_t>(gb >> 16) & 0xffffu) - g1, p2 = (static_cast<uint32_t>(gb >> 32) & 0xffffu) - g2;
      const uint32_t S = g0 + g1 + g2 + g3;
      *reinterpret_cast<uint4*>(&s_qc[tid][0]) = make_uint4(S - g0, S + p0 - g1, S + p0 + p1 - g2, S + p0 + p1 + p2 - g3);
    }
  }
  __syncthreads();
  FR_STAMP(5);
  // (the rank of a voxel, as pass d below computes it - for the close-first path, which comes back to its few far bricks later)
  auto rank_ctx = [&](uint32_t i, uint32_t xyz, unsigned long long (&M)[4], uint32_t (&Q)[4]) {
    const uint32_t row = fr_row(xyz, nby);
    const ulonglong2 a0 = nodeA[2 * i], a1 = nodeA[2 * i + 1];
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(rowT + static_cast<size_t>(row) * 4);
    const ulonglong2 t0 = src[0], t1 = src[1];
    const uint4 qq = *reinterpret_cast<const uint4*>(rowQ + static_cast<size_t>(row) * 4);
    M[0] = fr_excl16(t0.x) + a0.x;
    M[1] = fr_excl16(t0.y) + a0.y;
    M[2] = fr_excl16(t1.x) + a1.x;
    M[3] = fr_excl16(t1.y) + (a1.y & ~FR_BEGAN);
    if (!(a1.y & FR_BEGAN))
    {
#pragma unroll
      for (int zz = 0; zz < 4; zz++)
        M[zz] += s_cin[i >> 6][zz];
    }
    const uint4 qc = *reinterpret_cast<const uint4*>(&s_qc[fr_bz(xyz)][0]);
    Q[0] = qq.x + qc.x, Q[1] = qq.y + qc.y, Q[2] = qq.z + qc.z, Q[3] = qq.w + qc.w;
  };
  auto rank_of = [&](unsigned long long W, int p, const unsigned long long (&M)[4], const uint32_t (&Q)[4]) -> uint32_t {
    const int zz = p >> 4, yy = (p >> 2) & 3, xx = p & 3;
    const unsigned long long Mz = zz == 0 ? M[0] : zz == 1 ? M[1] : zz == 2 ? M[2] : M[3];
    const uint32_t Qz = zz == 0 ? Q[0] : zz == 1 ? Q[1] : zz == 2 ? Q[2] : Q[3];
    const uint32_t nib = static_cast<uint32_t>(W >> (p & ~3)) & 0xfu;
    return Qz + (static_cast<uint32_t>(Mz >> (16 * yy)) & 0xffffu) + __popc(nib & ((1u << xx) - 1u));
  };
  // Pass d: the voxel records leave at their ranks (voxel_grid_weighted.cpp:171-188): centre, weight 1 (+ extras below),
  // node of the brick (for the label pass); the lattice key only for the general kernels that may follow (!write_tables).
  for (uint32_t i = tid; i < n; i += FR_THREADS)
  {
    const uint32_t xyz = s_xyz[i];
    const uint32_t row = fr_row(xyz, nby);
    const unsigned long long W = s_word[i];
    unsigned long long M[4];
    uint32_t Q[4];
    {
      const ulonglong2 a0 = nodeA[2 * i], a1 = nodeA[2 * i + 1];
      const ulonglong2* src = reinterpret_cast<const ulonglong2*>(rowT + static_cast<size_t>(row) * 4);
      const ulonglong2 t0 = src[0], t1 = src[1];
      const uint4 qq = *reinterpret_cast<const uint4*>(rowQ + static_cast<size_t>(row) * 4);
      const bool began = (a1.y & FR_BEGAN) != 0ull;
      M[0] = fr_excl16(t0.x) + a0.x;
      M[1] = fr_excl16(t0.y) + a0.y;
      M[2] = fr_excl16(t1.x) + a1.x;
      M[3] = fr_excl16(t1.y) + (a1.y & ~FR_BEGAN);
      if (!began)
      {
#pragma unroll
        for (int zz = 0; zz < 4; zz++)
          M[zz] += s_cin[i >> 6][zz];
      }
      const uint4 qc = *reinterpret_cast<const uint4*>(&s_qc[fr_bz(xyz)][0]);
      Q[0] = qq.x + qc.x, Q[1] = qq.y + qc.y, Q[2] = qq.z + qc.z, Q[3] = qq.w + qc.w;
    }
    const int bx = fr_bx(xyz), by = fr_by(xyz), bz = fr_bz(xyz);
    unsigned long long w = W;
    bool first = true;
    const uint8_t* cntp = reinterpret_cast<const uint8_t*>(s_bb) + s_vbase[i];  // the node's voxels in bit order
    while (w)
    {
      const int p = __ffsll(static_cast<long long>(w)) - 1;
      w &= w - 1;
      const int zz = p >> 4, yy = (p >> 2) & 3, xx = p & 3;
      const unsigned long long Mz = zz == 0 ? M[0] : zz == 1 ? M[1] : zz == 2 ? M[2] : M[3];
      const uint32_t Qz = zz == 0 ? Q[0] : zz == 1 ? Q[1] : zz == 2 ? Q[2] : Q[3];
      const uint32_t nib = static_cast<uint32_t>(W >> (p & ~3)) & 0xfu;
      const uint32_t rank = Qz + (static_cast<uint32_t>(Mz >> (16 * yy)) & 0xffffu) + __popc(nib & ((1u << xx) - 1u));
      const int k0 = 4 * bx + xx + o0, k1 = 4 * by + yy + o1, k2 = 4 * bz + zz + o2;  // the frame's own cell
      float4 pt;
      pt.x = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k0), 0.5f), g.leaf[0]), hoff0);
      pt.y = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k1), 0.5f), g.leaf[1]), hoff1);
      pt.z = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(k2), 0.5f), g.leaf[2]), hoff2);
      pt.w = __uint_as_float(static_cast<uint32_t>(*cntp++));  // points in the voxel (the record list below adds what a byte cannot hold)
      va.pts[rank] = pt;  // (through L2 on purpose: it merges the scattered 16-byte records into lines; non-temporal stores cost 21 % of the throughput)
      if (!cf)  // (the label pass of the full clustering finds a voxel's brick here; the close-first path has no such pass)
        reinterpret_cast<uint16_t*>(va.bb)[rank] = static_cast<uint16_t>(i);  // (node < LB_MAX: 16 bits; the general kernels keep 32-bit brick codes here)
      if (!write_tables)
        va.key[rank] = static_cast<uint32_t>(k0 + k1 * dx + k2 * dxy);
      if (first && !cf)
        bmin_g[i] = rank;  // bit order inside a brick is the key order: the lowest bit is the brick's first voxel
      first = false;
    }
  }
  __syncthreads();
  FR_STAMP(6);
  // the records of byte counters that overflowed (none on LiDAR scans) add to their voxels' weights: rank = row base + channel
  // prefix of the earlier bricks of the row + bits below.  Four records per lane and round.
  {
    const uint32_t ne = s_ne;
    constexpr int EU = 4;
    for (uint32_t e0 = tid; e0 < ne; e0 += FR_THREADS * EU)
    {
      uint32_t rec[EU], node[EU], r[EU];
#pragma unroll
      for (int u = 0; u < EU; u++)
      {
        const uint32_t e = e0 + u * FR_THREADS;
        rec[u] = e < ne ? extras_g[e] : FR_CODE_NONE;
      }
      unsigned long long az[EU], a3[EU], tz[EU], wv[EU];
      uint32_t qz[EU];
#pragma unroll
      for (int u = 0; u < EU; u++)
      {
        const bool ok = rec[u] != FR_CODE_NONE;
        const uint32_t zz = (rec[u] >> 4) & 3u;
        node[u] = ok ? (rec[u] >> 6) & 0x7ffffu : 0u;
        r[u] = fr_row(s_xyz[node[u]], nby);
        const unsigned long long* na = fs.nodeA + (static_cast<size_t>(FRAME) * LB_MAX + node[u]) * 4;
        az[u] = na[zz];
        a3[u] = na[3];
        tz[u] = ok ? rowT[static_cast<size_t>(r[u]) * 4 + zz] : 0ull;
        qz[u] = ok ? rowQ[static_cast<size_t>(r[u]) * 4 + zz] + s_qc[fr_bz(s_xyz[node[u]])][zz] : 0u;
        wv[u] = s_word[node[u]];
      }
#pragma unroll
      for (int u = 0; u < EU; u++)
      {
        if (rec[u] == FR_CODE_NONE)
          continue;
        const uint32_t add = (rec[u] >> 25) + 1u, p = rec[u] & 63u;
        const uint32_t zz = p >> 4, yy = (p >> 2) & 3u, xx = p & 3u;
        unsigned long long Mz = fr_excl16(tz[u]) + (az[u] & ~FR_BEGAN);
        if (!(a3[u] & FR_BEGAN))
          Mz += s_cin[node[u] >> 6][zz];
        const uint32_t nib = static_cast<uint32_t>(wv[u] >> (p & ~3u)) & 0xfu;
        const uint32_t rank = qz[u] + (static_cast<uint32_t>(Mz >> (16 * yy)) & 0xffffu) + __popc(nib & ((1u << xx) - 1u));
        atomicAdd(reinterpret_cast<uint32_t*>(&va.pts[rank].w), add);
      }
    }
  }
  __syncthreads();  // counters and voxel bases are dead: the bitmap returns, the bases' storage becomes the union-find
  {
    const ulonglong2* bsave = reinterpret_cast<const ulonglong2*>(fs.bbsave + static_cast<size_t>(FRAME) * FR_BB64);
    for (int i = tid; i < FR_BB64 / 2; i += FR_THREADS)
      reinterpret_cast<ulonglong2*>(s_bb)[i] = bsave[i];
  }
  uint16_t* s_pfidx = reinterpret_cast<uint16_t*>(s_x2);  // close-first: node -> index in the pure-far list, 0xffff: the brick holds a close voxel
  for (uint32_t i = tid; i < n; i += FR_THREADS)
    s_par[i] = cf ? static_cast<uint16_t>(0xffffu) : static_cast<uint16_t>(i);  // (s_pfidx and s_par are the same storage)
  if (tid == 0)
    h.V = V;
  __syncthreads();
  if constexpr (cf)
  {
    for (uint32_t k = tid; k < s_npf; k += FR_THREADS)
    {
      s_pfidx[s_pf[k]] = static_cast<uint16_t>(k);
      s_pfpar[k] = static_cast<uint16_t>(k);
    }
    __syncthreads();
  }
  FR_STAMP(7);
  constexpr int VU = 8;  // voxel records fetched per lane and round in the label pass
  const uint32_t Vround = (V + 63u) & ~63u;
  // ---- D: probe, test, union (the bitmap's prefix is per 64-bit word).
  // D-a: every (brick, stencil row) reads one window of the brick bitmap; occupied neighbours go to the hit list.
  // D-b: adjacent-brick hits first (octant matrices, unions), flatten, then the hits two bricks away: most of them now join
  //      bricks of one component and are dismissed by two LDS reads.  Pairs the octant matrices leave open are collected.
  // D-c: the open pairs that still join different components are compacted, then tested exactly, one per lane.
  uint32_t* hits = scratch_all + static_cast<size_t>(FRAME) * g.vox_cap * 10u;
  const uint32_t hcap = g.vox_cap * 5u;
  uint32_t* opens = hits + hcap;
  // Adjacent bricks first: face neighbours (<= 3 per brick in the half stencil) and the other adjacent bricks (<= 10) go to
  // two hit lists.  Bricks two apart are looked at after these have been merged - and then only around the bricks outside
  // the largest component (D-a2 below): a pair inside one component has nothing left to decide.
  const uint32_t cap_axis = 3u * n, cap_near = 10u * n;
  if (!cf && hcap < 14u * n)
  {
    if (tid == 0)
    {
      h.status = CCL_RETRY_STATUS;
      h.V = 0;
    }
    return;
  }
  const uint32_t cap_far = hcap - cap_axis - cap_near;
  uint32_t* hits_near = hits + cap_axis;
  uint32_t* hits_far = hits_near + cap_near;
  const int R = s_tab.R, n_rows = s_tab.n_rows;
  constexpr int RPL = LB_MAX_ROWS / LB_LANES;  // rows per lane
  static_assert(RPL == 2, "the reservation below adds up two rows per lane");
  // per stencil row: offsets, and per window slot s (dx = s - R): in the half stencil / face neighbour / adjacent
  auto load_row = [&](int row, int& ddy, int& ddz, uint32_t& valid, uint32_t& axis, uint32_t& near, unsigned long long& ov) {
    ddy = ddz = 0;
    valid = axis = near = 0;
    ov = 0;
    if (row >= n_rows)
      return;
    const unsigned long long near_mask = s_tab.near_mask, axis_mask = s_tab.axis_mask;
    const unsigned long long q0 = reinterpret_cast<const unsigned long long*>(&s_tab.rows[row])[0];
    const unsigned long long q1 = reinterpret_cast<const unsigned long long*>(&s_tab.rows[row])[1];
    ddy = static_cast<int8_t>(q0 & 0xffu);
    ddz = static_cast<int8_t>((q0 >> 8) & 0xffu);
    valid = static_cast<uint32_t>(q0 >> 16) & 0xffu;
    ov = (q0 >> 24) | (q1 << 40);  // byte s: stencil index of dx = s - R
    for (int sl = 0; sl < LB_WIN; sl++)
      if ((valid >> sl) & 1u)
      {
        const uint32_t o = static_cast<uint32_t>(ov >> (8 * sl)) & 0xffu;
        axis |= static_cast<uint32_t>((axis_mask >> o) & 1ull) << sl;
        near |= static_cast<uint32_t>((near_mask >> o) & 1ull) << sl;
      }
  };
  // the bricks bx - R .. bx + R of lattice row (ny, nz) as a window: bit s = the brick at dx = s - R is occupied;
  // nb0 / raw / shw recover the node of a set bit: nb0 + popc(raw & ((1 << (s - shw)) - 1))
  auto window = [&](int bx, int ny, int nz, uint32_t& raw, uint32_t& nb0, int& shw) -> uint32_t {
    const int lo = max(bx - R, 0), hi = min(bx + R, nbx - 1);
    const uint32_t firstb = static_cast<uint32_t>((nz * nby + ny) * nbx) + lo;
    const uint32_t wi = firstb >> 6, sh = firstb & 63u;
    const unsigned long long w0 = s_bits64[wi], w1 = s_bits64[wi + 1];
    const uint32_t pre = s_pre[wi];
    const unsigned long long two = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
    raw = static_cast<uint32_t>(two) & ((1u << (hi - lo + 1)) - 1u);  // bit j: brick firstb + j
    shw = lo - (bx - R);
    nb0 = pre + __popcll(w0 & ((1ull << sh) - 1ull));  // node of the first occupied brick at or after `firstb`
    return raw << shw;
  };
  auto is_cand = [&](uint32_t close, uint32_t size, const int* box) {
    int ext_ok = 1;
    for (int a = 0; a < 3; a++)
      ext_ok &= (static_cast<float>(box[3 + a] - box[a]) * g.leaf[a] <= up.cand_max_extent);
    return !close && static_cast<int>(size) >= up.min_points && ext_ok;
  };
  if constexpr (cf)
  {
    // ---- close first, continued: edges and unions around the pure-far bricks (list entry k <-> thread k from F-b on).
    // Work item = (pure-far brick, row of the half stencil, direction), as phase D-a2 of the full clustering: a thread's items
    // share row and direction.  A neighbour brick that holds a close voxel TAINTS the entry; a pure-far neighbour is joined.
    //   F-a1  adjacent bricks (|d| <= 1 brick), every entry, both directions for taints, forward only for unions (the pair is
    //         seen from its base brick): nearly every pure-far brick next to the ground sheet or a wall is tainted here;
    //   F-a2  bricks two apart, around the entries still untainted, both directions for both kinds (the partner may be tainted
    //         and sit this pass out; a pair of two tainted entries has nothing left to decide);
    //   F-a3  the pairs the octant matrices left open, if they can still change anything: the exact ball test (FLANN's float
    //         expression on the tolerance's boundary).
    const uint32_t n_pf = s_npf;
    auto tainted = [&](uint32_t k) -> bool { return ((*reinterpret_cast<volatile uint32_t*>(&s_taint[k >> 5]) >> (k & 31u)) & 1u) != 0u; };
    auto join = [&](uint32_t ka, uint32_t kb) {
      uint32_t ra = lb_find(s_pfpar, ka), rb = lb_find(s_pfpar, kb);
      while (ra != rb)  // hook the larger root under the smaller
      {
        if (ra < rb)
        {
          const uint32_t tmp = ra;
          ra = rb;
          rb = tmp;
        }
        const uint32_t old = lb_cas16(s_pfpar, ra, ra, rb);
        if (old == ra)
          break;
        ra = old;
      }
    };
    constexpr uint32_t CF_OPEN_UNION = 1u << 23;  // open pair: k | t2 << 10 | kind
    // (row, direction) combinations and their slots: 32 descriptors in LDS, and the lists of those that have adjacent / far slots
    // at all - the items of a pass are dealt over ALL threads (a thread per (entry, combination); with a fixed combination per
    // thread, as in D-a2, a third of the threads carried the adjacent pass: 11-19 us of LDS latency chains)
    if (tid < 32)
    {
      const int row = tid >> 1, back = tid & 1;
      int ddy, ddz;
      uint32_t valid, axis, near;
      unsigned long long ov;
      load_row(row, ddy, ddz, valid, axis, near, ov);
      uint32_t slots_near = valid & near, slots_far = valid & ~near;
      if (back)
      {
        uint32_t rn = 0, rf = 0;  // the neighbour is the pair's base brick: it sees this brick at (-dx, ddy, ddz), i.e. slot 2R - s
        for (int sl = 0; sl <= 2 * R; sl++)
        {
          rn |= ((slots_near >> (2 * R - sl)) & 1u) << sl;
          rf |= ((slots_far >> (2 * R - sl)) & 1u) << sl;
        }
        slots_near = rn;
        slots_far = rf;
        ddy = -ddy;
        ddz = -ddz;
      }
      s_cfd[tid][0] = (static_cast<uint32_t>(ddy) & 0xffu) | ((static_cast<uint32_t>(ddz) & 0xffu) << 8) | (slots_near << 16) | (slots_far << 24);
      s_cfd[tid][1] = static_cast<uint32_t>(back);
      s_cfd[tid][2] = static_cast<uint32_t>(ov);
      s_cfd[tid][3] = static_cast<uint32_t>(ov >> 32);
      const unsigned long long mn = __ballot(slots_near != 0u), mf = __ballot(slots_far != 0u);
      if (slots_near)
        s_cfl[0][__popcll(mn & ((1ull << tid) - 1ull))] = static_cast<uint8_t>(tid);
      if (slots_far)
        s_cfl[1][__popcll(mf & ((1ull << tid) - 1ull))] = static_cast<uint8_t>(tid);
      if (tid == 0)
      {
        s_cfn[0] = static_cast<uint32_t>(__popcll(mn));
        s_cfn[1] = static_cast<uint32_t>(__popcll(mf));
      }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; pass++)
    {
      const uint32_t ncmb = s_cfn[pass];
      const uint32_t items = n_pf * ncmb;
      const uint32_t inv = ncmb > 1u ? 0xffffffffu / ncmb + 1u : 0u;  // it / ncmb == umulhi(it, ceil(2^32 / ncmb)) for it < 2^32 / ncmb
      for (uint32_t it = tid; it < items; it += FR_THREADS)
      {
        const uint32_t k = ncmb > 1u ? __umulhi(it, inv) : it;
        if (pass == 1 && tainted(k))
          continue;
        const uint4 dsc = *reinterpret_cast<const uint4*>(&s_cfd[s_cfl[pass][it - k * ncmb]][0]);
        const int ddy = static_cast<int8_t>(dsc.x & 0xffu), ddz = static_cast<int8_t>((dsc.x >> 8) & 0xffu);
        const uint32_t slots = pass == 0 ? (dsc.x >> 16) & 0xffu : dsc.x >> 24;
        const bool back = dsc.y != 0u;
        const unsigned long long ov = static_cast<unsigned long long>(dsc.z) | (static_cast<unsigned long long>(dsc.w) << 32);
        const uint32_t t = s_pf[k];
        const uint32_t xa = s_xyz[t];
        const int bx = fr_bx(xa), by = fr_by(xa), bz = fr_bz(xa);
        const int ny = by + ddy, nz = bz + ddz;  // (the descriptor of a backward combination carries the negated offsets)
        if (ny < 0 || ny >= nby || nz < 0 || nz >= nbz)
          continue;
        uint32_t raw, nb0;
        int shw;
        uint32_t win = window(bx, ny, nz, raw, nb0, shw) & slots;
        while (win)
        {
          const int sl = __ffs(static_cast<int>(win)) - 1;
          win &= win - 1;
          const uint32_t t2 = nb0 + __popc(raw & ((1u << (sl - shw)) - 1u));
          const uint32_t k2 = s_pfidx[t2];
          const uint32_t xb = s_xyz[t2];
          const bool other_close = k2 == 0xffffu;
          if (!other_close && ((pass == 0 && back) || lb_ld16(s_pfpar, k) == lb_ld16(s_pfpar, k2)))
            continue;
          const uint32_t o = static_cast<uint32_t>(ov >> (8 * (back ? 2 * R - sl : sl))) & 0xffu;
          const uint32_t A8 = (back ? xb : xa) >> 24, B8 = (back ? xa : xb) >> 24;  // (the matrices are indexed base brick x brick at the offset)
          if (lb_octtest(s_tab.oct[2 * o], A8, B8))
          {
            if (other_close)
              atomicOr(&s_taint[k >> 5], 1u << (k & 31u));
            else
              join(k, k2);
          }
          else if (lb_octtest(s_tab.oct[2 * o + 1], A8, B8))
          {
            const uint32_t pos = atomicAdd(&s_no, 1u);
            if (pos < hcap)
              hits[pos] = k | (t2 << 10) | (other_close ? 0u : CF_OPEN_UNION);
          }
        }
      }
      __syncthreads();
    }
    FR_STAMP(8);
    if (s_no > hcap)
    {
      // more open pairs than the scratch list holds (a tiny workspace: the list has five words per voxel slot): nothing may
      // be dropped - the batch takes the full clustering
      if (tid == 0)
      {
        h.status = CF_RETRY_STATUS;
        h.V = 0;
      }
      return;
    }
    {
      const uint32_t no = s_no;
      for (uint32_t i = tid; i < no; i += FR_THREADS)
      {
        const uint32_t e = hits[i];
        const uint32_t k = e & 1023u, t2 = (e >> 10) & 8191u;
        const uint32_t k2 = s_pfidx[t2];
        if (e & CF_OPEN_UNION)
        {
          if ((tainted(k) && tainted(k2)) || lb_find(s_pfpar, k) == lb_find(s_pfpar, k2))
            continue;
        }
        else if (tainted(k))
          continue;
        const uint32_t t = s_pf[k];
        const uint32_t xa = s_xyz[t], xb = s_xyz[t2];
        const int bx = fr_bx(xa), by = fr_by(xa), bz = fr_bz(xa);
        if (!lb_pair_conn(s_tab, g, bp, h, s_word[t], s_word[t2], 4 * bx + o0, 4 * by + o1, 4 * bz + o2, fr_bx(xb) - bx, fr_by(xb) - by, fr_bz(xb) - bz))
          continue;
        if (e & CF_OPEN_UNION)
          join(k, k2);
        else
          atomicOr(&s_taint[k >> 5], 1u << (k & 31u));
      }
    }
    __syncthreads();  // (the bitmap is dead from here on: its storage holds the components' accumulators, indexed by the root's list index)
    FR_STAMP(9);
    uint32_t* a_size = reinterpret_cast<uint32_t*>(s_bb);
    uint32_t* a_min = a_size + CF_MAX;
    int* a_box = reinterpret_cast<int*>(a_min + CF_MAX);
    uint8_t* a_cand = reinterpret_cast<uint8_t*>(a_box + 6 * CF_MAX);
    static_assert(static_cast<size_t>(CF_MAX) * 33u <= sizeof(unsigned long long) * FR_BB64, "the accumulators of the close-first path live in the bitmap's storage");
    static_assert(CF_MAX == FR_THREADS, "one pure-far brick per thread");
    const uint32_t k = tid;
    const bool live = k < n_pf;
    uint32_t root = k;
    if (live)
    {
      uint32_t p;
      while ((p = lb_ld16(s_pfpar, root)) != root)
        root = p;
      a_size[k] = 0u;
      a_min[k] = 0xffffffffu;
      for (int a = 0; a < 3; a++)
      {
        a_box[6 * k + a] = 0x7fffffff;
        a_box[6 * k + 3 + a] = static_cast<int>(0x80000000u);
      }
      a_cand[k] = 0;
    }
    __syncthreads();
    // F-b: sizes, lattice boxes, smallest members (= labels: rank of the first voxel of one of the component's bricks), taint
    unsigned long long W = 0ull, M[4] = {0ull, 0ull, 0ull, 0ull};
    uint32_t Q[4] = {0u, 0u, 0u, 0u};
    if (live)
    {
      const uint32_t node = s_pf[k];
      const uint32_t xyz = s_xyz[node];
      W = s_word[node];
      rank_ctx(node, xyz, M, Q);
      const uint32_t first = rank_of(W, __ffsll(static_cast<long long>(W)) - 1, M, Q);
      const int bx = fr_bx(xyz), by = fr_by(xyz), bz = fr_bz(xyz);
      unsigned long long t = W | (W >> 16) | (W >> 32) | (W >> 48);
      uint32_t ox = static_cast<uint32_t>(t) & 0xffffu;
      ox = (ox | (ox >> 4) | (ox >> 8) | (ox >> 12)) & 0xfu;
      t = W | (W >> 1);
      t |= t >> 2;  // bit 4y + 16z: row (y,z) is occupied
      const unsigned long long ty = t | (t >> 16) | (t >> 32) | (t >> 48);
      const uint32_t oy = (static_cast<uint32_t>(ty) & 1u) | ((static_cast<uint32_t>(ty) >> 3) & 2u) | ((static_cast<uint32_t>(ty) >> 6) & 4u) | ((static_cast<uint32_t>(ty) >> 9) & 8u);
      const uint32_t oz = ((W & 0xffffull) ? 1u : 0u) | ((W & 0xffff0000ull) ? 2u : 0u) | ((W & 0xffff00000000ull) ? 4u : 0u) | ((W >> 48) ? 8u : 0u);
      const int lo[3] = {4 * bx + o0 + __ffs(static_cast<int>(ox)) - 1, 4 * by + o1 + __ffs(static_cast<int>(oy)) - 1, 4 * bz + o2 + __ffs(static_cast<int>(oz)) - 1};
      const int hi[3] = {4 * bx + o0 + 31 - __clz(static_cast<int>(ox)), 4 * by + o1 + 31 - __clz(static_cast<int>(oy)), 4 * bz + o2 + 31 - __clz(static_cast<int>(oz))};
      atomicAdd(&a_size[root], static_cast<uint32_t>(__popcll(W)));
      atomicMin(&a_min[root], first);
      for (int a = 0; a < 3; a++)
      {
        atomicMin(&a_box[6 * root + a], lo[a]);
        atomicMax(&a_box[6 * root + 3 + a], hi[a]);
      }
      if (tainted(k))
        atomicOr(&s_troot[root >> 5], 1u << (root & 31u));
    }
    __syncthreads();
    FR_STAMP(10);
    // F-c: the surviving components ARE far_clusters_indices of vofod_nodelet.cpp:746: their records (close = 0) make the frame's
    // cluster table, the voxels of the candidates among them (enough points, small enough to pass max_size) the member list.
    // Both leave in the order the classification tail wants them (k_tail_far reads them without sorting, without LDS): the
    // candidates first, in the canonical cluster order (size descending, smallest member ascending: SURVEY H3), their members
    // cluster by cluster with ascending rank (the order PCL's moment sums run in).  Ordering is by counting: a frame has a
    // handful of candidates with tens of voxels.  Beyond TAIL_MAXC candidates / TAIL_MAXM members (the tail's capacities: it
    // hands such a batch to the host) the lists leave unordered.
    ClusterRec* table = table_all + static_cast<size_t>(FRAME) * g.vox_cap;
    CandMember* cands = cand_all + static_cast<size_t>(FRAME) * g.vox_cap;
    uint16_t* a_ord = reinterpret_cast<uint16_t*>(a_cand + CF_MAX);  // per root: its row of the table
    uint16_t* c_root = a_ord + CF_MAX;                               // candidate roots in arrival order
    uint16_t* c_byord = c_root + TAIL_MAXC;                          // ... by canonical order
    uint32_t* st_key = reinterpret_cast<uint32_t*>(c_byord + TAIL_MAXC);  // members: canonical order of the cluster << 16 | rank
    static_assert(static_cast<size_t>(CF_MAX) * 35u + 4u * TAIL_MAXC + 4u * TAIL_MAXM <= sizeof(unsigned long long) * FR_BB64, "staging of the close-first path lives in the bitmap's storage");
    const bool surv = live && !((s_troot[root >> 5] >> (root & 31u)) & 1u);
    bool my_cand = false;
    uint32_t my_slot = 0;
    if (surv && root == k)
    {
      my_cand = is_cand(0u, a_size[k], &a_box[6 * k]);
      a_cand[k] = my_cand ? 1 : 0;
      my_slot = atomicAdd(my_cand ? &s_nc : &s_no2, 1u);
      if (my_cand && my_slot < static_cast<uint32_t>(TAIL_MAXC))
        c_root[my_slot] = static_cast<uint16_t>(k);
    }
    if (close_first == 2)  // the far-only debug view: a label for every voxel
      for (uint32_t v = tid; v < V; v += FR_THREADS)
        labels[v] = CF_LABEL_NONE;
    __syncthreads();
    FR_STAMP(11);
    const uint32_t n_cc = s_nc;  // candidate clusters
    const bool ordered_c = n_cc <= static_cast<uint32_t>(TAIL_MAXC);
    if (surv && root == k)
    {
      uint32_t row = my_cand ? my_slot : n_cc + my_slot;
      if (my_cand && ordered_c)
      {
        row = 0;
        const uint32_t sz = a_size[k], mn = a_min[k];
        for (uint32_t j = 0; j < n_cc; j++)
        {
          const uint32_t kj = c_root[j];
          const uint32_t sj = a_size[kj], mj = a_min[kj];
          row += (sj > sz || (sj == sz && mj < mn)) ? 1u : 0u;
        }
        c_byord[row] = static_cast<uint16_t>(k);
      }
      a_ord[k] = static_cast<uint16_t>(min(row, 0xffffu));
      ClusterRec rec;
      rec.root = a_min[k];
      rec.size = a_size[k];
      for (int a = 0; a < 3; a++)
      {
        rec.imin[a] = a_box[6 * k + a];
        rec.imax[a] = a_box[6 * k + 3 + a];
      }
      rec.close = 0u;
      rec.cand = my_cand ? 1u : 0u;
      table[row] = rec;
    }
    // the candidates' voxels: space first ...
    const bool writes = surv && (a_cand[root] || close_first == 2);
    const bool cand_brick = surv && a_cand[root] != 0;
    uint32_t m_base = 0;
    if (cand_brick)
      m_base = atomicAdd(&s_ncand, static_cast<uint32_t>(__popcll(W)));
    __syncthreads();
    const uint32_t n_mem = s_ncand;
    const bool ordered_m = ordered_c && n_mem <= static_cast<uint32_t>(TAIL_MAXM);
    // ... then the records (unordered lists: straight to their places; ordered: staged as keys first)
    if (writes)
    {
      const uint32_t label = a_min[root];
      const uint32_t ord = a_ord[root];
      uint32_t pos = m_base;
      unsigned long long w = W;
      while (w)
      {
        const int p = __ffsll(static_cast<long long>(w)) - 1;
        w &= w - 1;
        const uint32_t rank = rank_of(W, p, M, Q);
        if (cand_brick)
        {
          if (ordered_m)
            st_key[pos++] = (ord << 16) | rank;  // (rank < V <= 65535: checked in phase 3b)
          else
          {
            CandMember cm;
            cm.root = label;
            cm.v = rank;
            cands[pos++] = cm;
          }
        }
        if (close_first == 2)
          labels[rank] = label;
      }
    }
    if (ordered_m)
    {
      __syncthreads();
      for (uint32_t i = tid; i < n_mem; i += FR_THREADS)
      {
        const uint32_t key = st_key[i];
        uint32_t at = 0;
        for (uint32_t j = 0; j < n_mem; j++)
          at += st_key[j] < key ? 1u : 0u;  // (keys are distinct: a voxel has one rank)
        CandMember cm;
        cm.root = a_min[c_byord[key >> 16]];
        cm.v = key & 0xffffu;
        cands[at] = cm;
      }
    }
    __syncthreads();
    if (tid == 0)
    {
      h.C = n_cc + s_no2;
      h.n_cand = n_mem;
      h.n_bricks = n;
      h.far_only = 1u;
      h.n_cand_clusters = n_cc;
    }
    FR_STAMP(12);
    FR_STAMP(13);
    if (prof && tid == 0)
    {
      prof[static_cast<size_t>(FRAME) * 32 + 24] = n_pf;
      prof[static_cast<size_t>(FRAME) * 32 + 25] = s_nc + s_no2;
      prof[static_cast<size_t>(FRAME) * 32 + 26] = n;
      prof[static_cast<size_t>(FRAME) * 32 + 27] = n_keys;
      prof[static_cast<size_t>(FRAME) * 32 + 28] = s_ne;
      prof[static_cast<size_t>(FRAME) * 32 + 29] = V;
      prof[static_cast<size_t>(FRAME) * 32 + 30] = s_ncand;
      prof[static_cast<size_t>(FRAME) * 32 + 31] = ~0ull;  // (marks a close-first frame for print_prof)
    }
    return;
  }
  {
    // the adjacent bricks sit in at most five stencil rows ((dy, dz) in the half stencil with |dy|, |dz| <= 1): work item =
    // (brick, one of these rows), P consecutive lanes per brick (round 2 gave every brick eight lanes, three of them idle);
    // the rows' descriptors wait in LDS
    if (tid == 0)
    {
      int k = 0;
      for (int row = 0; row < n_rows; row++)
      {
        int y, z;
        uint32_t v, ax, nr;
        unsigned long long o;
        load_row(row, y, z, v, ax, nr, o);
        if (!(v & nr) || k >= 5)
          continue;
        s_near[k][0] = (static_cast<uint32_t>(y) & 0xffu) | ((static_cast<uint32_t>(z) & 0xffu) << 8) | (((v & nr) & 0xffu) << 16) | ((ax & 0xffu) << 24);
        s_near[k][1] = 0u;
        s_near[k][2] = static_cast<uint32_t>(o);
        s_near[k][3] = static_cast<uint32_t>(o >> 32);
        k++;
      }
      s_near_n = static_cast<uint32_t>(k);
    }
    __syncthreads();
    const uint32_t P = max(s_near_n, 1u);
    const uint32_t Pm = P > 1u ? 0xffffffffu / P + 1u : 0u;  // idx / P == umulhi(idx, ceil(2^32 / P)) for idx < 2^32 / P
    const uint32_t items = n * P;
    const uint32_t n_round = (items + FR_THREADS - 1) / FR_THREADS * FR_THREADS;
    for (uint32_t idx = tid; idx < n_round; idx += FR_THREADS)  // wave-uniform trip counts: the reservations below scan the wave
    {
      const uint32_t t = P > 1u ? __umulhi(idx, Pm) : idx;
      const uint32_t sub = idx - t * P;
      const uint4 dsc = *reinterpret_cast<const uint4*>(&s_near[min(sub, 4u)][0]);
      const int ddy = static_cast<int8_t>(dsc.x & 0xffu), ddz = static_cast<int8_t>((dsc.x >> 8) & 0xffu);
      const uint32_t valid = s_near_n ? (dsc.x >> 16) & 0xffu : 0u, axis = dsc.x >> 24;
      const unsigned long long ov = static_cast<unsigned long long>(dsc.z) | (static_cast<unsigned long long>(dsc.w) << 32);
      const bool live = t < n;
      const uint32_t xyz = live ? s_xyz[t] : 0u;
      const int bx = fr_bx(xyz), ny = fr_by(xyz) + ddy, nz = fr_bz(xyz) + ddz;
      uint32_t win = 0, raw = 0, nb0 = 0;
      int shw = 0;
      if (live && valid && ny >= 0 && ny < nby && nz < nbz)
        win = window(bx, ny, nz, raw, nb0, shw) & valid;
      if (!__any(win != 0u))
        continue;
      // One reservation per wave for both lists: ballots count the hits of the three adjacent slots (dx = -1, 0, +1), one LDS
      // atomic on the packed counter (face-neighbour hits in the upper 15 bits, the others in the lower 17) hands out the
      // space, and every hit's place follows from the ballots - no scan, no loop over the window.
      unsigned long long mA[3], mN[3];
      uint32_t totA = 0, totN = 0;
#pragma unroll
      for (int q = 0; q < 3; q++)
      {
        const int sl = R - 1 + q;
        const bool hit = (win >> sl) & 1u, ax = (axis >> sl) & 1u;
        mA[q] = __ballot(hit && ax);
        mN[q] = __ballot(hit && !ax);
        totA += static_cast<uint32_t>(__popcll(mA[q]));
        totN += static_cast<uint32_t>(__popcll(mN[q]));
      }
      static_assert(3 * LB_MAX < (1 << 15) && 10 * LB_MAX < (1 << 17), "fields of the packed hit counter");
      uint32_t base = 0;
      if (lane == 0)
        base = atomicAdd(&s_nhn, (totA << 17) | totN);
      base = __builtin_amdgcn_readfirstlane(base);
      uint32_t pA = base >> 17, pN = base & 0x1ffffu;
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int q = 0; q < 3; q++)
      {
        const int sl = R - 1 + q;
        if ((win >> sl) & 1u)
        {
          const uint32_t o = static_cast<uint32_t>(ov >> (8 * sl)) & 0xffu;
          const uint32_t t2 = nb0 + __popc(raw & ((1u << (sl - shw)) - 1u));
          const uint32_t hv = t | (t2 << 13) | (o << 26);
          if ((axis >> sl) & 1u)
            hits[pA + __popcll(mA[q] & below)] = hv;
          else
            hits_near[pN + __popcll(mN[q] & below)] = hv;
        }
        pA += static_cast<uint32_t>(__popcll(mA[q]));
        pN += static_cast<uint32_t>(__popcll(mN[q]));
      }
    }
  }
  __syncthreads();
  const uint32_t nh_axis = s_nhn >> 17, nh_near = s_nhn & 0x1ffffu;
  FR_STAMP(8);
  auto link = [&](uint32_t ra, uint32_t rb) {  // hook the larger root under the smaller (labels: smallest member)
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
  };
  auto flatten = [&]() {
    uint32_t roots[LB_MAX / FR_THREADS];
#pragma unroll
    for (int r = 0; r < LB_MAX / FR_THREADS; r++)
    {
      const uint32_t i = r * FR_THREADS + tid;
      uint32_t root = i < n ? i : 0u, p;
      if (i < n)
        while ((p = lb_ld16(s_par, root)) != root)
          root = p;
      roots[r] = root;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LB_MAX / FR_THREADS; r++)
      if (r * FR_THREADS + tid < n)
        s_par[r * FR_THREADS + tid] = static_cast<uint16_t>(roots[r]);
    __syncthreads();
  };
  constexpr int HU = 8;
  auto hits_pass = [&](const uint32_t* __restrict__ list, const uint32_t nh) {
    // a lane takes HU consecutive hits: they mostly share the brick t (the list is in D-a's order), whose root is then found once
    for (uint32_t i0 = tid * HU; i0 < nh; i0 += FR_THREADS * HU)
    {
      uint32_t hv[HU], da[HU], db[HU];
      bool act[HU];
#pragma unroll
      for (int u = 0; u < HU; u++)
      {
        const uint32_t i = i0 + u;
        hv[u] = i < nh ? list[i] : 0xffffffffu;
      }
#pragma unroll
      for (int u = 0; u < HU; u++)
      {
        act[u] = hv[u] != 0xffffffffu;
        const uint32_t t = act[u] ? hv[u] & 8191u : 0u, t2 = act[u] ? (hv[u] >> 13) & 8191u : 0u;
        // equal parents: one component already (after a flatten: equal roots) - nothing to test
        act[u] = act[u] && lb_ld16(s_par, t) != lb_ld16(s_par, t2);
        da[u] = s_xyz[t];
        db[u] = s_xyz[t2];
      }
      unsigned long long Ms[HU], Mm[HU];
#pragma unroll
      for (int u = 0; u < HU; u++)
      {
        const uint32_t o = act[u] ? hv[u] >> 26 : 0u;
        Ms[u] = s_tab.oct[2 * o];
        Mm[u] = s_tab.oct[2 * o + 1];
      }
      uint32_t kind[HU];  // 0 nothing, 1 accepted by the octant matrices, 2 open
#pragma unroll
      for (int u = 0; u < HU; u++)
      {
        const uint32_t A8 = da[u] >> 24, B8 = db[u] >> 24;
        kind[u] = !act[u] ? 0u : lb_octtest(Ms[u], A8, B8) ? 1u : lb_octtest(Mm[u], A8, B8) ? 2u : 0u;
      }
#pragma unroll
      for (int u = 0; u < HU; u++)
      {
        const unsigned long long m = __ballot(kind[u] == 2u);
        if (m)
        {
          const int leader = __ffsll(static_cast<long long>(m)) - 1;
          uint32_t base = 0;
          if (lane == leader)
            base = atomicAdd(&s_no, static_cast<uint32_t>(__popcll(m)));
          base = __builtin_amdgcn_readlane(base, leader);
          if (kind[u] == 2u)
            opens[base + __popcll(m & ((1ull << lane) - 1ull))] = hv[u];
        }
      }
      uint32_t cur_t = 0xffffffffu, cur_root = 0;
#pragma unroll
      for (int u = 0; u < HU; u++)
      {
        if (kind[u] != 1u)
          continue;
        const uint32_t t = hv[u] & 8191u, t2 = (hv[u] >> 13) & 8191u;
        cur_root = lb_find(s_par, t == cur_t ? cur_root : t);
        cur_t = t;
        const uint32_t ra = cur_root, rb = lb_find(s_par, t2);
        if (ra == rb)
          continue;
        cur_root = min(ra, rb);
        link(ra, rb);
      }
    }
  };
  hits_pass(hits, nh_axis);
  __syncthreads();
  flatten();
  hits_pass(hits_near, nh_near);
  __syncthreads();
  flatten();
  FR_STAMP(9);
  // D-a2: the bricks two apart.  G = the component most bricks belong to (any choice is correct: a pair of different
  // components has at least one end outside G); the bricks outside G look at their whole stencil, forward and backward,
  // and keep the neighbours of other components.
  if (wave == 0)
  {
    const uint32_t r = lb_ld16(s_par, static_cast<uint32_t>((static_cast<unsigned long long>(lane) * n) >> 6));
    uint32_t votes = 0;
    for (int k = 0; k < 64; k++)
      votes += __builtin_amdgcn_readlane(r, k) == r ? 1u : 0u;
    uint32_t best = (votes << 16) | r;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1)
      best = max(best, static_cast<uint32_t>(__shfl_xor(best, sft)));
    if (lane == 0)
    {
      s_nh = best & 0xffffu;  // G
      s_nn = 0;               // number of bricks outside G
    }
  }
  __syncthreads();
  const uint32_t G = s_nh;
  uint32_t* small = hits;  // (the face-neighbour list is dead)
  {
    const uint32_t n_r = (n + 63u) & ~63u;
    for (uint32_t i = tid; i < n_r; i += FR_THREADS)
    {
      const bool out = i < n && lb_ld16(s_par, i) != G;
      const unsigned long long m = __ballot(out);
      if (m)
      {
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&s_nn, static_cast<uint32_t>(__popcll(m)));
        base = __builtin_amdgcn_readlane(base, leader);
        if (out)
          small[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
      }
    }
  }
  __syncthreads();
  const uint32_t n_small = s_nn;
  {
    // work item = (brick outside G, stencil row, direction).  A thread's items all have the same (row, direction) when the
    // stride is a multiple of 32: the row's descriptor and its far slots are decoded once.
    const uint32_t items = n_small * 32u;
    const int row = (tid >> 1) & 15, back = tid & 1;
    int ddy, ddz;
    uint32_t valid, axis, near;
    unsigned long long ov;
    load_row(row, ddy, ddz, valid, axis, near, ov);
    uint32_t far = valid & ~near;  // slots of the half stencil two bricks away (forward view)
    if (back)
    {
      // the neighbour is the pair's base brick: it sees this brick at (-dx, ddy, ddz), i.e. slot 2R - s
      uint32_t rev = 0;
      for (int sl = 0; sl <= 2 * R; sl++)
        rev |= ((far >> (2 * R - sl)) & 1u) << sl;
      far = rev;
    }
    static_assert(FR_THREADS % 32 == 0, "items of a thread share row and direction");
    if (far)
      for (uint32_t it = tid; it < items; it += FR_THREADS)
      {
        const uint32_t t = small[it >> 5];
        const uint32_t xyz = s_xyz[t];
        const int bx = fr_bx(xyz), ny = fr_by(xyz) + (back ? -ddy : ddy), nz = fr_bz(xyz) + (back ? -ddz : ddz);
        if (ny < 0 || ny >= nby || nz < 0 || nz >= nbz)
          continue;
        uint32_t raw, nb0;
        int shw;
        uint32_t win = window(bx, ny, nz, raw, nb0, shw) & far;
        const uint32_t rt = lb_ld16(s_par, t);
        while (win)
        {
          const int sl = __ffs(static_cast<int>(win)) - 1;
          win &= win - 1;
          const uint32_t t2 = nb0 + __popc(raw & ((1u << (sl - shw)) - 1u));
          if (lb_ld16(s_par, t2) == rt)
            continue;  // one component already
          const uint32_t o = static_cast<uint32_t>(ov >> (8 * (back ? 2 * R - sl : sl))) & 0xffu;
          const uint32_t pos = atomicAdd(&s_nf, 1u);
          if (pos < cap_far)
            hits_far[pos] = back ? (t2 | (t << 13) | (o << 26)) : (t | (t2 << 13) | (o << 26));
        }
      }
  }
  __syncthreads();
  const uint32_t nh_far = s_nf;
  if (nh_far > cap_far)
  {
    if (tid == 0)
    {
      h.status = CCL_RETRY_STATUS;
      h.V = 0;
    }
    return;
  }
  hits_pass(hits_far, nh_far);
  __syncthreads();
  flatten();
  const uint32_t no = s_no;
  const uint32_t nh = nh_axis + nh_near + nh_far;
  FR_STAMP(10);
  // D-c: keep the open pairs whose ends still sit in different components (roots after the flatten) ...
  if (tid == 0)
    s_nh = 0;
  __syncthreads();
  {
    const uint32_t no_round = (no + 63u) & ~63u;
    for (uint32_t i = tid; i < no_round; i += FR_THREADS)
    {
      const uint32_t hv = i < no ? opens[i] : 0xffffffffu;
      bool keep = false;
      if (hv != 0xffffffffu)
        keep = lb_ld16(s_par, hv & 8191u) != lb_ld16(s_par, (hv >> 13) & 8191u);
      const unsigned long long m = __ballot(keep);
      if (m)
      {
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&s_nh, static_cast<uint32_t>(__popcll(m)));
        base = __builtin_amdgcn_readlane(base, leader);
        if (keep)
          hits[base + __popcll(m & ((1ull << lane) - 1ull))] = hv;  // the hit list is dead: it holds the survivors
      }
    }
  }
  __syncthreads();
  const uint32_t n_surv = s_nh;
  // ... and test them exactly, one per lane (components merge while this runs: the roots are looked up again)
  for (uint32_t i = tid; i < n_surv; i += FR_THREADS)
  {
    const uint32_t hv = hits[i];
    const uint32_t t = hv & 8191u, t2 = (hv >> 13) & 8191u;
    uint32_t ra = lb_find(s_par, t), rb = lb_find(s_par, t2);
    if (ra == rb)
      continue;
    const uint32_t xa = s_xyz[t], xb = s_xyz[t2];
    const int bx = fr_bx(xa), by = fr_by(xa), bz = fr_bz(xa);
    const int ddx = fr_bx(xb) - bx, ddy = fr_by(xb) - by, ddz = fr_bz(xb) - bz;
    if (!lb_pair_conn(s_tab, g, bp, h, s_word[t], s_word[t2], 4 * bx + o0, 4 * by + o1, 4 * bz + o2, ddx, ddy, ddz))
      continue;
    ra = lb_find(s_par, ra);
    rb = lb_find(s_par, rb);
    link(ra, rb);
  }
  __syncthreads();
  FR_STAMP(11);
  if (prof && tid == 0)
  {
    prof[static_cast<size_t>(FRAME) * 32 + 24] = nh;
    prof[static_cast<size_t>(FRAME) * 32 + 25] = no;
    prof[static_cast<size_t>(FRAME) * 32 + 26] = n;
    prof[static_cast<size_t>(FRAME) * 32 + 27] = n_keys;
    prof[static_cast<size_t>(FRAME) * 32 + 28] = s_ne;
    prof[static_cast<size_t>(FRAME) * 32 + 29] = V;
    prof[static_cast<size_t>(FRAME) * 32 + 30] = n_surv;
    prof[static_cast<size_t>(FRAME) * 32 + 31] = n_small | (static_cast<unsigned long long>(nh_far) << 32);
  }
  // ---- E: component minima: the smallest voxel rank of a component is the first voxel of one of its bricks
  uint32_t my_root[LB_MAX / FR_THREADS], my_min[LB_MAX / FR_THREADS];
  unsigned long long my_w[LB_MAX / FR_THREADS];
#pragma unroll
  for (int r = 0; r < LB_MAX / FR_THREADS; r++)
  {
    const uint32_t i = r * FR_THREADS + tid;
    my_root[r] = 0xffffffffu;
    my_min[r] = 0xffffffffu;
    my_w[r] = 0ull;
    if (i < n)
    {
      uint32_t root = i, p;
      while ((p = lb_ld16(s_par, root)) != root)
        root = p;
      my_root[r] = root;
      my_w[r] = s_word[i];
      my_min[r] = bmin_g[i];
    }
  }
  __syncthreads();  // roots and words are in registers: flatten the forest, turn the words into the minima
#pragma unroll
  for (int r = 0; r < LB_MAX / FR_THREADS; r++)
  {
    const uint32_t i = r * FR_THREADS + tid;
    if (i < n)
    {
      s_par[i] = static_cast<uint16_t>(my_root[r]);
      s_cmin[i] = 0xffffffffu;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LB_MAX / FR_THREADS; r++)
    if (my_root[r] != 0xffffffffu)
      atomicMin(&s_cmin[my_root[r]], my_min[r]);
  __syncthreads();
  // ---- cluster statistics (size, lattice box, close flag) brick by brick
  uint16_t* s_cidx = reinterpret_cast<uint16_t*>(s_word + LB_MAX / 2);               // node (root) -> component index
  uint32_t* st_label = reinterpret_cast<uint32_t*>(s_cidx + LB_MAX);                 // LB_ST_ROWS x {label, count, close, box[6]}
  uint32_t* st_cnt = st_label + LB_ST_ROWS;
  uint32_t* st_close = st_cnt + LB_ST_ROWS;
  int* st_box = reinterpret_cast<int*>(st_close + LB_ST_ROWS);
  if (tid == 0)
    s_nh = 0;  // reused: number of components
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LB_MAX / FR_THREADS; r++)
  {
    const bool is_root = my_root[r] == static_cast<uint32_t>(r * FR_THREADS + tid);
    const unsigned long long m = __ballot(is_root);
    if (!m)
      continue;
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    uint32_t base = 0;
    if (lane == leader)
      base = atomicAdd(&s_nh, static_cast<uint32_t>(__popcll(m)));
    base = __builtin_amdgcn_readlane(base, leader);
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
  for (int r = 0; r < LB_MAX / FR_THREADS; r++)
  {
    const uint32_t i = r * FR_THREADS + tid;
    if (r * FR_THREADS + (tid & ~63) >= n)  // the whole wave is past the last brick
      continue;
    const bool live = i < n;
    const unsigned long long W = live ? my_w[r] : 1ull;
    const uint32_t c = live ? s_cidx[my_root[r]] : 0xffffffffu;
    const uint32_t xyz = live ? s_xyz[i] : 0u;
    const int bx = fr_bx(xyz), by = fr_by(xyz), bz = fr_bz(xyz);
    unsigned long long t = W | (W >> 16) | (W >> 32) | (W >> 48);
    uint32_t ox = static_cast<uint32_t>(t) & 0xffffu;
    ox = (ox | (ox >> 4) | (ox >> 8) | (ox >> 12)) & 0xfu;
    t = W | (W >> 1);
    t |= t >> 2;  // bit 4y + 16z: row (y,z) is occupied
    unsigned long long ty = t | (t >> 16) | (t >> 32) | (t >> 48);
    const uint32_t oy = (static_cast<uint32_t>(ty) & 1u) | ((static_cast<uint32_t>(ty) >> 3) & 2u) | ((static_cast<uint32_t>(ty) >> 6) & 4u) | ((static_cast<uint32_t>(ty) >> 9) & 8u);
    const uint32_t oz = ((W & 0xffffull) ? 1u : 0u) | ((W & 0xffff0000ull) ? 2u : 0u) | ((W & 0xffff00000000ull) ? 4u : 0u) | ((W >> 48) ? 8u : 0u);
    const int lo[3] = {4 * bx + o0 + __ffs(static_cast<int>(ox)) - 1, 4 * by + o1 + __ffs(static_cast<int>(oy)) - 1, 4 * bz + o2 + __ffs(static_cast<int>(oz)) - 1};
    const int hi[3] = {4 * bx + o0 + 31 - __clz(static_cast<int>(ox)), 4 * by + o1 + 31 - __clz(static_cast<int>(oy)), 4 * bz + o2 + 31 - __clz(static_cast<int>(oz))};
    uint32_t cnt = __popcll(W);
    bool hit = false;
    if (live && mapclose && !(c < LB_ST_ROWS ? st_close[c] : 0u))
    {
      unsigned long long a = W;
      while (a && !hit)
      {
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
          int mx_, my_, mz_;
          if (s_mapk[3])
          {
            // the lattice is a translate of the map's by whole voxels (checked in the prologue): the map cell of a voxel
            // centre is its lattice cell plus a constant - integer adds instead of three float expressions per voxel
            mx_ = 4 * bx + (p & 3) + s_mapk[0];
            my_ = 4 * by + ((p >> 2) & 3) + s_mapk[1];
            mz_ = 4 * bz + (p >> 4) + s_mapk[2];
          }
          else
          {
            const float cx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bx + o0 + (p & 3)), 0.5f), g.leaf[0]), hoff0);
            const float cy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * by + o1 + ((p >> 2) & 3)), 0.5f), g.leaf[1]), hoff1);
            const float cz = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(4 * bz + o2 + (p >> 4)), 0.5f), g.leaf[2]), hoff2);
            mx_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cx, mg.off[0]), mg.vs_inv)));
            my_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cy, mg.off[1]), mg.vs_inv)));
            mz_ = static_cast<int>(floorf(__fmul_rn(__fsub_rn(cz, mg.off[2]), mg.vs_inv)));
          }
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
    {
      const uint32_t lead = __builtin_amdgcn_readlane(c, __ffsll(static_cast<long long>(__ballot(1))) - 1);
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
        rc = wave_sum(rc);
#pragma unroll
        for (int a = 0; a < 3; a++)
        {
          rlo[a] = wave_min(rlo[a]);
          rhi[a] = wave_max(rhi[a]);
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
  uint8_t* st_cand = reinterpret_cast<uint8_t*>(st_box + 6 * LB_ST_ROWS);
  ClusterRec* table = table_all + static_cast<size_t>(FRAME) * g.vox_cap;  // (the hit list that lived here is dead)
  CandMember* cands = cand_all + static_cast<size_t>(FRAME) * g.vox_cap;
  {
    const uint32_t nc = min(s_nh, static_cast<uint32_t>(LB_ST_ROWS));
    for (uint32_t c = tid; c < nc; c += FR_THREADS)
    {
      const uint32_t label = st_label[c];
      // (per-label slots serve the unfused consumers)
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
  FR_STAMP(12);
  // ---- labels (+ candidate members): a voxel's brick node was stored with its record
  for (uint32_t v0 = tid; v0 < Vround; v0 += FR_THREADS * VU)
  {
    uint32_t nodev[VU];
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      const uint32_t v = v0 + u * FR_THREADS;
      nodev[u] = v < V ? reinterpret_cast<const uint16_t*>(va.bb)[v] : 0xffffu;
    }
#pragma unroll
    for (int u = 0; u < VU; u++)
    {
      const uint32_t v = v0 + u * FR_THREADS;
      if (v0 + u * FR_THREADS - tid >= Vround)  // block-uniform
        break;
      bool cand = false;
      uint32_t label = 0;
      if (v < V)
      {
        const uint32_t node = nodev[u];
        const uint32_t root = s_par[node];
        label = s_cmin[root];
        labels[v] = label;
        if (write_tables)
        {
          const uint32_t c = s_cidx[root];
          if (c < LB_ST_ROWS)
            cand = st_cand[c] != 0;
          else
          {
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
          base = __builtin_amdgcn_readlane(base, leader);
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
  FR_STAMP(13);
  if (tid == 0)
    h.n_bricks = n;
#undef FR_STAMP
}

// the instantiations under names without a comma (the launch macro records the kernel's name as written)
// (`_p`: every frame of the batch has packed float columns, 16-byte aligned, a multiple of 4 points - 16-byte loads in the input pass)
constexpr auto k_frame_lds_full = &k_frame_lds<0, false>;   // voxelise + cluster everything (debug view, cold maps)
constexpr auto k_frame_lds_full_p = &k_frame_lds<0, true>;
constexpr auto k_frame_lds_far = &k_frame_lds<1, false>;    // voxelise + cluster the far voxels only (read-only batches)
constexpr auto k_frame_lds_far_p = &k_frame_lds<1, true>;

}  // namespace vk
