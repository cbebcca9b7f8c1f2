// Micro-benchmark of k_key1 (the batch path's only pass over the input) against a variant that reserves its output per WAVE
// instead of per workgroup: no barrier, no LDS, no wait of three waves for the fourth.  DESIGN.md 8: "not tried yet" after round 4
// (the kernel is bound neither by its vector instructions nor by HBM: 3.5 TB/s, VALU ~50 % busy, 55 % of the wave-cycles waiting).
// Both kernels run on the same synthetic batch, interleaved rounds in ONE process (cdna_hip_programming.md rule 24); the
// outputs are compared as sets (lists are unordered across workgroups / waves by construction).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vofod_amd/csrc -I include -o tools/ubench/key1 tools/ubench/key1.hip && tools/ubench/key1
// The variant's body above the epilogue is a copy of k_key1's (kernels_frame.h at the end of round 4)
// - copy it again when k_key1 changes: only the epilogue differs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "common.h"
#include "kernels_frame.h"
#include "kernels_voxelize.h"

#define CHK(x)                                                      \
  do                                                                \
  {                                                                 \
    hipError_t e_ = (x);                                            \
    if (e_ != hipSuccess)                                           \
    {                                                               \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));           \
      std::exit(1);                                                 \
    }                                                               \
  } while (0)

namespace vk
{
template <bool PACKED>
__global__ __launch_bounds__(KEY1_THREADS) void k_key1_wave(const FrameArgs* __restrict__ args, const GridParams g, FrameHdr* __restrict__ hdrs, SlabArrays sa, uint32_t pt_cap, const RefLattice rl)
{
#pragma clang fp contract(off)
  // (2-D grid: blockIdx.y = frame - the frame / block split of a 1-D grid costs two integer divisions per wave, ~7 % of this
  // kernel's vector instructions)
  const uint32_t FRAME = blockIdx.y, BX = blockIdx.x;
  const FrameArgs a = args[FRAME];  // (a copy: the transform stays in scalar registers)
  const uint32_t base_blk = BX * KEY1_THREADS * KEY1_PPT;
  if (base_blk >= a.n)
    return;
  const uint32_t i0 = base_blk + threadIdx.x * KEY1_PPT;
  float px[KEY1_PPT], py[KEY1_PPT], pz[KEY1_PPT];
  if constexpr (PACKED)
  {
    // packed float columns, 16-byte aligned, the number of points a multiple of 4 (the host checks): whole 16-byte loads
    // only, no strided path in this instantiation (its 64-bit address arithmetic costs registers the packed path never uses)
#pragma unroll
    for (int q = 0; q < KEY1_PPT / 4; q++)
    {
      // (a quad behind the cloud's end reads the last quad instead - the number of points is a multiple of 4, and at least 4
      // here; the index test below drops its points: no conditional load, no registers to clear first)
      const uint64_t o = static_cast<uint64_t>(min(i0 + 4u * q, a.n - 4u)) * 4;
      const float4 x0 = ldg_f4(a.x + o), y0 = ldg_f4(a.y + o), z0 = ldg_f4(a.z + o);
      px[4 * q] = x0.x, px[4 * q + 1] = x0.y, px[4 * q + 2] = x0.z, px[4 * q + 3] = x0.w;
      py[4 * q] = y0.x, py[4 * q + 1] = y0.y, py[4 * q + 2] = y0.z, py[4 * q + 3] = y0.w;
      pz[4 * q] = z0.x, pz[4 * q + 1] = z0.y, pz[4 * q + 2] = z0.z, pz[4 * q + 3] = z0.w;
    }
  }
  else
  {
#pragma unroll
    for (int j = 0; j < KEY1_PPT; j++)
    {
      const uint32_t i = i0 + j;
      const bool ok = i < a.n;
      px[j] = ok ? ldf(a.x, a.stride, i) : 0.0f;
      py[j] = ok ? ldf(a.y, a.stride, i) : 0.0f;
      pz[j] = ok ? ldf(a.z, a.stride, i) : 0.0f;
    }
  }
  float fmn[3] = {INFINITY, INFINITY, INFINITY}, fmx[3] = {-INFINITY, -INFINITY, -INFINITY};
  uint32_t code[KEY1_PPT];
  uint32_t cnt = 0, frag_mask = 0;
  uint32_t* frag = sa.extras + static_cast<size_t>(FRAME) * pt_cap;  // fragile points: their transformed coordinates, 3 floats each
  float sq0 = 0.0f, sq1 = 0.0f, sq2 = 0.0f;  // ... of the thread's FIRST fragile point (8 % of the threads have one, 0.3 % a second)
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
#pragma unroll
  for (int jp = 0; jp < KEY1_PPT / 2; jp++)
  {
    const int j0 = 2 * jp, j1 = 2 * jp + 1;
    const f32x2 X = {px[j0], px[j1]}, Y = {py[j0], py[j1]}, Z = {pz[j0], pz[j1]};
    bool in_ex[2], in_op[2], keep[2];
#pragma unroll
    for (int e = 0; e < 2; e++)
      in_ex[e] = static_cast<int>(inside(X[e], g.ex_min[0], ex_hi[0])) & inside(Y[e], g.ex_min[1], ex_hi[1]) & inside(Z[e], g.ex_min[2], ex_hi[2]);
    f32x2 q[3];
#pragma unroll
    for (int r = 0; r < 3; r++)  // pcl::detail::Transformer<float>::se3: c0*x + (c1*y + (c2*z + c3)), every op rounded
      q[r] = pk_mul_s(a.tf[4 * r + 0], X) + (pk_mul_s(a.tf[4 * r + 1], Y) + pk_add_s(a.tf[4 * r + 3], pk_mul_s(a.tf[4 * r + 2], Z)));
#pragma unroll
    for (int e = 0; e < 2; e++)
    {
      in_op[e] = static_cast<int>(inside(q[0][e], g.op_min[0], op_hi[0])) & inside(q[1][e], g.op_min[1], op_hi[1]) & inside(q[2][e], g.op_min[2], op_hi[2]);
      keep[e] = (i0 + j0 + e < a.n) & !in_ex[e] & in_op[e];
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
      const bool fits = (k0 | k1 | (k2 << 1)) < 2048u;
      const bool solid = keep[e] & fits & (max3_abs(gmid[0][e], gmid[1][e], gmid[2][e]) <= solid_lim);
      cnt += solid ? 1u : 0u;
      code[j0 + e] = solid ? (k0 | (k1 << 11) | (k2 << 22)) : FR_CODE_NONE;
      const bool fragile = keep[e] & !solid;  // kept aside with its transformed coordinates: k_frame_lds encodes it with the frame's own offset
      const bool first_fr = fragile & (frag_mask == 0u);
      sq0 = first_fr ? q[0][e] : sq0;
      sq1 = first_fr ? q[1][e] : sq1;
      sq2 = first_fr ? q[2][e] : sq2;
      frag_mask |= fragile ? (1u << (j0 + e)) : 0u;
    }
  }
  // ---- epilogue, per WAVE: every wave reserves its own piece of the two lists (one returning atomic), no barrier, no LDS.
  // (The order of the code list across workgroups is the order of their atomics in k_key1 already: nothing depends on it.)
  int mn[3], mx[3];
  wave_bbox(fmn, fmx);
#pragma unroll
  for (int c = 0; c < 3; c++)
  {
    mn[c] = f2ord(fmn[c]);
    mx[c] = f2ord(fmx[c]);
  }
  const int lane = threadIdx.x & 63;
  const uint32_t incl = wave_incl_scan(cnt);
  const uint32_t fcnt = __popc(frag_mask), fincl = wave_incl_scan(fcnt);
  const uint32_t total = __builtin_amdgcn_readlane(incl, 63), ftotal = __builtin_amdgcn_readlane(fincl, 63);
  if ((total | ftotal) == 0u)
    return;  // (wave-uniform)
  unsigned long long both = 0ull;
  if (lane == 0)
  {
    both = atomicAdd(reinterpret_cast<unsigned long long*>(&sa.counts[2 * FRAME]), static_cast<unsigned long long>(total) | (static_cast<unsigned long long>(ftotal) << 32));
    FrameHdr& h = hdrs[FRAME];
    atomicAdd(&h.n_in, total + ftotal);
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
      atomicMin(&h.bb_min[c], mn[c]);
      atomicMax(&h.bb_max[c], mx[c]);
    }
  }
  const uint32_t base = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(both)), fbase = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(both >> 32));
  uint32_t* out = sa.keys + static_cast<size_t>(FRAME) * pt_cap + base + (incl - cnt);
#pragma unroll
  for (int j = 0; j < KEY1_PPT; j++)
    if (code[j] != FR_CODE_NONE)
      *out++ = code[j];
  if (frag_mask)
  {
    float* fout = reinterpret_cast<float*>(frag) + 3u * (fbase + (fincl - fcnt));
    fout[0] = sq0, fout[1] = sq1, fout[2] = sq2;
    fout += 3;
    uint32_t rest = frag_mask & (frag_mask - 1u);
    while (rest)
    {
      const uint32_t pi = i0 + static_cast<uint32_t>(__ffs(static_cast<int>(rest)) - 1);
      rest &= rest - 1u;
      const uint64_t st = PACKED ? 4u : a.stride;
      const float p0 = ldf(a.x, st, pi), p1 = ldf(a.y, st, pi), p2 = ldf(a.z, st, pi);
#pragma unroll
      for (int r = 0; r < 3; r++)
        *fout++ = __fadd_rn(__fmul_rn(a.tf[4 * r + 0], p0), __fadd_rn(__fmul_rn(a.tf[4 * r + 1], p1), __fadd_rn(__fmul_rn(a.tf[4 * r + 2], p2), a.tf[4 * r + 3])));
    }
  }
}
template <bool PACKED>
__global__ __launch_bounds__(KEY1_THREADS) void k_key1_seg(const FrameArgs* __restrict__ args, const GridParams g, FrameHdr* __restrict__ hdrs, SlabArrays sa, uint32_t pt_cap, const RefLattice rl, uint32_t* __restrict__ segcnt)
{
#pragma clang fp contract(off)
  // (2-D grid: blockIdx.y = frame - the frame / block split of a 1-D grid costs two integer divisions per wave, ~7 % of this
  // kernel's vector instructions)
  const uint32_t FRAME = blockIdx.y, BX = blockIdx.x;
  const FrameArgs a = args[FRAME];  // (a copy: the transform stays in scalar registers)
  const uint32_t base_blk = BX * KEY1_THREADS * KEY1_PPT;
  if (base_blk >= a.n)
    return;
  const uint32_t i0 = base_blk + threadIdx.x * KEY1_PPT;
  float px[KEY1_PPT], py[KEY1_PPT], pz[KEY1_PPT];
  if constexpr (PACKED)
  {
    // packed float columns, 16-byte aligned, the number of points a multiple of 4 (the host checks): whole 16-byte loads
    // only, no strided path in this instantiation (its 64-bit address arithmetic costs registers the packed path never uses)
#pragma unroll
    for (int q = 0; q < KEY1_PPT / 4; q++)
    {
      // (a quad behind the cloud's end reads the last quad instead - the number of points is a multiple of 4, and at least 4
      // here; the index test below drops its points: no conditional load, no registers to clear first)
      const uint64_t o = static_cast<uint64_t>(min(i0 + 4u * q, a.n - 4u)) * 4;
      const float4 x0 = ldg_f4(a.x + o), y0 = ldg_f4(a.y + o), z0 = ldg_f4(a.z + o);
      px[4 * q] = x0.x, px[4 * q + 1] = x0.y, px[4 * q + 2] = x0.z, px[4 * q + 3] = x0.w;
      py[4 * q] = y0.x, py[4 * q + 1] = y0.y, py[4 * q + 2] = y0.z, py[4 * q + 3] = y0.w;
      pz[4 * q] = z0.x, pz[4 * q + 1] = z0.y, pz[4 * q + 2] = z0.z, pz[4 * q + 3] = z0.w;
    }
  }
  else
  {
#pragma unroll
    for (int j = 0; j < KEY1_PPT; j++)
    {
      const uint32_t i = i0 + j;
      const bool ok = i < a.n;
      px[j] = ok ? ldf(a.x, a.stride, i) : 0.0f;
      py[j] = ok ? ldf(a.y, a.stride, i) : 0.0f;
      pz[j] = ok ? ldf(a.z, a.stride, i) : 0.0f;
    }
  }
  float fmn[3] = {INFINITY, INFINITY, INFINITY}, fmx[3] = {-INFINITY, -INFINITY, -INFINITY};
  uint32_t code[KEY1_PPT];
  uint32_t cnt = 0, frag_mask = 0;
  uint32_t* frag = sa.extras + static_cast<size_t>(FRAME) * pt_cap;  // fragile points: their transformed coordinates, 3 floats each
  float sq0 = 0.0f, sq1 = 0.0f, sq2 = 0.0f;  // ... of the thread's FIRST fragile point (8 % of the threads have one, 0.3 % a second)
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
#pragma unroll
  for (int jp = 0; jp < KEY1_PPT / 2; jp++)
  {
    const int j0 = 2 * jp, j1 = 2 * jp + 1;
    const f32x2 X = {px[j0], px[j1]}, Y = {py[j0], py[j1]}, Z = {pz[j0], pz[j1]};
    bool in_ex[2], in_op[2], keep[2];
#pragma unroll
    for (int e = 0; e < 2; e++)
      in_ex[e] = static_cast<int>(inside(X[e], g.ex_min[0], ex_hi[0])) & inside(Y[e], g.ex_min[1], ex_hi[1]) & inside(Z[e], g.ex_min[2], ex_hi[2]);
    f32x2 q[3];
#pragma unroll
    for (int r = 0; r < 3; r++)  // pcl::detail::Transformer<float>::se3: c0*x + (c1*y + (c2*z + c3)), every op rounded
      q[r] = pk_mul_s(a.tf[4 * r + 0], X) + (pk_mul_s(a.tf[4 * r + 1], Y) + pk_add_s(a.tf[4 * r + 3], pk_mul_s(a.tf[4 * r + 2], Z)));
#pragma unroll
    for (int e = 0; e < 2; e++)
    {
      in_op[e] = static_cast<int>(inside(q[0][e], g.op_min[0], op_hi[0])) & inside(q[1][e], g.op_min[1], op_hi[1]) & inside(q[2][e], g.op_min[2], op_hi[2]);
      keep[e] = (i0 + j0 + e < a.n) & !in_ex[e] & in_op[e];
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
      const bool fits = (k0 | k1 | (k2 << 1)) < 2048u;
      const bool solid = keep[e] & fits & (max3_abs(gmid[0][e], gmid[1][e], gmid[2][e]) <= solid_lim);
      cnt += solid ? 1u : 0u;
      code[j0 + e] = solid ? (k0 | (k1 << 11) | (k2 << 22)) : FR_CODE_NONE;
      const bool fragile = keep[e] & !solid;  // kept aside with its transformed coordinates: k_frame_lds encodes it with the frame's own offset
      const bool first_fr = fragile & (frag_mask == 0u);
      sq0 = first_fr ? q[0][e] : sq0;
      sq1 = first_fr ? q[1][e] : sq1;
      sq2 = first_fr ? q[2][e] : sq2;
      frag_mask |= fragile ? (1u << (j0 + e)) : 0u;
    }
  }
  // ---- epilogue, per wave, NO returning atomic: a wave owns the 64 * KEY1_PPT slots of its own points in both lists (a fixed
  // segment) and publishes its two counts in a table; the consumer walks the segments.  Nothing to wait for but the loads.
  int mn[3], mx[3];
  wave_bbox(fmn, fmx);
#pragma unroll
  for (int c = 0; c < 3; c++)
  {
    mn[c] = f2ord(fmn[c]);
    mx[c] = f2ord(fmx[c]);
  }
  const int lane = threadIdx.x & 63;
  const uint32_t wave_in_frame = BX * (KEY1_THREADS / 64) + (threadIdx.x >> 6);
  const uint32_t incl = wave_incl_scan(cnt);
  const uint32_t fcnt = __popc(frag_mask), fincl = wave_incl_scan(fcnt);
  const uint32_t total = __builtin_amdgcn_readlane(incl, 63), ftotal = __builtin_amdgcn_readlane(fincl, 63);
  if (lane == 0)
  {
    segcnt[FRAME * gridDim.x * (KEY1_THREADS / 64) + wave_in_frame] = total | (ftotal << 16);
    if (total | ftotal)
    {
      FrameHdr& h = hdrs[FRAME];
      atomicAdd(&h.n_in, total + ftotal);
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        atomicMin(&h.bb_min[c], mn[c]);
        atomicMax(&h.bb_max[c], mx[c]);
      }
    }
  }
  const uint32_t segbase = wave_in_frame * 64u * KEY1_PPT;
  uint32_t* out = sa.keys + static_cast<size_t>(FRAME) * pt_cap + segbase + (incl - cnt);
#pragma unroll
  for (int j = 0; j < KEY1_PPT; j++)
    if (code[j] != FR_CODE_NONE)
      *out++ = code[j];
  if (frag_mask)
  {
    float* fout = reinterpret_cast<float*>(sa.extras) + 3u * (static_cast<size_t>(FRAME) * pt_cap + segbase + (fincl - fcnt));  // (this list: 3 words per point, pitch 3 * pt_cap)
    fout[0] = sq0, fout[1] = sq1, fout[2] = sq2;
    fout += 3;
    uint32_t rest = frag_mask & (frag_mask - 1u);
    while (rest)
    {
      const uint32_t pi = i0 + static_cast<uint32_t>(__ffs(static_cast<int>(rest)) - 1);
      rest &= rest - 1u;
      const uint64_t st = PACKED ? 4u : a.stride;
      const float p0 = ldf(a.x, st, pi), p1 = ldf(a.y, st, pi), p2 = ldf(a.z, st, pi);
#pragma unroll
      for (int r = 0; r < 3; r++)
        *fout++ = __fadd_rn(__fmul_rn(a.tf[4 * r + 0], p0), __fadd_rn(__fmul_rn(a.tf[4 * r + 1], p1), __fadd_rn(__fmul_rn(a.tf[4 * r + 2], p2), a.tf[4 * r + 3])));
    }
  }
}
}  // namespace vk

using namespace vk;

__global__ void k_reset(FrameHdr* hdrs, uint32_t* counts, uint32_t n)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n)
    return;
  FrameHdr& h = hdrs[f];
  for (int c = 0; c < 3; c++)
  {
    h.bb_min[c] = 0x7fffffff;
    h.bb_max[c] = static_cast<int>(0x80000000u);
  }
  h.n_in = 0;
  counts[2 * f] = counts[2 * f + 1] = 0;
}

int main(int argc, char** argv)
{
  const uint32_t F = argc > 1 ? std::atoi(argv[1]) : 256, N = 131072, rounds = argc > 2 ? std::atoi(argv[2]) : 10;
  // an OS1-128-like batch: 35 % of the rays return nothing (0, 0, 0), the rest lie 2-60 m away; two alternating input sets
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  std::vector<float> hx(static_cast<size_t>(N) * 3 * 2);
  for (int set = 0; set < 2; set++)
    for (uint32_t i = 0; i < N; i++)
    {
      float* p = &hx[static_cast<size_t>(set) * 3 * N];
      if (U(rng) < 0.35f)
        p[i] = p[N + i] = p[2 * N + i] = 0.0f;
      else
      {
        const float r = 2.0f + 58.0f * U(rng), az = 6.2831853f * U(rng), el = -0.39f + 0.78f * U(rng);
        p[i] = r * std::cos(el) * std::cos(az), p[N + i] = r * std::cos(el) * std::sin(az), p[2 * N + i] = r * std::sin(el);
      }
    }
  float* d_in;
  CHK(hipMalloc(&d_in, hx.size() * 4));
  CHK(hipMemcpy(d_in, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  GridParams g{};
  for (int a = 0; a < 3; a++)
  {
    g.leaf[a] = 0.25f;
    g.inv[a] = 4.0f;
    g.aco[a] = 0.0f;
    g.ex_min[a] = -1.0f;
    g.ex_max[a] = 1.0f;
  }
  const float opmin[3] = {-20.0f, -30.0f, -1.25f}, opmax[3] = {100.0f, 70.0f, 23.75f};  // sim.yaml's operation area
  for (int a = 0; a < 3; a++)
    g.op_min[a] = opmin[a], g.op_max[a] = opmax[a];
  g.n_frames = F;
  RefLattice rl{};
  for (int a = 0; a < 3; a++)
  {
    rl.off[a] = std::floor(g.op_min[a] * g.inv[a]) * g.leaf[a];
    rl.dims[a] = static_cast<int>(std::floor((g.op_max[a] - rl.off[a]) * g.inv[a])) + 2;
  }
  rl.eps = 2e-3f;
  rl.on = 1;
  std::vector<FrameArgs> ha(F);
  for (uint32_t f = 0; f < F; f++)
  {
    FrameArgs& a = ha[f];
    const float* p = d_in + static_cast<size_t>(f & 1) * 3 * N;
    a.x = reinterpret_cast<const char*>(p);
    a.y = reinterpret_cast<const char*>(p + N);
    a.z = reinterpret_cast<const char*>(p + 2 * N);
    a.intensity = nullptr;
    a.stride = 4;
    a.n = N;
    a.flags = FA_SCAN;
    const float yaw = 0.01f * f, c = std::cos(yaw), s = std::sin(yaw);
    const float tf[12] = {c, -s, 0, 0.3f * (f % 7), s, c, 0, -0.2f * (f % 5), 0, 0, 1, 2.0f};
    std::memcpy(a.tf, tf, sizeof tf);
  }
  FrameArgs* d_args;
  FrameHdr* d_hdrs;
  SlabArrays sa[3];
  uint32_t* d_segcnt;
  const uint32_t WPF = ((N + KEY1_THREADS * KEY1_PPT - 1) / (KEY1_THREADS * KEY1_PPT)) * (KEY1_THREADS / 64);  // waves (= segments) per frame
  CHK(hipMalloc(&d_segcnt, sizeof(uint32_t) * F * WPF));
  CHK(hipMalloc(&d_args, sizeof(FrameArgs) * F));
  CHK(hipMemcpy(d_args, ha.data(), sizeof(FrameArgs) * F, hipMemcpyHostToDevice));
  CHK(hipMalloc(&d_hdrs, sizeof(FrameHdr) * F * 3));
  for (int v = 0; v < 3; v++)
  {
    CHK(hipMalloc(&sa[v].keys, sizeof(uint32_t) * F * static_cast<size_t>(N)));
    CHK(hipMalloc(&sa[v].extras, sizeof(uint32_t) * F * static_cast<size_t>(N) * (v == 2 ? 3 : 1)));
    CHK(hipMalloc(&sa[v].counts, sizeof(uint32_t) * 2 * F));
  }
  const dim3 grid((N + KEY1_THREADS * KEY1_PPT - 1) / (KEY1_THREADS * KEY1_PPT), F);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  double sum[3] = {0, 0, 0};
  for (uint32_t r = 0; r < rounds + 2; r++)
    for (int v = 0; v < 3; v++)
    {
      k_reset<<<(F + 63) / 64, 64>>>(d_hdrs + v * F, sa[v].counts, F);
      if (v == 2)
        CHK(hipMemsetAsync(d_segcnt, 0, sizeof(uint32_t) * F * WPF));
      CHK(hipEventRecord(e0));
      if (v == 0)
        k_key1<true><<<grid, KEY1_THREADS>>>(d_args, g, d_hdrs, sa[0], N, rl);
      else if (v == 1)
        k_key1_wave<true><<<grid, KEY1_THREADS>>>(d_args, g, d_hdrs + F, sa[1], N, rl);
      else
        k_key1_seg<true><<<grid, KEY1_THREADS>>>(d_args, g, d_hdrs + 2 * F, sa[2], N, rl, d_segcnt);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2)
        sum[v] += ms;
    }
  std::printf("k_key1       %.1f us per %u frames (%.2f TB/s of input)\n", 1e3 * sum[0] / rounds, F, 12.0 * N * F / (sum[0] / rounds * 1e-3) / 1e12);
  std::printf("k_key1_wave  %.1f us per %u frames (%.2f TB/s of input)\n", 1e3 * sum[1] / rounds, F, 12.0 * N * F / (sum[1] / rounds * 1e-3) / 1e12);
  std::printf("k_key1_seg   %.1f us per %u frames (%.2f TB/s of input)\n", 1e3 * sum[2] / rounds, F, 12.0 * N * F / (sum[2] / rounds * 1e-3) / 1e12);
  {
    // the segmented lists against the baseline: per-frame totals, and frame 0's codes / fragile points as sorted lists
    std::vector<uint32_t> sc(static_cast<size_t>(F) * WPF), c0(2 * F);
    CHK(hipMemcpy(sc.data(), d_segcnt, 4 * sc.size(), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(c0.data(), sa[0].counts, 4 * c0.size(), hipMemcpyDeviceToHost));
    int bad_seg = 0;
    for (uint32_t f = 0; f < F; f++)
    {
      uint32_t a = 0, b = 0;
      for (uint32_t w = 0; w < WPF; w++)
        a += sc[f * WPF + w] & 0xffffu, b += sc[f * WPF + w] >> 16;
      if (a != c0[2 * f] || b != c0[2 * f + 1])
        bad_seg++;
    }
    std::vector<uint32_t> kall(N), k0(c0[0]), ks;
    std::vector<float> qall(3 * static_cast<size_t>(N)), q0(3 * static_cast<size_t>(c0[1]));
    CHK(hipMemcpy(kall.data(), sa[2].keys, 4 * kall.size(), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(qall.data(), sa[2].extras, 4 * qall.size(), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(k0.data(), sa[0].keys, 4 * k0.size(), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(q0.data(), sa[0].extras, 4 * q0.size(), hipMemcpyDeviceToHost));
    std::vector<std::array<float, 3>> t0(q0.size() / 3), ts;
    std::memcpy(t0.data(), q0.data(), 4 * q0.size());
    for (uint32_t w = 0; w < WPF; w++)
    {
      const uint32_t seg = w * 64u * KEY1_PPT;
      for (uint32_t i = 0; i < (sc[w] & 0xffffu); i++)
        ks.push_back(kall[seg + i]);
      for (uint32_t i = 0; i < (sc[w] >> 16); i++)
        ts.push_back({qall[3 * (seg + i)], qall[3 * (seg + i) + 1], qall[3 * (seg + i) + 2]});
    }
    std::sort(ks.begin(), ks.end()), std::sort(k0.begin(), k0.end()), std::sort(ts.begin(), ts.end()), std::sort(t0.begin(), t0.end());
    if (ks != k0 || ts != t0)
      bad_seg++;
    std::printf("k_key1_seg: %d mismatches against k_key1\n", bad_seg);
  }
  // same results: counts, bounding boxes, and - as sorted lists - the codes and the fragile points of a few frames
  std::vector<FrameHdr> hh(2 * F);
  std::vector<uint32_t> hc[2];
  CHK(hipMemcpy(hh.data(), d_hdrs, sizeof(FrameHdr) * 2 * F, hipMemcpyDeviceToHost));
  for (int v = 0; v < 2; v++)
  {
    hc[v].resize(2 * F);
    CHK(hipMemcpy(hc[v].data(), sa[v].counts, sizeof(uint32_t) * 2 * F, hipMemcpyDeviceToHost));
  }
  int bad = 0;
  for (uint32_t f = 0; f < F; f++)
  {
    if (hc[0][2 * f] != hc[1][2 * f] || hc[0][2 * f + 1] != hc[1][2 * f + 1] || hh[f].n_in != hh[F + f].n_in || std::memcmp(hh[f].bb_min, hh[F + f].bb_min, 24) != 0)
      bad++;
  }
  for (uint32_t f : {0u, 1u, F - 1})
  {
    std::vector<uint32_t> k[2];
    std::vector<float> q[2];
    for (int v = 0; v < 2; v++)
    {
      k[v].resize(hc[v][2 * f]);
      q[v].resize(3 * static_cast<size_t>(hc[v][2 * f + 1]));
      CHK(hipMemcpy(k[v].data(), sa[v].keys + static_cast<size_t>(f) * N, 4 * k[v].size(), hipMemcpyDeviceToHost));
      CHK(hipMemcpy(q[v].data(), sa[v].extras + static_cast<size_t>(f) * N, 4 * q[v].size(), hipMemcpyDeviceToHost));
      std::sort(k[v].begin(), k[v].end());
      std::vector<std::array<float, 3>> t(q[v].size() / 3);
      std::memcpy(t.data(), q[v].data(), 4 * q[v].size());
      std::sort(t.begin(), t.end());
      std::memcpy(q[v].data(), t.data(), 4 * q[v].size());
    }
    if (k[0] != k[1] || q[0] != q[1])
      bad++;
    std::printf("frame %u: %zu codes, %zu fragile points\n", f, k[0].size(), q[0].size() / 3);
  }
  std::printf(bad ? "MISMATCH in %d checks\n" : "same results (%d mismatches)\n", bad);
  return bad ? 1 : 0;
}
