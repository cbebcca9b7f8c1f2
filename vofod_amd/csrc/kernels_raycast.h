// Ray-cast map update (K14, K15) and separated-background-cluster removal (K16, K17, K6') kernels.
//
// raycast_cloud (vofod_nodelet.cpp:1397-1605): one lane per LiDAR ray walks the voxel map with the
// Amanatides-Woo DDA of VoxelMap::forEachRay (voxel_map.cpp:229-263) and adds the in-voxel path length
// to the raycast map with hardware float atomics; a single fused streaming pass then applies the
// exponential pull of un-flagged traversed voxels towards scores/ray and clears flags + raycast map.
//
// updateSeparatedBGClusters (vofod_nodelet.cpp:1126-1277): thresholded voxels are enumerated in the
// reference's x-outer/z-inner order through a transposed occupancy bitmap, voxelised with the counted
// grid (positional count quirk), clustered, and unsure clusters are erased with an
// order-independent atomic compare-and-swap application of m <- w1*m + w2*ray per (voxel, offset) pair.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_voxelize.h"

namespace vr
{
using namespace vk;

struct RayParams
{
  float R[9];
  float origin[3];
  float max_dist, min_intensity, voxel_size;
  uint32_t n;
};

__device__ __forceinline__ int c2i(float x, float off, float inv) { return static_cast<int>(floorf(__fmul_rn(__fsub_rn(x, off), inv))); }

#ifndef RAY_ACC
#define RAY_ACC 0  // diagnostics only (results wrong): 1 = the walk without the accumulation, 2 = plain stores instead of atomics
#endif

// One DDA walk per lane: the state of forEachRay (voxel_map.cpp:229-263) in a form without branches.
//   tmax / tdelta as in the reference (the per-ray sequence tmax += tdelta is kept: same floats, same voxel sequence);
//   rem[a]  = steps left on axis a before the walk would leave the map (cur[a] == last[a] of the reference <=> rem[a] == 0);
//   lin     = linear index of the current voxel, moved by lstep[a] = step[a] * stride[a].
struct RayWalk
{
  float tmax[3], tdelta[3], prev, length;
  int rem[3], lstep[3];
  uint32_t lin;
  bool active;
};

// What bounds it (round 5, profiles/r05_raycast_variants.txt, one OS1-128 scan at 0.25 m): the float atomics.  The same kernel
// without the accumulation: 78.5 us; with plain stores to the same addresses: 83.7 us; with the atomics: 218-222 us.  Tried and
// not kept: two / four rays per lane for ILP (218.0 / 241.1 us against 219.8), and dealing the rays to the XCDs by azimuth sector
// so that all rays through one wedge of the map are walked on one XCD and its L2 keeps the wedge's lines (221.7 us, 1 164 us at
// OS2-128 x 2048 / 0.1 m against 1 168) - neither where the rays sit nor which L2 they go through changes what ~4 M single-float
// read-modify-writes per scan cost.  The run merging below stays: without it there would be several times as many.
__global__ __launch_bounds__(256) void k_raycast(const RayParams rp, const MapGeom mg, const char* __restrict__ intensity, const char* __restrict__ range, uint64_t stride,
                                                 const float* __restrict__ lut_dirs, const float* __restrict__ lut_offs, const uint8_t* __restrict__ mask,
                                                 float* __restrict__ ray, uint32_t* __restrict__ any_hit)
{
  constexpr int RPL = 1;
  const int lane = threadIdx.x & 63;
  RayWalk w[RPL];
#pragma unroll
  for (int k = 0; k < RPL; k++)
  {
    const uint32_t idx_raw = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = idx_raw < rp.n;
    const uint32_t idx = alive ? idx_raw : 0u;
    const float inten = *reinterpret_cast<const float*>(intensity + static_cast<uint64_t>(idx) * stride);
    const uint32_t rng = *reinterpret_cast<const uint32_t*>(range + static_cast<uint64_t>(idx) * stride);
    if (inten < rp.min_intensity || (!mask[idx] && rng == 0))  // vofod_nodelet.cpp:1449
      alive = false;
    float dir[3], start[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
    {
      const float* R = &rp.R[3 * r];
      dir[r] = __fadd_rn(__fadd_rn(__fmul_rn(R[0], lut_dirs[3 * idx]), __fmul_rn(R[1], lut_dirs[3 * idx + 1])), __fmul_rn(R[2], lut_dirs[3 * idx + 2]));
      start[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(R[0], lut_offs[3 * idx]), __fmul_rn(R[1], lut_offs[3 * idx + 1])), __fmul_rn(R[2], lut_offs[3 * idx + 2])), rp.origin[r]);
    }
    const float ray_dist = __fmul_rn(0.001f, static_cast<float>(rng));                                             // :1455-1456
    w[k].length = ray_dist == 0.0f ? rp.max_dist : fminf(__fsub_rn(ray_dist, rp.voxel_size), rp.max_dist);          // :1457
    const int cur[3] = {c2i(start[0], mg.off[0], mg.vs_inv), c2i(start[1], mg.off[1], mg.vs_inv), c2i(start[2], mg.off[2], mg.vs_inv)};
    const int lim[3] = {mg.sx, mg.sy, mg.sz};
    if (cur[0] < 0 || cur[0] >= lim[0] || cur[1] < 0 || cur[1] >= lim[1] || cur[2] < 0 || cur[2] >= lim[2])  // :1482
      alive = false;
    // forEachRay voxel_map.cpp:229-263
    const float half = mg.vs / 2.0f;
    const int lstride[3] = {1, mg.sx, mg.sx * mg.sy};
#pragma unroll
    for (int a = 0; a < 3; a++)
    {
      const float absdir = fabsf(dir[a]);
      const int step = (dir[a] > 0.0f) - (dir[a] < 0.0f);
      w[k].tdelta[a] = __fmul_rn(__fdiv_rn(1.0f, absdir), mg.vs);
      const float ctr = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(cur[a]), 0.5f), mg.vs), mg.off[a]);
      const float ctr_offset = __fsub_rn(ctr, start[a]);
      w[k].tmax[a] = __fdiv_rn(__fadd_rn(half, __fmul_rn(static_cast<float>(step), ctr_offset)), absdir);
      w[k].rem[a] = step > 0 ? lim[a] - 1 - cur[a] : cur[a];  // (last[a] = step > 0 ? lim - 1 : 0, :246)
      w[k].lstep[a] = step * lstride[a];
    }
    w[k].lin = alive ? static_cast<uint32_t>((static_cast<uint64_t>(cur[2]) * mg.sy + cur[1]) * mg.sx + cur[0]) : 0u;
    w[k].prev = 0.0f;
    w[k].active = alive && 0.0f < w[k].length;
  }
  // Neighbouring lanes are neighbouring azimuth columns of one ring: their walks visit almost the same voxels in
  // almost the same order, so per DDA step the wave merges runs of lanes that sit in the same voxel (segmented
  // sum) and issues one float atomic per run instead of one per lane.  The loop is kept wave-uniform.
  bool any = false;
  while (true)
  {
    bool go = false;
#pragma unroll
    for (int k = 0; k < RPL; k++)
      go |= w[k].active;
    if (!__ballot(go))
      break;
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
      RayWalk& r = w[k];
      // the axis of the smallest tmax, first minimum on ties (Eigen's minCoeff, voxel_map.cpp:252)
      const bool s1 = r.tmax[1] < r.tmax[0];
      const float m01 = s1 ? r.tmax[1] : r.tmax[0];
      const bool s2 = r.tmax[2] < m01;
      const float dist = s2 ? r.tmax[2] : m01;
      float dd = __fsub_rn(fminf(dist, r.length), r.prev);
      dd = r.active ? dd : 0.0f;
      const uint32_t key = dd != 0.0f ? r.lin : 0xffffffffu;
      const int remi = s2 ? r.rem[2] : (s1 ? r.rem[1] : r.rem[0]);
      const bool adv = r.active & (remi != 0);
      const bool a2 = adv & s2, a1 = adv & s1 & !s2, a0 = adv & !s1 & !s2;
      r.tmax[0] = a0 ? __fadd_rn(r.tmax[0], r.tdelta[0]) : r.tmax[0];
      r.tmax[1] = a1 ? __fadd_rn(r.tmax[1], r.tdelta[1]) : r.tmax[1];
      r.tmax[2] = a2 ? __fadd_rn(r.tmax[2], r.tdelta[2]) : r.tmax[2];
      r.rem[0] -= a0 ? 1 : 0;
      r.rem[1] -= a1 ? 1 : 0;
      r.rem[2] -= a2 ? 1 : 0;
      r.lin += static_cast<uint32_t>(a0 ? r.lstep[0] : (a1 ? r.lstep[1] : (a2 ? r.lstep[2] : 0)));
      r.prev = r.active ? dist : r.prev;
      r.active = adv & (dist < r.length);
      // Runs of lanes in one voxel: segmented inclusive sum with DPP moves (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: vector ALU
      // only, no LDS crossbar).  The last lane of a run holds its total and issues the atomic.
      const uint32_t kprev = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(~key), static_cast<int>(key), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
      const bool head = lane == 0 || kprev != key;
      uint32_t f = head ? 1u : 0u;
      float run = dd;
      auto segstep = [&](auto ctrl_tag, auto mask_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
        const float t = __uint_as_float(dpp_mov0<CTRL, MASK>(__float_as_uint(run)));
        const uint32_t ft = dpp_mov0<CTRL, MASK>(f);
        run = f ? run : __fadd_rn(run, t);
        f |= ft;
      };
      segstep(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
      segstep(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
      segstep(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
      segstep(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
      segstep(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
      segstep(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
      const unsigned long long H = __ballot(head);
      const bool tail = lane == 63 || ((H >> (lane + 1)) & 1ull);
#if RAY_ACC == 1
      if (tail && key != 0xffffffffu && run == 12345.678f)
#else
      if (tail && key != 0xffffffffu)
#endif
      {
#if RAY_ACC == 2
        ray[key] = run;
#elif RAY_ACC == 3
        atomicAdd(reinterpret_cast<unsigned int*>(ray) + key, __float_as_uint(run));  // (timing of an integer atomic on the same addresses)
#elif RAY_ACC == 4
        atomicAdd(reinterpret_cast<unsigned long long*>(ray) + (key >> 1), static_cast<unsigned long long>(__float_as_uint(run)));
#else
        unsafeAtomicAdd(&ray[key], run);
#endif
        any = true;
      }
    }
  }
  if (any)
    *any_hit = 1u;
}

// max of a non-negative float array (order-preserving on the raw bits), for the old update rule (:1542)
__global__ __launch_bounds__(256) void k_max_nonneg(const float* __restrict__ v, uint64_t n, uint32_t* out)
{
  uint32_t m = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    const float x = v[i];
    if (x > 0.0f)
      m = max(m, __float_as_uint(x));
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1)
    m = max(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0 && m)
    atomicMax(out, m);
}

struct SweepParams
{
  float its_diff;
  float ray_score;
  float weighting_factor;  // new rule: coef / (sqrt(3)*vs)          (:1555-1556)
  float weight;            // old rule: coef                          (:1578)
  float max_val;           // old rule normaliser                      (:1542)
  int32_t new_rule;
};

// K15: fused update sweep (:1557-1602).  Reads flags + raycast (+ map where a ray passed), writes map,
// clears flags and the raycast accumulator.  Stores are issued only where a value actually changes.
__global__ __launch_bounds__(256) void k_ray_sweep(const SweepParams sp, uint64_t n, float* __restrict__ map, float* __restrict__ flags, float* __restrict__ ray)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    const float flag = flags[i];
    const float r = ray[i];
    if (flag == 0.0f && r > 0.0f)
    {
      float w1;
      if (sp.new_rule)
      {
        const float n_int = __fmul_rn(sp.weighting_factor, r);
        w1 = static_cast<float>(exp2(static_cast<double>(__fmul_rn(-sp.its_diff, n_int))));  // std::pow(2, x) evaluates in double
      }
      else
      {
        const float norm_val = __fdiv_rn(r, sp.max_val);
        const float w_single = __fmul_rn(sp.weight, __fsqrt_rn(norm_val));
        w1 = fminf(fmaxf(powf(__fsub_rn(1.0f, w_single), sp.its_diff), 0.0f), 1.0f);
      }
      const float w2 = __fsub_rn(1.0f, w1);
      map[i] = __fadd_rn(__fmul_rn(w1, map[i]), __fmul_rn(w2, sp.ray_score));
    }
    if (flag != 0.0f)
      flags[i] = 0.0f;
    if (r != 0.0f)
      ray[i] = 0.0f;
  }
}

// ------------------------------------------------------------------ generic exclusive scan (u32)
constexpr int GS_EPT = 8;
constexpr int GS_EPB = 256 * GS_EPT;

__global__ __launch_bounds__(256) void k_gscan_a(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ bsum)
{
  const uint32_t base = blockIdx.x * GS_EPB + threadIdx.x * GS_EPT;
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < GS_EPT; k++)
    if (base + k < n)
      c += in[base + k];
  __shared__ uint32_t lds4[4];
  uint32_t total;
  block_excl_scan_256(c, lds4, &total);
  if (threadIdx.x == 0)
    bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_gscan_b(uint32_t* bsum, uint32_t nblk, uint32_t* total_out)
{
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblk; base += 1024)
  {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblk ? bsum[i] : 0;
    const uint32_t incl = wave_incl_scan(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63)
      wsum[wave] = incl;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; w++)
      off += wsum[w];
    if (i < nblk)
      bsum[i] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry_s = off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out)
    *total_out = carry_s;
}

// out has n+1 entries: out[i] = sum(in[0..i)), out[n] = total
__global__ __launch_bounds__(256) void k_gscan_c(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ out)
{
  const uint32_t base = blockIdx.x * GS_EPB + threadIdx.x * GS_EPT;
  uint32_t vals[GS_EPT];
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < GS_EPT; k++)
  {
    vals[k] = (base + k < n) ? in[base + k] : 0u;
    c += vals[k];
  }
  __shared__ uint32_t lds4[4];
  uint32_t total;
  uint32_t run = block_excl_scan_256(c, lds4, &total) + bsum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < GS_EPT; k++)
  {
    if (base + k < n)
      out[base + k] = run;
    run += vals[k];
    if (base + k + 1 == n)
      out[n] = run;
  }
}

// ------------------------------------------------------------------ sepclusters

// K16: voxelsAsVoxelPC (voxel_map.cpp:187-212) enumerates the thresholded voxels x-outer / y / z-inner.  One lane
// owns one (x,y) column; lanes of a wave are consecutive in x, so the occupancy word of a (y,z) row is fetched
// once per wave and every lane tests its own bit.  Pass 1 counts the set bits per column into x-major order,
// a scan turns the counts into each column's first position, pass 2 writes the points (x,y,z as floats, map value)
// and the "sure" flags the counted grid's positional count consumes.
__global__ __launch_bounds__(256) void k_col_count(const MapGeom mg, const unsigned long long* __restrict__ mapbits, uint32_t* __restrict__ colcount_t)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ncol = static_cast<uint32_t>(mg.sx) * mg.sy;
  if (t >= ncol)
    return;
  const uint32_t x = t % mg.sx, y = t / mg.sx;
  const uint64_t plane = static_cast<uint64_t>(mg.sx) * mg.sy;
  uint32_t c = 0;
  uint64_t li = t;
  for (int z = 0; z < mg.sz; z++, li += plane)
    c += static_cast<uint32_t>((mapbits[li >> 6] >> (li & 63)) & 1ull);
  colcount_t[x * mg.sy + y] = c;
}

// voxelsAsPC (voxel_map.cpp:157-183): the debug clouds of the nodelet (background: map > new_obstacles; sure air: !(map >
// frontiers), vofod_nodelet.cpp:999-1013) in the reference's order, x outer / y / z inner, as world coordinates + map value.
// Same column walk as above, on the float map itself: ((m > threshold) == greater_than) needs no occupancy image.
__global__ __launch_bounds__(256) void k_col_count_thr(const MapGeom mg, const float* __restrict__ map, float threshold, int greater_than, uint32_t* __restrict__ colcount_t)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ncol = static_cast<uint32_t>(mg.sx) * mg.sy;
  if (t >= ncol)
    return;
  const uint32_t x = t % mg.sx, y = t / mg.sx;
  const uint64_t plane = static_cast<uint64_t>(mg.sx) * mg.sy;
  uint32_t c = 0;
  uint64_t li = t;
  for (int z = 0; z < mg.sz; z++, li += plane)
    c += ((map[li] > threshold) == (greater_than != 0)) ? 1u : 0u;
  colcount_t[x * mg.sy + y] = c;
}

__global__ __launch_bounds__(256) void k_col_emit_xyzi(const MapGeom mg, const float* __restrict__ map, float threshold, int greater_than, const uint32_t* __restrict__ colbase_t,
                                                       uint32_t cap, float4* __restrict__ out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ncol = static_cast<uint32_t>(mg.sx) * mg.sy;
  if (t >= ncol)
    return;
  const uint32_t x = t % mg.sx, y = t / mg.sx;
  const uint64_t plane = static_cast<uint64_t>(mg.sx) * mg.sy;
  uint32_t pos = colbase_t[x * mg.sy + y];
  if (colbase_t[x * mg.sy + y + 1] == pos)
    return;  // empty column
  const float cx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(x), 0.5f), mg.vs), mg.off[0]);  // idxToCoord voxel_map.cpp:610-613
  const float cy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(y), 0.5f), mg.vs), mg.off[1]);
  uint64_t li = t;
  for (int z = 0; z < mg.sz; z++, li += plane)
  {
    const float m = map[li];
    if ((m > threshold) == (greater_than != 0))
    {
      if (pos < cap)
        out[pos] = make_float4(cx, cy, __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(z), 0.5f), mg.vs), mg.off[2]), m);
      pos++;
    }
  }
}

__global__ __launch_bounds__(256) void k_col_emit(const MapGeom mg, const float* __restrict__ map, const unsigned long long* __restrict__ mapbits,
                                                  const uint32_t* __restrict__ colbase_t, float thr_sure, float* __restrict__ px, float* __restrict__ py,
                                                  float* __restrict__ pz, float* __restrict__ pi, uint32_t* __restrict__ sure)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ncol = static_cast<uint32_t>(mg.sx) * mg.sy;
  if (t >= ncol)
    return;
  const uint32_t x = t % mg.sx, y = t / mg.sx;
  const uint64_t plane = static_cast<uint64_t>(mg.sx) * mg.sy;
  uint32_t pos = colbase_t[x * mg.sy + y];
  if (colbase_t[x * mg.sy + y + 1] == pos)
    return;  // empty column
  uint64_t li = t;
  for (int z = 0; z < mg.sz; z++, li += plane)
    if ((mapbits[li >> 6] >> (li & 63)) & 1ull)
    {
      const float m = map[li];
      px[pos] = static_cast<float>(x);
      py[pos] = static_cast<float>(y);
      pz[pos] = static_cast<float>(z);
      pi[pos] = m;
      sure[pos] = m > thr_sure ? 1u : 0u;
      pos++;
    }
}

// per-voxel point counts of the weighted emission -> u32 array
__global__ __launch_bounds__(256) void k_voxel_counts(const FrameHdr* hdr, const float4* __restrict__ pts, uint32_t* __restrict__ out)
{
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < hdr->V)
    out[v] = __float_as_uint(pts[v].w);
}

// K6': voxel_grid_counted.cpp:185-191: range_k = #{input positions p in [first_k, last_k): intensity_p > thr}
// with [first_k,last_k) the run of voxel k in the sorted index vector = exclusive prefix of the voxel sizes.
__global__ __launch_bounds__(256) void k_counted_range(const FrameHdr* hdr, const uint32_t* __restrict__ first, const uint32_t* __restrict__ sure_prefix, uint32_t n_points,
                                                       float4* __restrict__ pts)
{
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= hdr->V)
    return;
  const uint32_t a = min(first[v], n_points), b = min(first[v + 1], n_points);
  pts[v].w = __uint_as_float(sure_prefix[b] - sure_prefix[a]);
}

// intensity > threshold flags of an arbitrary strided column (stateless counted grid)
__global__ __launch_bounds__(256) void k_flag_over(const char* __restrict__ col, uint64_t stride, uint32_t n, float thr, uint32_t* __restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = *reinterpret_cast<const float*>(col + static_cast<uint64_t>(i) * stride) > thr ? 1u : 0u;
}

// sum of `range` per cluster (vofod_nodelet.cpp:1175-1183) + "any cluster sure" flag
__global__ __launch_bounds__(256) void k_cluster_sure(const FrameHdr* hdr, const float4* __restrict__ pts, const uint32_t* __restrict__ labels, uint32_t* __restrict__ n_sure)
{
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t key = 0xffffffffu, r = 0;
  if (v < hdr->V)
  {
    r = __float_as_uint(pts[v].w);
    if (r)
      key = labels[v];
  }
  // consecutive voxels mostly share the cluster: one atomic per run of equal labels inside the wave
  int end;
  const bool head = run_heads(key, lane, end);
#pragma unroll
  for (int s2 = 1; s2 < 64; s2 <<= 1)
  {
    const uint32_t t = __shfl_down(r, s2);
    if (lane + s2 < end)
      r += t;
  }
  if (head && key != 0xffffffffu)
    atomicAdd(&n_sure[key], r);
}

__global__ __launch_bounds__(256) void k_any_sure(const FrameHdr* hdr, const uint32_t* __restrict__ labels, const uint32_t* __restrict__ n_sure, uint32_t min_sure, uint32_t* out)
{
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= hdr->V)
    return;
  if (labels[v] == v && n_sure[v] >= min_sure)
    out[0] = 1u;
  if (labels[v] == v)
    atomicMax(&out[1], n_sure[v]);
}

struct EraseParams
{
  float w1, w2, update_val;
  uint32_t min_sure;
  int32_t n_offsets;
};

// K17: erase stencil (vofod_nodelet.cpp:1244-1272).  Every (voxel of an unsure cluster, offset) pair applies
// m <- w1*m + w2*ray once; all applications are the same function, so applying them with an atomic
// compare-and-swap in any order reproduces the reference's sequential result bit for bit.
__global__ __launch_bounds__(256) void k_sep_erase(const EraseParams ep, const MapGeom mg, const FrameHdr* hdr, const float4* __restrict__ pts,
                                                   const uint32_t* __restrict__ labels, const uint32_t* __restrict__ n_sure, const int* __restrict__ offsets,
                                                   float* __restrict__ map)
{
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= hdr->V)
    return;
  if (n_sure[labels[v]] >= ep.min_sure)
    return;
  const float4 p = pts[v];
  const int pos[3] = {static_cast<int>(p.x), static_cast<int>(p.y), static_cast<int>(p.z)};  // cast<int>() truncates (:1252)
  // (the stencil - thousands of offsets at small voxel sizes - is dealt over blockIdx.y: the voxels of unsure clusters are few,
  // one thread per voxel walking the whole stencil left the chip empty: 2.4 ms at 0.1 m)
  for (int o = blockIdx.y; o < ep.n_offsets; o += gridDim.y)
  {
    const int x = pos[0] + offsets[3 * o], y = pos[1] + offsets[3 * o + 1], z = pos[2] + offsets[3 * o + 2];
    if (x < 0 || x >= mg.sx || y < 0 || y >= mg.sy || z < 0 || z >= mg.sz)
      continue;
    uint32_t* a = reinterpret_cast<uint32_t*>(&map[(static_cast<uint64_t>(z) * mg.sy + y) * mg.sx + x]);
    uint32_t old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true)
    {
      const float m = __uint_as_float(old);
      const float nm = __fadd_rn(__fmul_rn(ep.w1, m), __fmul_rn(ep.w2, ep.update_val));
      const uint32_t prev = atomicCAS(a, old, __float_as_uint(nm));
      if (prev == old)
        break;
      old = prev;
    }
  }
}

// device buffers owned by the sepclusters stage
struct SepState
{
  unsigned long long* d_tbits = nullptr;
  uint32_t* d_tpop = nullptr;     // word popcounts, then reused
  uint32_t* d_tprefix = nullptr;  // n_words + 1
  uint32_t* d_bsum = nullptr;
  size_t words_cap = 0;
  float *d_px = nullptr, *d_py = nullptr, *d_pz = nullptr, *d_pi = nullptr;
  uint32_t* d_sure = nullptr;      // flags per input position
  uint32_t* d_sure_pre = nullptr;  // P + 1
  uint32_t* d_vcnt = nullptr;      // per ds voxel
  uint32_t* d_first = nullptr;     // V' + 1
  uint32_t* d_nsure = nullptr;     // per root
  size_t pts_cap = 0;
  int* d_offsets = nullptr;
  uint32_t* d_small = nullptr;  // [0] P total, [1] any sure, [2] max sure, [3] V total
  uint32_t* h_small = nullptr;  // pinned
  uint32_t P = 0;
  GridParams g{};
};

}  // namespace vr
