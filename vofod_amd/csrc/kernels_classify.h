// Device-side flood fill and confidence sums of the classification tail (K12, K13) for gfx950.
//
// VoxelMap::exploreToGround (voxel_map.cpp:402-488) answers "is this voxel connected to the ground through
// unknown voxels inside a Manhattan ball?".  Its result does not depend on the visiting order:
// "connected" iff some reachable voxel passes one of the exit tests, otherwise the set of reachable unknown
// voxels.  One wave per frame therefore runs the fill 64 voxels at a time from a LIFO work list (depth
// first in spirit, so an all-unknown neighbourhood reaches the ball's rim in ~R rounds), with the visited
// set as a bitset in LDS.  Jobs of one frame run in the reference's order because a "not connected" result
// rewrites the explored voxels to the frontier value (vofod_nodelet.cpp:1712-1715) and later fills see it;
// under VOFOD_SCAN_NO_MAP_UPDATE those writes go to a per-frame overlay bitset instead of the map.
// Afterwards the same wave evaluates extractDetections' uncertainty sum (vofod_nodelet.cpp:851-865).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace vc
{
using namespace vk;

constexpr int EX_MAX_R = 32;                                        // Manhattan radius handled on the device
constexpr int EX_SIDE = 2 * EX_MAX_R + 1;                            // 65
constexpr uint32_t EX_CELLS = EX_SIDE * EX_SIDE * EX_SIDE;           // 274 625 visited bits: 34 KB per frame in global memory (L2), all-zero
                                                                     // outside a fill.  Not in LDS: a fill's wave then fits a CU beside a frame
                                                                     // workgroup (153 KB of LDS) and the tail never keeps a CU from the next batch
constexpr uint32_t EX_WORDS = (EX_CELLS + 31) / 32;
constexpr uint32_t EX_MAX_JOBS = 1024;                               // explore jobs per frame handled on the device

struct ExploreJob
{
  uint32_t frame;
  uint32_t n_members;
  uint32_t member_off;  // into the int3 member voxel list
  int32_t R;            // max_explore_voxel_size (vofod_nodelet.cpp:1696)
  int32_t box_lo[3], box_hi[3];  // getSubmapCopy(aabb, 2) index box (voxel_map.cpp:550-559), inclusive
  uint32_t result_slot;
};

struct ExploreResult
{
  uint32_t floating;
  uint32_t overflow;  // work list exhausted (cannot happen for R <= EX_MAX_R)
  double conf_sum;    // sum over the sub-map of (1 - v/ray), cluster voxels counted as ray
};

struct ExploreParams
{
  float thr_unknown, thr_ground, frontier_value;
  double ray_score;
  int32_t no_update;
  uint32_t stack_cap;  // entries per frame in the global work list / explored list
};

// (Round 5: the fences of the flood fill are workgroup-scope.  They only order the wave's own traffic - every frame slot has its
// own work list, visited bits and overlay, and the words other lanes test are updated with atomics at L2.  Rounds 2-4 used
// __threadfence(): an agent-scope fence writes back and invalidates the L2 of the wave's XCD on this multi-die part - a few hundred
// of them per batch cost the frame kernels running beside the tail 14 % of their throughput: 967 k frames/s without the tail
// kernel, 830 k with it.)
// Memory ordering between the lanes of ONE wave (what __syncthreads() is in a 64-thread workgroup, where the compiler drops the
// s_barrier): the tail kernels run several such waves - one frame each, diverging freely - in one workgroup, so a real barrier
// would hang them.
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ uint32_t pack_rel(int dx, int dy, int dz) { return static_cast<uint32_t>((dx + 128) | ((dy + 128) << 8) | ((dz + 128) << 16)); }

// The jobs jb..je of frame slot `slot`, in order, by the calling wave; s_float: one byte per job, s_walk: 6 * 32 bytes, both in LDS
// and the wave's own.  k_explore below and the fused tail of the close-first path (kernels_tail.h) run this.
__device__ __forceinline__ void explore_frame(const ExploreParams& ep, const MapGeom& mg, const ExploreJob* __restrict__ jobs, const uint32_t jb, const uint32_t je, const int* __restrict__ members,
                                              float* __restrict__ map, unsigned long long* __restrict__ overlay_all, uint32_t* __restrict__ stack_all, uint32_t* __restrict__ explored_all,
                                              uint32_t* __restrict__ touched_all, uint32_t* __restrict__ ovl_list_all, uint32_t* __restrict__ ovl_count_all, ExploreResult* __restrict__ results,
                                              uint32_t* __restrict__ visited_all, const uint32_t slot, uint8_t* s_float, uint8_t* s_walk)
{
  const int lane = threadIdx.x & 63;  // (the calling wave: a 64-thread workgroup, or one wave of a wider one)
  if (jb == je)
    return;
  const uint64_t ovl_words = (mg.n + 63) >> 6;
  unsigned long long* overlay = overlay_all + static_cast<uint64_t>(slot) * ovl_words;
  uint32_t* stack = stack_all + static_cast<uint64_t>(slot) * ep.stack_cap;
  uint32_t* explored = explored_all + static_cast<uint64_t>(slot) * ep.stack_cap;
  uint32_t* ovl_list = ovl_list_all + static_cast<uint64_t>(slot) * ep.stack_cap;
  uint32_t* touched = touched_all + static_cast<uint64_t>(slot) * ep.stack_cap;
  uint32_t* visited = visited_all + static_cast<uint64_t>(slot) * EX_WORDS;  // clean on entry: every fill resets what it set

  auto map_read = [&](uint64_t li) -> float {
    // the overlay is updated with atomics (L2); read it there too so this CU's L1 cannot serve a stale word.  Both loads are
    // issued together (round 5: the map value used to be fetched only after the overlay word had come back - two round trips
    // in every link of the fills' dependent chains; the map word of a valid cell can always be read)
    const float mv = map[li];
    if (!ep.no_update)
      return mv;
    const unsigned long long ow = __hip_atomic_load(&overlay[li >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((ow >> (li & 63)) & 1ull) ? ep.frontier_value : mv;
  };

  // ---- phase 1: classify_cluster's flood fills, job by job (vofod_nodelet.cpp:1694-1718)
  for (uint32_t j = jb; j < je; j++)
  {
    const ExploreJob job = jobs[j];
    bool floating = true;
    uint32_t overflow = 0;
    const float max_voxel_dist = static_cast<float>(job.R);
    uint32_t m = 0;
    while (m < job.n_members && floating)
    {
      // Look at the next 64 start voxels at once: a member whose start voxel is neither ground nor unknown (typically
      // because an earlier member's fill has just turned it into a frontier) explores nothing and is skipped; the
      // first member that needs a decision is handled alone so that the reference's sequential semantics hold.
      {
        const uint32_t mm = m + lane;
        int kind = 0;  // 0 skip, 1 connected at once (border or ground), 2 needs a fill
        if (mm < job.n_members)
        {
          const int sx_ = members[3 * (job.member_off + mm)], sy_ = members[3 * (job.member_off + mm) + 1], sz_ = members[3 * (job.member_off + mm) + 2];
          if (sx_ <= 0 || sy_ <= 0 || sz_ <= 0 || sx_ >= mg.sx - 1 || sy_ >= mg.sy - 1 || sz_ >= mg.sz - 1)
            kind = 1;  // voxel_map.cpp:408-411
          else
          {
            const float sv = map_read((static_cast<uint64_t>(sz_) * mg.sy + sy_) * mg.sx + sx_);
            kind = sv > ep.thr_ground ? 1 : (sv > ep.thr_unknown ? 2 : 0);
          }
        }
        const unsigned long long need = __ballot(kind != 0);
        if (!need)
        {
          m += 64;
          continue;
        }
        const int first = __ffsll(static_cast<long long>(need)) - 1;
        m += first;
        if (__shfl(kind, first) == 1)
        {
          floating = false;
          break;
        }
      }
      const int ox = members[3 * (job.member_off + m)], oy = members[3 * (job.member_off + m) + 1], oz = members[3 * (job.member_off + m) + 2];
      // Fast path: the six axis-aligned walks from the start voxel are legal paths of the fill (6-neighbour moves,
      // Manhattan distance = step count).  If one of them runs over unknown voxels up to the rim (distance R-1) or
      // meets a ground voxel first, the reference's fill returns "connected" whatever else it would explore, so the
      // whole ball need not be searched.  All voxels of the six walks are fetched at once.
      if (job.R - 1 >= 1 && job.R - 1 <= 31)
      {
        const int len = job.R - 1;
        // (three voxels per lane; their loads go out together - the start voxel is an inner voxel of the map, a walk that leaves
        // the map reads the start voxel's word instead and is marked blocked)
        float wv[3];
        bool win[3];
        const uint64_t li0 = (static_cast<uint64_t>(oz) * mg.sy + oy) * mg.sx + ox;
#pragma unroll
        for (int u = 0; u < 3; u++)
        {
          const int e = lane + 64 * u, q = e >> 5, d = (e & 31) + 1;
          const int ax = ox + (q == 0 ? d : q == 3 ? -d : 0), ay = oy + (q == 1 ? d : q == 4 ? -d : 0), az = oz + (q == 2 ? d : q == 5 ? -d : 0);
          win[u] = d <= len && ax >= 0 && ax <= mg.sx - 1 && ay >= 0 && ay <= mg.sy - 1 && az >= 0 && az <= mg.sz - 1;
          wv[u] = map_read(win[u] ? (static_cast<uint64_t>(az) * mg.sy + ay) * mg.sx + ax : li0);
        }
#pragma unroll
        for (int u = 0; u < 3; u++)
          s_walk[lane + 64 * u] = !win[u] ? 0 : (wv[u] > ep.thr_ground ? 2 : (wv[u] > ep.thr_unknown ? 1 : 0));  // 0 blocked, 1 unknown, 2 ground
        wave_sync();
        bool walk_ok = false;
        if (lane < 6)
        {
          walk_ok = true;
          for (int d = 1; d <= len; d++)
          {
            const uint8_t st = s_walk[(lane << 5) + d - 1];
            if (st == 2)
              break;  // ground reached through unknown voxels
            if (st == 0)
            {
              walk_ok = false;
              break;
            }
          }
        }
        const bool any_walk = __ballot(walk_ok) != 0ull;
        wave_sync();
        if (any_walk)
        {
          floating = false;
          break;
        }
      }
      uint32_t n_stack = 1, n_expl = 0, n_touched = 1;  // wave-uniform
      if (lane == 0)
      {
        stack[0] = pack_rel(0, 0, 0);
        const uint32_t c = (EX_MAX_R * EX_SIDE + EX_MAX_R) * EX_SIDE + EX_MAX_R;
        atomicOr(&visited[c >> 5], 1u << (c & 31));
        touched[0] = c;
      }
      wave_sync();
      bool connected = false;
      while (n_stack > 0 && !connected)
      {
        const uint32_t take = min(n_stack, 64u);
        const uint32_t base = n_stack - take;
        n_stack = base;
        bool have = static_cast<uint32_t>(lane) < take;
        int dx = 0, dy = 0, dz = 0;
        float val = 0.0f;
        if (have)
        {
          const uint32_t e = stack[base + lane];
          dx = static_cast<int>(e & 255u) - 128;
          dy = static_cast<int>((e >> 8) & 255u) - 128;
          dz = static_cast<int>((e >> 16) & 255u) - 128;
          val = map_read((static_cast<uint64_t>(oz + dz) * mg.sy + (oy + dy)) * mg.sx + (ox + dx));
        }
        const bool ground = have && val > ep.thr_ground;           // :424
        const bool unknown = have && !ground && val > ep.thr_unknown;  // :426
        const int md = abs(dx) + abs(dy) + abs(dz);
        const bool rim = unknown && static_cast<float>(md) == max_voxel_dist - 1.0f;  // :430
        if (__ballot(ground || rim))
        {
          connected = true;
          break;
        }
        // explored_unknown (:428)
        const unsigned long long um = __ballot(unknown);
        if (unknown)
        {
          const uint32_t pos = n_expl + __popcll(um & ((1ull << lane) - 1ull));
          if (pos < ep.stack_cap)
            explored[pos] = pack_rel(dx, dy, dz);
        }
        n_expl += __popcll(um);
        // expansion (:437-477): six neighbours inside the map and the Manhattan ball, visited on push
        uint32_t mine[6];
        int n_mine = 0;
        if (unknown)
        {
          const int nb[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
          // the six test-and-set operations go out together (global memory: one round trip for all of them)
          uint32_t oldw[6], bitq[6];
          bool okq[6];
#pragma unroll
          for (int q = 0; q < 6; q++)
          {
            const int tx = dx + nb[q][0], ty = dy + nb[q][1], tz = dz + nb[q][2];
            const int ax = ox + tx, ay = oy + ty, az = oz + tz;
            const int tmd = abs(tx) + abs(ty) + abs(tz);
            okq[q] = !(ax < 0 || ax > mg.sx - 1 || ay < 0 || ay > mg.sy - 1 || az < 0 || az > mg.sz - 1) && static_cast<float>(tmd) <= max_voxel_dist;
            const uint32_t c = okq[q] ? ((tz + EX_MAX_R) * EX_SIDE + (ty + EX_MAX_R)) * EX_SIDE + (tx + EX_MAX_R) : 0u;
            bitq[q] = 1u << (c & 31);
            oldw[q] = okq[q] ? atomicOr(&visited[c >> 5], bitq[q]) : 0xffffffffu;
          }
#pragma unroll
          for (int q = 0; q < 6; q++)
            if (okq[q] && !(oldw[q] & bitq[q]))
              mine[n_mine++] = pack_rel(dx + nb[q][0], dy + nb[q][1], dz + nb[q][2]);
        }
        // wave-wide exclusive scan of the push counts
        uint32_t incl = static_cast<uint32_t>(n_mine);
#pragma unroll
        for (int s = 1; s < 64; s <<= 1)
        {
          const uint32_t t = __shfl_up(incl, s);
          if (lane >= s)
            incl += t;
        }
        const uint32_t total = __shfl(incl, 63);
        uint32_t at = n_stack + incl - n_mine;
        uint32_t tat = n_touched + incl - n_mine;
        for (int q = 0; q < n_mine; q++, at++, tat++)
        {
          if (at < ep.stack_cap)
            stack[at] = mine[q];
          else
            overflow = 1;
          if (tat < ep.stack_cap)
          {
            const int tx = static_cast<int>(mine[q] & 255u) - 128, ty = static_cast<int>((mine[q] >> 8) & 255u) - 128, tz = static_cast<int>((mine[q] >> 16) & 255u) - 128;
            touched[tat] = ((tz + EX_MAX_R) * EX_SIDE + (ty + EX_MAX_R)) * EX_SIDE + (tx + EX_MAX_R);
          }
        }
        n_stack = min(n_stack + total, ep.stack_cap);
        n_touched += total;
        wave_sync();  // stack / visited traffic of this round is complete before the next pop
      }
      // reset the visited bits this fill has set (only those: the bitset is 34 KB, a fill usually touches a few cells)
      wave_sync();
      if (n_touched <= ep.stack_cap)
      {
        for (uint32_t e = lane; e < n_touched; e += 64)
          __hip_atomic_store(&visited[touched[e] >> 5], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      else
        for (uint32_t w = lane; w < EX_WORDS; w += 64)
          __hip_atomic_store(&visited[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_block();
      wave_sync();
      if (connected)
      {
        floating = false;
        break;
      }
      // not connected: the explored unknown voxels become frontiers (:1712-1715)
      for (uint32_t e = lane; e < min(n_expl, ep.stack_cap); e += 64)
      {
        const uint32_t pk = explored[e];
        const int dx = static_cast<int>(pk & 255u) - 128, dy = static_cast<int>((pk >> 8) & 255u) - 128, dz = static_cast<int>((pk >> 16) & 255u) - 128;
        const uint64_t li = (static_cast<uint64_t>(oz + dz) * mg.sy + (oy + dy)) * mg.sx + (ox + dx);
        if (ep.no_update)
        {
          const unsigned long long bit = 1ull << (li & 63);
          const unsigned long long old = atomicOr(&overlay[li >> 6], bit);
          if (!(old & bit))
          {
            const uint32_t pos = atomicAdd(&ovl_count_all[slot], 1u);
            if (pos < ep.stack_cap)
              ovl_list[pos] = static_cast<uint32_t>(li >> 6);
          }
        }
        else
          map[li] = ep.frontier_value;
      }
      __threadfence_block();
      wave_sync();
      m++;
    }
    if (lane == 0)
    {
      s_float[j - jb] = floating ? 1 : 0;
      results[job.result_slot].floating = floating ? 1u : 0u;
      results[job.result_slot].overflow = overflow;
      results[job.result_slot].conf_sum = 0.0;
    }
  }
  __threadfence_block();
  wave_sync();

  // ---- phase 2: extractDetections' uncertainty sum for the floating clusters (vofod_nodelet.cpp:851-865)
  for (uint32_t j = jb; j < je; j++)
  {
    const ExploreJob job = jobs[j];
    if (!s_float[j - jb])
      continue;
    const int nx = job.box_hi[0] - job.box_lo[0] + 1, ny = job.box_hi[1] - job.box_lo[1] + 1, nz = job.box_hi[2] - job.box_lo[2] + 1;
    const uint32_t cells = static_cast<uint32_t>(nx) * ny * nz;
    double acc = 0.0;
    for (uint32_t c = lane; c < cells; c += 64)
    {
      const int x = job.box_lo[0] + static_cast<int>(c % nx), y = job.box_lo[1] + static_cast<int>((c / nx) % ny), z = job.box_lo[2] + static_cast<int>(c / (nx * ny));
      bool is_member = false;
      for (uint32_t m = 0; m < job.n_members; m++)
      {
        const int* o = &members[3 * (job.member_off + m)];
        is_member |= (o[0] == x && o[1] == y && o[2] == z);
      }
      const float v = is_member ? static_cast<float>(ep.ray_score) : map_read((static_cast<uint64_t>(z) * mg.sy + y) * mg.sx + x);
      acc += 1.0 - static_cast<double>(v) / ep.ray_score;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1)
      acc += __shfl_xor(acc, s);
    if (lane == 0)
      results[job.result_slot].conf_sum = acc;
  }
  // ---- leave the overlay clean for the next call
  if (ep.no_update)
  {
    wave_sync();
    const uint32_t n = __hip_atomic_load(&ovl_count_all[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n <= ep.stack_cap)
      for (uint32_t e = lane; e < n; e += 64)
        overlay[ovl_list[e]] = 0ull;
    else
      for (uint64_t w = lane; w < ovl_words; w += 64)
        overlay[w] = 0ull;
    wave_sync();
    if (lane == 0)
      ovl_count_all[slot] = 0;
  }
}

// one wave (64 threads) per frame with jobs; job_begin[f]..job_end[f] index that frame's jobs in order
__global__ __attribute__((amdgpu_waves_per_eu(8, 8))) __launch_bounds__(64) void k_explore(const ExploreParams ep, const MapGeom mg, const ExploreJob* __restrict__ jobs, const uint32_t* __restrict__ job_begin, const uint32_t* __restrict__ job_end,
                                                const int* __restrict__ members, float* __restrict__ map, unsigned long long* __restrict__ overlay_all,
                                                uint32_t* __restrict__ stack_all, uint32_t* __restrict__ explored_all, uint32_t* __restrict__ touched_all,
                                                uint32_t* __restrict__ ovl_list_all, uint32_t* __restrict__ ovl_count_all, ExploreResult* __restrict__ results, uint32_t* __restrict__ visited_all)
{
  __shared__ uint8_t s_float[EX_MAX_JOBS];
  __shared__ uint8_t s_walk[6 * 32];
  const uint32_t slot = blockIdx.x;
  explore_frame(ep, mg, jobs, job_begin[slot], job_end[slot], members, map, overlay_all, stack_all, explored_all, touched_all, ovl_list_all, ovl_count_all, results, visited_all, slot, s_float, s_walk);
}

}  // namespace vc
