// Device-side classification tail of a read-only batch (gfx950): classify_cluster's boxes and gates
// (vofod_nodelet.cpp:1648-1690) and the selection of the detections (extractDetections :834-879), so that a batch is one
// stream-ordered chain - k_frame_lds -> k_tail_prep -> k_explore -> k_tail_finish - and the host only reads the few
// detection records back.  Round 1 did this part on the host between two device round trips (the candidate tables came
// back over PCIe, the boxes were computed by host threads, the explore jobs went up again): with the device chain down to
// 0.65 ms per 256 frames the host tail (0.4-0.7 ms) had become the bottleneck.
//   k_tail_prep    one workgroup per frame: candidate clusters (far, small) from the cluster table, their members sorted by
//                  (cluster, voxel rank) with a bitonic sort in LDS, canonical cluster order (size descending, smallest member
//                  ascending: SURVEY H3), then one lane per cluster: pcl::MomentOfInertiaEstimation boxes (vt::boxes_of_n:
//                  the very function of the host tail), the three gates, the explore job (Manhattan radius, sub-map box,
//                  members' map voxels) for clusters that reach the flood fill;
//   k_explore      (kernels_classify.h) flood fills + uncertainty sums, one wave per frame, jobs in the reference's order;
//   k_tail_finish  one wave per frame: the clusters found floating become raw detection records in cluster order.
// A frame beyond a capacity here (members, clusters, jobs, detections, Manhattan radius) sets a flag: the host then runs
// its own tail for that batch (the round-1 path, still used for debug output).
#pragma once
#include <hip/hip_runtime.h>

#include "host_tail.h"
#include "kernels_classify.h"
#include "kernels_voxelize.h"

namespace vtd
{

#ifndef TAIL_WPB_DEF
#define TAIL_WPB_DEF 8
#endif
constexpr int TAIL_WPB = TAIL_WPB_DEF;  // frames (waves) per workgroup of k_tail_far
#if TAIL_WPB_DEF >= 16
#define TAIL_OCC_ATTR
#else
#define TAIL_OCC_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
using namespace vk;

constexpr int TAIL_STAGE = 512;  // candidate members per frame whose centres k_tail_far keeps in LDS (6 KB per wave)
constexpr int TP_THREADS = 256;
constexpr int TP_MAXM = TAIL_MAXM;  // candidate members per frame
constexpr int TP_MAXC = TAIL_MAXC;  // candidate clusters per frame (one lane each)
constexpr int TP_MAXD = 16;    // detections per frame read back

struct TailParams
{
  int32_t min_points;       // classification__min_points
  double max_distance;      // classification__max_distance
  double max_size;          // classification__max_size
  double max_explore;       // classification__max_explore_distance
  float voxel_size;
  int32_t latches;          // background_pts_sufficient && sure_background_sufficient (:1694)
};

struct TailCluster
{
  uint32_t root, n_members;
  float obb_center[3];
  int32_t job;  // result slot of the cluster's explore job, or -1
};

struct DetRaw
{
  uint32_t root, n_points;
  float center[3];
  uint32_t pad;
  double conf_sum;
};

struct FrameDets
{
  uint32_t n;         // detections of the frame (cluster order)
  uint32_t fallback;  // a capacity was exceeded: the host tail must redo this batch
  int32_t status;     // FrameHdr::status
  uint32_t n_jobs;
  DetRaw d[TP_MAXD];
};

enum : uint32_t { TAIL_FB_MEMBERS = 1u, TAIL_FB_CLUSTERS = 2u, TAIL_FB_RADIUS = 4u, TAIL_FB_DETS = 8u, TAIL_FB_EXPLORE = 16u };

__global__ __launch_bounds__(TP_THREADS) void k_tail_prep(const GridParams g, const FrameHdr* __restrict__ hdrs, const FrameArgs* __restrict__ args, const ClusterRec* __restrict__ table_all,
                                                         const CandMember* __restrict__ cand_all, VoxelArrays va_all, const MapGeom mg, const TailParams tp, vc::ExploreJob* __restrict__ jobs,
                                                         uint32_t* __restrict__ job_begin, uint32_t* __restrict__ job_end, int* __restrict__ members_out, TailCluster* __restrict__ tailc,
                                                         FrameDets* __restrict__ dets)
{
  __shared__ unsigned long long s_key[TP_MAXM];  // (cluster root << 32) | voxel rank, sorted ascending
  __shared__ ClusterRec s_rec[TP_MAXC];
  __shared__ uint32_t s_ord[TP_MAXC];
  __shared__ uint32_t s_nc;
  const uint32_t f = blockIdx.x;
  const FrameHdr h = hdrs[f];
  const int tid = threadIdx.x;
  FrameDets& out = dets[f];
  uint32_t fallback = 0;
  if (tid == 0)
  {
    s_nc = 0;
    job_begin[f] = f * TP_MAXC;
    job_end[f] = f * TP_MAXC;
  }
  __syncthreads();
  const uint32_t n_cand = h.status == VOFOD_OK ? h.n_cand : 0u;
  const uint32_t C = h.status == VOFOD_OK ? h.C : 0u;
  if (n_cand > TP_MAXM)
    fallback |= TAIL_FB_MEMBERS;
  // candidate clusters of the frame (any order in the table)
  const ClusterRec* table = table_all + static_cast<size_t>(f) * g.vox_cap;
  for (uint32_t c = tid; c < C; c += TP_THREADS)
  {
    const ClusterRec r = table[c];
    if (r.cand && !r.close)
    {
      const uint32_t pos = atomicAdd(&s_nc, 1u);
      if (pos < TP_MAXC)
        s_rec[pos] = r;
    }
  }
  // members: sort by (root, voxel rank)
  uint32_t np2 = 1;
  while (np2 < n_cand && np2 < TP_MAXM)
    np2 <<= 1;
  const CandMember* cands = cand_all + static_cast<size_t>(f) * g.vox_cap;
  if (!(fallback & TAIL_FB_MEMBERS))
    for (uint32_t i = tid; i < np2; i += TP_THREADS)
    {
      unsigned long long k = ~0ull;
      if (i < n_cand)
      {
        const CandMember cm = cands[i];
        k = (static_cast<unsigned long long>(cm.root) << 32) | cm.v;
      }
      s_key[i] = k;
    }
  __syncthreads();
  if (!(fallback & TAIL_FB_MEMBERS) && n_cand > 1)
    for (uint32_t k = 2; k <= np2; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1)
      {
        for (uint32_t i = tid; i < np2; i += TP_THREADS)
        {
          const uint32_t l = i ^ j;
          if (l > i)
          {
            const unsigned long long a = s_key[i], b = s_key[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up)
            {
              s_key[i] = b;
              s_key[l] = a;
            }
          }
        }
        __syncthreads();
      }
  const uint32_t nc_all = s_nc;
  if (nc_all > TP_MAXC)
    fallback |= TAIL_FB_CLUSTERS;
  const uint32_t nc = fallback ? 0u : nc_all;
  // canonical order: size descending, smallest member (= root) ascending; rank by counting (<= 64 clusters)
  if (tid < static_cast<int>(nc))
  {
    const ClusterRec me = s_rec[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nc; j++)
    {
      const ClusterRec o = s_rec[j];
      rank += (o.size > me.size || (o.size == me.size && o.root < me.root)) ? 1u : 0u;
    }
    s_ord[rank] = tid;
  }
  __syncthreads();
  if (tid >= 64)
    return;  // one wave goes on: a lane per cluster, in canonical order
  const int lane = tid;
  bool live = lane < static_cast<int>(nc);
  const VoxelArrays va = frame_voxels(va_all, f, g.vox_cap);
  const FrameArgs& a = args[f];
  bool wants_job = false;
  vc::ExploreJob job{};
  TailCluster tc{};
  tc.job = -1;
  uint32_t m_begin = 0, m_count = 0;
  if (live)
  {
    const ClusterRec rec = s_rec[s_ord[lane]];
    tc.root = rec.root;
    // the cluster's members: the run of keys with this root
    uint32_t lo = 0, hi = n_cand;
    const unsigned long long want = static_cast<unsigned long long>(rec.root) << 32;
    while (lo < hi)
    {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_key[mid] < want)
        lo = mid + 1;
      else
        hi = mid;
    }
    m_begin = lo;
    while (m_begin + m_count < n_cand && static_cast<uint32_t>(s_key[m_begin + m_count] >> 32) == rec.root)
      m_count++;
    tc.n_members = m_count;
    auto get = [&](size_t i, float p[3]) {
      const float4 q = va.pts[static_cast<uint32_t>(s_key[m_begin + i])];
      p[0] = q.x;
      p[1] = q.y;
      p[2] = q.z;
    };
    // classify_cluster :1648-1690: boxes and gates
    bool pass = m_count > 0 && static_cast<int>(m_count) >= tp.min_points;
    if (m_count > 0)
    {
      const vt::Boxes bx = vt::boxes_of_n(m_count, get);
      for (int q = 0; q < 3; q++)
        tc.obb_center[q] = bx.obb_center[q];
      if (pass)
      {
        const float d[3] = {a.tf[3] - bx.obb_center[0], a.tf[7] - bx.obb_center[1], a.tf[11] - bx.obb_center[2]};
        const double dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        pass = !(dist > tp.max_distance);
      }
      float obb_size = 0.0f;
      if (pass)
      {
        const float d[3] = {bx.obb_max[0] - bx.obb_min[0], bx.obb_max[1] - bx.obb_min[1], bx.obb_max[2] - bx.obb_min[2]};
        obb_size = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        pass = !(obb_size > tp.max_size);
      }
      if (pass && tp.latches)  // without the latches the cluster stays "unknown" (:1694, :1719-1722): no detection
      {
        wants_job = true;
        job.frame = f;
        job.n_members = m_count;
        job.member_off = f * TP_MAXM + m_begin;
        job.R = static_cast<int>((obb_size + tp.max_explore) / tp.voxel_size);  // :1696
        if (job.R > vc::EX_MAX_R || job.R < 0)
          fallback |= TAIL_FB_RADIUS;
        const int s3[3] = {mg.sx, mg.sy, mg.sz};
        for (int q = 0; q < 3; q++)  // getSubmapCopy(aabb, inflate 2) voxel_map.cpp:550-559
        {
          const int mn = static_cast<int>(floorf((bx.aabb_min[q] - mg.off[q]) * mg.vs_inv)), mx = static_cast<int>(floorf((bx.aabb_max[q] - mg.off[q]) * mg.vs_inv));
          job.box_lo[q] = min(max(mn - 2, 0), s3[q] - 1);
          job.box_hi[q] = min(max(mx + 2, 0), s3[q] - 1);
        }
        for (uint32_t i = 0; i < m_count; i++)
        {
          float p[3];
          get(i, p);
          int* o = members_out + 3 * static_cast<size_t>(job.member_off + i);
          for (int q = 0; q < 3; q++)
            o[q] = static_cast<int>(floorf((p[q] - mg.off[q]) * mg.vs_inv));
        }
      }
    }
  }
  // jobs in cluster order
  const unsigned long long jm = __ballot(wants_job);
  const uint32_t n_jobs = __popcll(jm);
  if (wants_job)
  {
    const uint32_t slot = f * TP_MAXC + __popcll(jm & ((1ull << lane) - 1ull));
    job.result_slot = slot;
    jobs[slot] = job;
    tc.job = static_cast<int32_t>(slot);
  }
  const unsigned long long fb_any = __ballot(fallback != 0u);
  uint32_t fb_all = fallback;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1)
    fb_all |= __shfl_xor(fb_all, s);
  (void)fb_any;
  if (live)
    tailc[f * TP_MAXC + lane] = tc;
  if (lane == 0)
  {
    out.n = 0;
    out.fallback = fb_all;
    out.status = h.status;
    out.n_jobs = fb_all ? 0u : n_jobs;
    if (!fb_all)
      job_end[f] = f * TP_MAXC + n_jobs;
    // clusters past nc are marked empty for k_tail_finish
  }
  if (!live && lane < TP_MAXC)
  {
    TailCluster e{};
    e.job = -1;
    tailc[f * TP_MAXC + lane] = e;
  }
}

// ---- the whole tail of a close-first frame in one kernel, one wave per frame, 256 bytes of LDS (round 4) --------------------------
// k_frame_lds_far leaves the candidate clusters at the head of the frame's cluster table in the canonical order and their
// members cluster by cluster with ascending rank: no sorting here, hence no LDS to speak of - the wave fits a CU beside a frame
// workgroup of the next batch (k_tail_prep's 11 KB had to wait for a CU to come free: 50 us alone, 150-200 us in company, and
// the three tail kernels of a batch, one after the other on the tail stream, had become the pace of the pipeline).
// Lane c = candidate cluster c: boxes + gates + explore job as k_tail_prep; then the wave runs the frame's flood fills
// (explore_frame: the code of k_explore) and writes the detection records as k_tail_finish.
// Round 5: TAIL_WPB waves (frames) per workgroup.  With one wave per workgroup the 256 tails of a batch were dealt out one by one,
// each to the next CU a frame workgroup had just left - and a CU that holds even one such wave has no room for the next frame
// workgroup (4 x 128 registers on every SIMD): the tails kept ~15 % of the CUs away from the frame kernels (967 k frames/s without
// the tail kernel, 819 k with it).  Whole workgroups of tails fill a few CUs instead.
// (five waves per SIMD = at most 96 registers: one of this kernel's waves then fits beside the four 104-register waves a frame
// workgroup keeps on every SIMD - 4 x 104 + 96 = 512 - instead of waiting for a CU without one)
__global__ TAIL_OCC_ATTR __launch_bounds__(64 * TAIL_WPB) void k_tail_far(const GridParams g, const FrameHdr* __restrict__ hdrs, const FrameArgs* __restrict__ args, const ClusterRec* __restrict__ table_all,
                                                const CandMember* __restrict__ cand_all, VoxelArrays va_all, const MapGeom mg, const TailParams tp, const vc::ExploreParams ep, vc::ExploreJob* __restrict__ jobs,
                                                int* __restrict__ members_out, float* __restrict__ map, unsigned long long* __restrict__ overlay_all, uint32_t* __restrict__ stack_all,
                                                uint32_t* __restrict__ explored_all, uint32_t* __restrict__ touched_all, uint32_t* __restrict__ ovl_list_all, uint32_t* __restrict__ ovl_count_all,
                                                vc::ExploreResult* __restrict__ results, uint32_t* __restrict__ visited_all, FrameDets* __restrict__ dets, FrameDets* __restrict__ hout,
                                                TailCluster* __restrict__ tailc, unsigned long long* __restrict__ prof)
{
  __shared__ uint8_t s_float_all[TAIL_WPB][TP_MAXC];
  __shared__ uint8_t s_walk_all[TAIL_WPB][6 * 32];
  __shared__ float s_pts_all[TAIL_WPB][TAIL_STAGE][3];
  const uint32_t f = blockIdx.x * TAIL_WPB + (threadIdx.x >> 6);  // one wave per frame, TAIL_WPB frames per workgroup
  if (f >= g.n_frames)
    return;
  uint8_t* s_float = s_float_all[threadIdx.x >> 6];
  uint8_t* s_walk = s_walk_all[threadIdx.x >> 6];
  const FrameHdr h = hdrs[f];
  const int lane = threadIdx.x & 63;
  // (diagnostics, VOFOD_LDS_PROF=2: slots 22 / 23 of the frame's stamps - start, then boxes | explore | end as 20-bit offsets in 10 ns)
  const unsigned long long tp0 = prof ? wall_clock64() : 0ull;
  unsigned long long tp1 = 0ull, tp2 = 0ull;
  FrameDets& out = dets[f];
  uint32_t fallback = 0;
  const bool ok = h.status == VOFOD_OK;
  const uint32_t n_cand = ok ? h.n_cand : 0u;
  const uint32_t nc_all = ok ? h.n_cand_clusters : 0u;
  if (n_cand > static_cast<uint32_t>(TP_MAXM))
    fallback |= TAIL_FB_MEMBERS;
  if (nc_all > static_cast<uint32_t>(TP_MAXC))
    fallback |= TAIL_FB_CLUSTERS;
  const uint32_t nc = fallback ? 0u : nc_all;
  const bool live = lane < static_cast<int>(nc);
  const ClusterRec* table = table_all + static_cast<size_t>(f) * g.vox_cap;
  const CandMember* cands = cand_all + static_cast<size_t>(f) * g.vox_cap;
  const VoxelArrays va = frame_voxels(va_all, f, g.vox_cap);
  const FrameArgs& a = args[f];
  ClusterRec rec{};
  if (live)
    rec = table[lane];
  // the cluster's members: the run behind those of the clusters in front (a candidate's members are all its voxels)
  uint32_t m_count = live ? rec.size : 0u, m_begin = 0;
  {
    uint32_t incl = m_count;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
    {
      const uint32_t t = __shfl_up(incl, s);
      if (lane >= s)
        incl += t;
    }
    m_begin = incl - m_count;
  }
  // Round 5: the candidates' member centres are fetched ONCE, by all 64 lanes together, into the wave's own LDS; boxes_of_n walks
  // a cluster's members three times (mean, covariance, OBB extents), one lane per cluster, and every visit was two dependent
  // global loads (member list -> voxel record): ~34 us per frame of nothing but latency.  Members beyond the staging area (a
  // frame with more than TAIL_STAGE candidate members) are read from global memory as before - same values either way.
  float (*s_pts)[3] = s_pts_all[threadIdx.x >> 6];
  {
    const uint32_t total = min(static_cast<uint32_t>(__shfl(m_begin + m_count, 63)), static_cast<uint32_t>(TAIL_STAGE));
    for (uint32_t i = lane; i < total; i += 64)
    {
      const float4 q = va.pts[cands[i].v];
      s_pts[i][0] = q.x;
      s_pts[i][1] = q.y;
      s_pts[i][2] = q.z;
    }
    vc::wave_sync();
  }
  bool wants_job = false;
  vc::ExploreJob job{};
  TailCluster tc{};
  tc.job = -1;
  if (live)
  {
    tc.root = rec.root;
    tc.n_members = m_count;
    auto get = [&](size_t i, float p[3]) {
      const uint32_t k = m_begin + static_cast<uint32_t>(i);
      if (k < static_cast<uint32_t>(TAIL_STAGE))
      {
        p[0] = s_pts[k][0];
        p[1] = s_pts[k][1];
        p[2] = s_pts[k][2];
        return;
      }
      const float4 q = va.pts[cands[k].v];
      p[0] = q.x;
      p[1] = q.y;
      p[2] = q.z;
    };
    // classify_cluster :1648-1690: boxes and gates
    bool pass = m_count > 0 && static_cast<int>(m_count) >= tp.min_points;
    if (m_count > 0)
    {
      const vt::Boxes bx = vt::boxes_of_n(m_count, get);
      for (int q = 0; q < 3; q++)
        tc.obb_center[q] = bx.obb_center[q];
      if (pass)
      {
        const float d[3] = {a.tf[3] - bx.obb_center[0], a.tf[7] - bx.obb_center[1], a.tf[11] - bx.obb_center[2]};
        const double dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        pass = !(dist > tp.max_distance);
      }
      float obb_size = 0.0f;
      if (pass)
      {
        const float d[3] = {bx.obb_max[0] - bx.obb_min[0], bx.obb_max[1] - bx.obb_min[1], bx.obb_max[2] - bx.obb_min[2]};
        obb_size = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        pass = !(obb_size > tp.max_size);
      }
      if (pass && tp.latches)  // without the latches the cluster stays "unknown" (:1694, :1719-1722): no detection
      {
        wants_job = true;
        job.frame = f;
        job.n_members = m_count;
        job.member_off = f * TP_MAXM + m_begin;
        job.R = static_cast<int>((obb_size + tp.max_explore) / tp.voxel_size);  // :1696
        if (job.R > vc::EX_MAX_R || job.R < 0)
          fallback |= TAIL_FB_RADIUS;
        const int s3[3] = {mg.sx, mg.sy, mg.sz};
        for (int q = 0; q < 3; q++)  // getSubmapCopy(aabb, inflate 2) voxel_map.cpp:550-559
        {
          const int mn = static_cast<int>(floorf((bx.aabb_min[q] - mg.off[q]) * mg.vs_inv)), mx = static_cast<int>(floorf((bx.aabb_max[q] - mg.off[q]) * mg.vs_inv));
          job.box_lo[q] = min(max(mn - 2, 0), s3[q] - 1);
          job.box_hi[q] = min(max(mx + 2, 0), s3[q] - 1);
        }
        for (uint32_t i = 0; i < m_count; i++)
        {
          float p[3];
          get(i, p);
          int* o = members_out + 3 * static_cast<size_t>(job.member_off + i);
          for (int q = 0; q < 3; q++)
            o[q] = static_cast<int>(floorf((p[q] - mg.off[q]) * mg.vs_inv));
        }
      }
    }
  }
  // jobs in cluster order
  const unsigned long long jm = __ballot(wants_job);
  uint32_t n_jobs = __popcll(jm);
  const uint32_t jb = f * TP_MAXC;
  if (wants_job)
  {
    const uint32_t slot = jb + __popcll(jm & ((1ull << lane) - 1ull));
    job.result_slot = slot;
    jobs[slot] = job;
    tc.job = static_cast<int32_t>(slot);
  }
  if (tailc)  // (a map-updating scan: the host rebuilds the detections from these when the record slots overflow)
  {
    if (!live)
    {
      tc = TailCluster{};
      tc.job = -1;
    }
    tailc[f * TP_MAXC + lane] = tc;
  }
  uint32_t fb_all = fallback;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1)
    fb_all |= __shfl_xor(fb_all, s);
  if (fb_all)
  {
    if (lane == 0)
    {
      out.n = 0;
      out.fallback = fb_all;
      out.status = h.status;
      out.n_jobs = 0;
      if (hout)
      {
        hout[f].n = 0;
        hout[f].fallback = fb_all;
        hout[f].status = h.status;
        hout[f].n_jobs = 0;
      }
    }
    return;
  }
  __threadfence_block();
  vc::wave_sync();  // the jobs and their members' map voxels are written: the wave reads them back
  if (prof)
    tp1 = wall_clock64();
  vc::explore_frame(ep, mg, jobs, jb, jb + n_jobs, members_out, map, overlay_all, stack_all, explored_all, touched_all, ovl_list_all, ovl_count_all, results, visited_all, f, s_float, s_walk);
  __threadfence_block();
  vc::wave_sync();
  if (prof)
    tp2 = wall_clock64();
  // the floating clusters (extractDetections :843-846) in cluster order, as k_tail_finish
  bool det = false, bad = false;
  double conf = 0.0;
  if (tc.job >= 0)
  {
    const vc::ExploreResult r = results[tc.job];
    det = r.floating != 0u;
    bad = r.overflow != 0u;
    conf = r.conf_sum;
  }
  const unsigned long long dm = __ballot(det), bm = __ballot(bad);
  const uint32_t n = __popcll(dm);
  if (det)
  {
    const uint32_t pos = __popcll(dm & ((1ull << lane) - 1ull));
    if (pos < TP_MAXD)
    {
      DetRaw d;
      d.root = tc.root;
      d.n_points = tc.n_members;
      for (int q = 0; q < 3; q++)
        d.center[q] = tc.obb_center[q];
      d.pad = 0;
      d.conf_sum = conf;
      out.d[pos] = d;
      if (hout)
        hout[f].d[pos] = d;
    }
  }
  if (lane == 0)
  {
    uint32_t fb = 0;
    if (n > TP_MAXD)
      fb |= TAIL_FB_DETS;
    if (bm)
      fb |= TAIL_FB_EXPLORE;
    out.n = n;
    out.fallback = fb;
    out.status = h.status;
    out.n_jobs = n_jobs;
    if (hout)
    {
      hout[f].n = n;
      hout[f].fallback = fb;
      hout[f].status = h.status;
      hout[f].n_jobs = n_jobs;
    }
    if (prof)
    {
      const unsigned long long te = wall_clock64();
      auto d20 = [&](unsigned long long t) { return (t - tp0) > 0xfffffull ? 0xfffffull : (t - tp0); };
      prof[static_cast<size_t>(f) * 32 + 22] = tp0;
      prof[static_cast<size_t>(f) * 32 + 23] = d20(tp1) | (d20(tp2) << 20) | (d20(te) << 40) | (static_cast<unsigned long long>(n_jobs > 15u ? 15u : n_jobs) << 60);
    }
  }
}

// one wave per frame: the floating clusters (extractDetections :843-846) in cluster order.  The frame's record is written
// twice: to the workspace (device memory, read by the capacity fall-back) and straight into the caller-visible pinned host
// slot `hout` (135 KB per 256 frames over PCIe).  No copy command follows the tail: a D2H hipMemcpyAsync queued behind kernels
// occupies an SDMA engine while it waits, the runtime then brings up a further engine for the next batch's copy, and each
// first use of an engine blocked vofod_batch_submit for 6-7 ms (three times within the first twenty batches of a process).
__global__ __launch_bounds__(64) void k_tail_finish(const TailCluster* __restrict__ tailc, const vc::ExploreResult* __restrict__ results, FrameDets* __restrict__ dets, FrameDets* __restrict__ hout)
{
  const uint32_t f = blockIdx.x;
  const int lane = threadIdx.x;
  FrameDets& out = dets[f];
  const uint32_t fb_in = out.fallback;
  if (fb_in)
  {
    if (hout && lane == 0)
    {
      hout[f].n = 0;
      hout[f].fallback = fb_in;
      hout[f].status = out.status;
      hout[f].n_jobs = out.n_jobs;
    }
    return;
  }
  const TailCluster tc = tailc[f * TP_MAXC + lane];
  bool det = false, bad = false;
  double conf = 0.0;
  if (tc.job >= 0)
  {
    const vc::ExploreResult r = results[tc.job];
    det = r.floating != 0u;
    bad = r.overflow != 0u;
    conf = r.conf_sum;
  }
  const unsigned long long dm = __ballot(det), bm = __ballot(bad);
  const uint32_t n = __popcll(dm);
  if (det)
  {
    const uint32_t pos = __popcll(dm & ((1ull << lane) - 1ull));
    if (pos < TP_MAXD)
    {
      DetRaw d;
      d.root = tc.root;
      d.n_points = tc.n_members;
      for (int q = 0; q < 3; q++)
        d.center[q] = tc.obb_center[q];
      d.pad = 0;
      d.conf_sum = conf;
      out.d[pos] = d;
      if (hout)
        hout[f].d[pos] = d;
    }
  }
  if (lane == 0)
  {
    uint32_t fb = 0;
    if (n > TP_MAXD)
      fb |= TAIL_FB_DETS;
    if (bm)
      fb |= TAIL_FB_EXPLORE;
    out.n = n;
    out.fallback = fb;
    if (hout)
    {
      hout[f].n = n;
      hout[f].fallback = fb;
      hout[f].status = out.status;
      hout[f].n_jobs = out.n_jobs;
    }
  }
}

}  // namespace vtd
