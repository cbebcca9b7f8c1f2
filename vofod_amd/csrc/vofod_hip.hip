// libvofod_hip.so — the product: C++ host driver + hand-written gfx950 kernels behind include/vofod.h.
// One handle owns one HIP stream, the three voxel maps in HBM and a per-frame workspace; a scan (or a
// batch of independent scans, one grid.y slice each) is one stream-ordered chain of kernels followed by
// a single read-back of the small cluster table, after which the host runs the sequential
// classification tail (host_tail.h) on the few candidate clusters.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "host_tail.h"
#include "thread_pool.h"
#include "kernels_brick.h"
#include "kernels_brick_lds.h"
#include "kernels_classify.h"
#include "kernels_frame.h"
#include "kernels_cluster.h"
#include "kernels_raycast.h"
#include "kernels_slab.h"
#include "kernels_tail.h"
#include "kernels_far.h"
#include "kernels_voxelize.h"

using namespace vk;

namespace
{

constexpr uint32_t SPEC_C = 192;   // cluster records read back speculatively with the header
constexpr uint32_t SPEC_M = 768;  // candidate members read back speculatively

struct CandMemberX
{
  uint32_t root, v;
  float x, y, z;
  uint32_t count;
};

struct PackedFrame
{
  FrameHdr hdr;
  uint32_t pad[32 - sizeof(FrameHdr) / 4];
  ClusterRec table[SPEC_C];
  CandMemberX members[SPEC_M];
};
static_assert(sizeof(FrameHdr) <= 128, "FrameHdr grew past its slot");

__global__ void k_pack(const GridParams g, const FrameHdr* hdrs, const ClusterRec* table_all, const CandMember* cand_all, VoxelArrays va_all, PackedFrame* out)
{
  uint32_t FRAME, BX, GX;
  if (!frame_block(g, FRAME, BX, GX))
    return;
  (void)GX;
  const uint32_t f = FRAME;
  const FrameHdr h = hdrs[f];
  PackedFrame& o = out[f];
  const uint32_t t = BX * blockDim.x + threadIdx.x;
  if (t == 0)
    o.hdr = h;
  if (t < min(h.C, SPEC_C))
    o.table[t] = table_all[static_cast<size_t>(f) * g.vox_cap + t];
  if (t < min(h.n_cand, SPEC_M))
  {
    const CandMember cm = cand_all[static_cast<size_t>(f) * g.vox_cap + t];
    const float4 p = va_all.pts[static_cast<size_t>(f) * g.vox_cap + cm.v];
    CandMemberX x;
    x.root = cm.root;
    x.v = cm.v;
    x.x = p.x;
    x.y = p.y;
    x.z = p.z;
    x.count = __float_as_uint(p.w);
    o.members[t] = x;
  }
}

// Read-back of a batch nobody debugs: only what the classification tail consumes - the header, the candidate clusters'
// records (far, small enough: the others can neither be classified nor detected) and the candidate members.  6.9 KB per
// frame instead of the 26 KB speculative slot, copied on a stream of its own so that the next batch's chain does not wait for PCIe.
constexpr uint32_t LITE_C = 16;
constexpr uint32_t LITE_M = 256;
struct PackedLite
{
  FrameHdr hdr;
  uint32_t pad[32 - sizeof(FrameHdr) / 4];
  uint32_t n_recs, n_members, pad2[2];  // counts found on the device (beyond LITE_C / LITE_M: the host fetches the frame's full lists)
  ClusterRec recs[LITE_C];
  CandMemberX members[LITE_M];
};

__global__ __launch_bounds__(256) void k_pack_lite(const GridParams g, const FrameHdr* hdrs, const ClusterRec* table_all, const CandMember* cand_all, VoxelArrays va_all, PackedLite* out)
{
  __shared__ uint32_t s_cnt;
  const uint32_t f = blockIdx.x;
  const FrameHdr h = hdrs[f];
  PackedLite& o = out[f];
  if (threadIdx.x == 0)
  {
    o.hdr = h;
    s_cnt = 0;
  }
  __syncthreads();
  const ClusterRec* table = table_all + static_cast<size_t>(f) * g.vox_cap;
  for (uint32_t c = threadIdx.x; c < h.C; c += blockDim.x)
  {
    const ClusterRec r = table[c];
    if (r.cand && !r.close)
    {
      const uint32_t pos = atomicAdd(&s_cnt, 1u);
      if (pos < LITE_C)
        o.recs[pos] = r;
    }
  }
  static_assert(LITE_M <= 256, "one member per thread");
  if (threadIdx.x < min(h.n_cand, LITE_M))
  {
    const CandMember cm = cand_all[static_cast<size_t>(f) * g.vox_cap + threadIdx.x];
    const float4 p = va_all.pts[static_cast<size_t>(f) * g.vox_cap + cm.v];
    CandMemberX x;
    x.root = cm.root;
    x.v = cm.v;
    x.x = p.x;
    x.y = p.y;
    x.z = p.z;
    x.count = __float_as_uint(p.w);
    o.members[threadIdx.x] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    o.n_recs = s_cnt;
    o.n_members = h.n_cand;
  }
}

__global__ void k_gather_members(const GridParams g, uint32_t frame, uint32_t n, const CandMember* cand_all, VoxelArrays va_all, CandMemberX* out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n)
    return;
  const CandMember cm = cand_all[static_cast<size_t>(frame) * g.vox_cap + t];
  const float4 p = va_all.pts[static_cast<size_t>(frame) * g.vox_cap + cm.v];
  CandMemberX x;
  x.root = cm.root;
  x.v = cm.v;
  x.x = p.x;
  x.y = p.y;
  x.z = p.z;
  x.count = __float_as_uint(p.w);
  out[t] = x;
}

using clk = std::chrono::steady_clock;
inline double ms_since(const clk::time_point& t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

inline float ulp32(float x)
{
  x = std::fabs(x);
  return std::nextafterf(x, INFINITY) - x;
}

}  // namespace

// VOFOD_CALLTRACE=1 (diagnostics): every HIP call of the driver that keeps the host longer than 0.3 ms is reported
struct SlowCall
{
  const char* what;
  int line;
  std::chrono::steady_clock::time_point t0;
  static bool on()
  {
    static const bool v = std::getenv("VOFOD_CALLTRACE") != nullptr;
    return v;
  }
  SlowCall(const char* w, int l) : what(w), line(l)
  {
    if (on())
      t0 = std::chrono::steady_clock::now();
  }
  ~SlowCall()
  {
    if (!on())
      return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 0.3)
      std::fprintf(stderr, "[vofod calltrace] %.2f ms in %s (line %d)\n", ms, what, line);
  }
};

#define HIPCHK(expr)                                                                                         \
  do                                                                                                         \
  {                                                                                                          \
    SlowCall sc_(#expr, __LINE__);                                                                           \
    const hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                                    \
    {                                                                                                        \
      h->err = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"; \
      return VOFOD_ERR_DEVICE;                                                                               \
    }                                                                                                        \
  } while (0)

// ---- optional per-kernel timing with HIP events on the handle's stream (vofod_profile_*) ----------
struct Prof
{
  bool on = false;
  struct Rec
  {
    const char* name;
    hipEvent_t a, b;
  };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get()
  {
    if (!pool.empty())
    {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
};

#define KLAUNCH(h, kern, grid, block, ...)                                   \
  do                                                                         \
  {                                                                          \
    if ((h)->prof.on)                                                        \
    {                                                                        \
      Prof::Rec r_{#kern, (h)->prof.get(), (h)->prof.get()};                 \
      (void)hipEventRecord(r_.a, (h)->stream);                               \
      hipLaunchKernelGGL(kern, grid, block, 0, (h)->stream, __VA_ARGS__);    \
      (void)hipEventRecord(r_.b, (h)->stream);                               \
      (h)->prof.recs.push_back(r_);                                          \
    }                                                                        \
    else                                                                     \
    {                                                                        \
      SlowCall sc_(#kern, __LINE__);                                         \
      hipLaunchKernelGGL(kern, grid, block, 0, (h)->stream, __VA_ARGS__);    \
    }                                                                        \
  } while (0)

// (the same under another name in the profile: the instantiations of one kernel share their algorithmic name)
#define KLAUNCH_AS(h, label, kern, grid, block, ...)                        \
  do                                                                         \
  {                                                                          \
    if ((h)->prof.on)                                                        \
    {                                                                        \
      Prof::Rec r_{label, (h)->prof.get(), (h)->prof.get()};                 \
      (void)hipEventRecord(r_.a, (h)->stream);                               \
      hipLaunchKernelGGL(kern, grid, block, 0, (h)->stream, __VA_ARGS__);    \
      (void)hipEventRecord(r_.b, (h)->stream);                               \
      (h)->prof.recs.push_back(r_);                                          \
    }                                                                        \
    else                                                                     \
    {                                                                        \
      SlowCall sc_(label, __LINE__);                                         \
      hipLaunchKernelGGL(kern, grid, block, 0, (h)->stream, __VA_ARGS__);    \
    }                                                                        \
  } while (0)

// (the same with dynamic LDS)
#define KLAUNCH_LDS(h, kern, grid, block, lds, ...)                                   \
  do                                                                         \
  {                                                                          \
    if ((h)->prof.on)                                                        \
    {                                                                        \
      Prof::Rec r_{#kern, (h)->prof.get(), (h)->prof.get()};                 \
      (void)hipEventRecord(r_.a, (h)->stream);                               \
      hipLaunchKernelGGL(kern, grid, block, lds, (h)->stream, __VA_ARGS__);    \
      (void)hipEventRecord(r_.b, (h)->stream);                               \
      (h)->prof.recs.push_back(r_);                                          \
    }                                                                        \
    else                                                                     \
      hipLaunchKernelGGL(kern, grid, block, lds, (h)->stream, __VA_ARGS__);    \
  } while (0)

// VOFOD_<NAME>=0 switches a fast path off (README "Environment switches").  Read on every call, not cached: a dozen getenv per
// batch cost ~1 us, and the tests flip the fallbacks inside one process (a `static const` here made every switch stick to
// its first reading - and every in-process fallback test after the first call vacuous).
inline bool switch_off(const char* name)
{
  const char* v = std::getenv(name);
  return v && std::atoi(v) == 0;
}

struct Workspace
{
  uint32_t F = 0, pt_cap = 0, vox_cap = 0, words_cap = 0, nblk_cap = 0, bricks_cap = 0;
  BrickArrays ba{};
  unsigned long long* d_bconn = nullptr;  // per brick: connectivity mask over the forward stencil (transitive reduction)
  FrameArgs* d_args = nullptr;
  FrameHdr* d_hdrs = nullptr;
  unsigned long long* d_bitmaps = nullptr;
  uint32_t* d_wprefix = nullptr;
  uint32_t* d_blocksums = nullptr;
  VoxelArrays va{};
  uint32_t* d_labels = nullptr;
  ClusterRec* d_table = nullptr;
  CandMember* d_cand = nullptr;
  uint32_t* d_ptrank = nullptr;
  SlabArrays sa{};  // key list / extras of the LDS-slab voxelisation (keys share d_ptrank's storage)
  FrameScratch fs{};  // row tables of the brick-first frame kernel (kernels_frame.h)
  RefLattice ref_lattice{};  // single-pass input: the reference lattice the list's cells refer to (on = 0: brick codes from k_key2)
  bool frame_fused = false;  // k_key2 ran: launch_cluster runs k_frame_lds (voxel records + clustering in one kernel)
  float* d_stage = nullptr;  // F * pt_cap * 5 words: x, y, z, intensity, range of host-resident inputs
  char* d_stage_aos = nullptr;  // host-resident array-of-structs clouds (the nodelet's 48-byte ouster_ros::Point) cross the link as they are:
  size_t stage_aos_bytes = 0;   // F * aos_pitch bytes, allocated on first use; the kernels read x / y / z in place at the struct's stride
  size_t aos_pitch = 0;
  PackedFrame* d_packed = nullptr;
  PackedFrame* h_packed = nullptr;  // pinned
  PackedLite* d_lite = nullptr;
  PackedLite* h_lite = nullptr;  // pinned
  bool lite = false;  // the batch in this workspace was read back through the lite slots (no debug output asked for)
  bool mapbits_patched = false;  // k_finalize_far kept the map's occupancy image and counters up to date with this scan's update
  bool prof_deferred = false;  // VOFOD_LDS_PROF=2: the frame kernel's stamps of this batch are printed when it is collected
  uint32_t prof_slot0 = 0;
  bool in_packed = false;  // the batch's columns are packed, 16-byte aligned floats, a multiple of 4 points each: the frame kernel's input pass uses 16-byte loads
  bool far_ran = false;  // launch_cluster ran k_frame_lds_far: the cluster table and the member list are in the order k_tail_far reads
  int close_first = 0;  // k_frame_lds: 1 = cluster the far voxels only (read-only batches), 2 = the same with labels for the far-only debug view
  bool dtail = false;  // ... or its classification tail ran on the device (kernels_tail.h): only detection records come back
  vtd::TailCluster* d_tailc = nullptr;
  vtd::FrameDets* d_dets = nullptr;
  vtd::FrameDets* h_dets = nullptr;  // pinned
  vtd::FrameDets* h_dets_dev = nullptr;  // the same slots as the device sees them (k_tail_finish writes the records there)
  uint32_t* d_job_be = nullptr;      // [2][F]: first and one-past-last explore job of every frame
  // per-frame launch arguments in pinned host memory: their upload is a true asynchronous copy, so a batch is enqueued while the
  // previous chain still runs (a copy from pageable memory makes the submitting thread wait for the stream)
  struct PinnedArgs
  {
    FrameArgs* p = nullptr;
    size_t n = 0;
    FrameArgs& operator[](size_t i) { return p[i]; }
    const FrameArgs& operator[](size_t i) const { return p[i]; }
    FrameArgs* data() { return p; }
    hipError_t assign(size_t n_, const FrameArgs& v)
    {
      if (p)
        (void)hipHostFree(p);
      p = nullptr;
      n = n_;
      if (hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(FrameArgs) * std::max<size_t>(n, 1)); e != hipSuccess)
        return e;
      for (size_t i = 0; i < n; i++)
        p[i] = v;
      return hipSuccess;
    }
  } h_args;
  bool bricks_preset = false;  // k_emit already registered the voxels in their bricks (fused brick_set)
  std::vector<vofod_scan> job_scans;  // the submitted batch (re-run when the LDS clustering kernel overflows)
  bool slab_bitmap = false;     // the current bitmaps were written by k_slab (dense, zeros included)
  bool lean_emit = false;       // k_emit skipped the per-root slots: launch_cluster must run the LDS clustering kernel
  bool finalize_fused = false;  // ... and the cluster table / candidate list too
  bool closefar_fused = false;  // launch_cluster answered hasCloseTo as well (dilated image inside k_flatten)
  bool bitmap_clean = false;   // the occupancy bitmaps are all-zero (k_finalize clears the words it used)
  vofod_dyn_params job_dp{};   // the dynamic parameters the pending batch was submitted with
  bool rerun = false;          // the next launch repeats this workspace's batch (LDS overflow): frame arguments and staged columns are kept
  void* d_members_big = nullptr;  // vox_cap gathered candidate members: read-back of a frame whose list exceeds the packed slot
  // state of a submitted, not yet collected batch (vofod_batch_submit / vofod_batch_collect)
  bool pending = false;
  uint32_t job_n = 0;
  GridParams job_g{};
  std::vector<float> job_tfs;
  hipEvent_t ev_done = nullptr;
  hipEvent_t ev_packed = nullptr;   // the read-back slots are complete on the chain's stream
  hipEvent_t ev_key = nullptr;      // staged pipeline: the batch's streaming kernels (brick codes) are through
  hipEvent_t ev_h2d = nullptr;      // the host-resident columns of a submitted batch have crossed the link (vofod_batch_submit returns behind it)
  hipStream_t copy_stream = nullptr;  // device-to-host copy of the slots: the chain's stream goes on with the next batch meanwhile

  void release()
  {
    if (d_stage_aos)
      (void)hipFree(d_stage_aos);
    d_stage_aos = nullptr;
    stage_aos_bytes = 0;
    void* ptrs[] = {d_bconn, ba.bricks, ba.bparent, ba.bmin, ba.bcmin, ba.blist, d_args, d_hdrs, d_bitmaps, d_wprefix, d_blocksums, va.pts, va.key, va.parent, va.csize, va.cbox, va.cclose, va.bb, d_labels, d_table, d_cand, d_ptrank, sa.extras, sa.counts, d_stage, d_packed, d_lite, d_tailc, d_dets, d_job_be, d_members_big, fs.rowT, fs.rowQ, fs.bmin, fs.nodeA, fs.bbsave, fs.frag};
    for (void* p : ptrs)
      if (p)
        (void)hipFree(p);
    if (h_packed)
      (void)hipHostFree(h_packed);
    if (h_lite)
      (void)hipHostFree(h_lite);
    if (h_args.p)
      (void)hipHostFree(h_args.p);
    if (h_dets)
      (void)hipHostFree(h_dets);
    if (ev_done)
      (void)hipEventDestroy(ev_done);
    if (ev_packed)
      (void)hipEventDestroy(ev_packed);
    if (ev_key)
      (void)hipEventDestroy(ev_key);
    if (ev_h2d)
      (void)hipEventDestroy(ev_h2d);
    if (copy_stream)
      (void)hipStreamDestroy(copy_stream);
    *this = Workspace();
  }

  hipError_t ensure(uint32_t F_, uint32_t pt_cap_, uint32_t vox_cap_, uint32_t words_cap_, uint32_t bricks_cap_ = 0)
  {
    if (F_ <= F && pt_cap_ <= pt_cap && vox_cap_ <= vox_cap && words_cap_ <= words_cap && bricks_cap_ <= bricks_cap)
      return hipSuccess;
    F_ = std::max(F_, F);
    pt_cap_ = std::max(pt_cap_, pt_cap);
    vox_cap_ = std::max(vox_cap_, vox_cap);
    words_cap_ = std::max(words_cap_, words_cap);
    bricks_cap_ = std::max(bricks_cap_, bricks_cap);
    release();
    F = F_;
    pt_cap = pt_cap_;
    vox_cap = vox_cap_;
    words_cap = words_cap_;
    bricks_cap = bricks_cap_;
    nblk_cap = (words_cap + SCAN_WPB - 1) / SCAN_WPB + 1;
    hipError_t e;
    const size_t FV = static_cast<size_t>(F) * vox_cap;
#define WS_ALLOC(ptr, bytes)                                     \
  if ((e = hipMalloc(reinterpret_cast<void**>(&ptr), (bytes))) != hipSuccess) \
    return e;
    WS_ALLOC(ba.bricks, sizeof(unsigned long long) * F * std::max<size_t>(bricks_cap, 1));
    if ((e = hipMemset(ba.bricks, 0, sizeof(unsigned long long) * F * std::max<size_t>(bricks_cap, 1))) != hipSuccess)
      return e;
    WS_ALLOC(d_bconn, sizeof(unsigned long long) * F * std::max<size_t>(bricks_cap, 1));
    WS_ALLOC(ba.bparent, sizeof(uint32_t) * F * std::max<size_t>(bricks_cap, 1));
    WS_ALLOC(ba.bmin, sizeof(uint32_t) * F * std::max<size_t>(bricks_cap, 1));
    WS_ALLOC(ba.bcmin, sizeof(uint32_t) * F * std::max<size_t>(bricks_cap, 1));
    if ((e = hipMemset(ba.bmin, 0xff, sizeof(uint32_t) * F * std::max<size_t>(bricks_cap, 1))) != hipSuccess)
      return e;
    if ((e = hipMemset(ba.bcmin, 0xff, sizeof(uint32_t) * F * std::max<size_t>(bricks_cap, 1))) != hipSuccess)
      return e;
    WS_ALLOC(ba.blist, sizeof(uint32_t) * FV);
    WS_ALLOC(d_args, sizeof(FrameArgs) * F);
    WS_ALLOC(d_hdrs, sizeof(FrameHdr) * F);
    WS_ALLOC(d_bitmaps, sizeof(unsigned long long) * F * (static_cast<size_t>(words_cap) + 2));
    WS_ALLOC(d_wprefix, sizeof(uint32_t) * F * (static_cast<size_t>(words_cap) + 2));
    WS_ALLOC(d_blocksums, sizeof(uint32_t) * F * nblk_cap);
    WS_ALLOC(va.pts, sizeof(float4) * FV);
    WS_ALLOC(va.key, sizeof(uint32_t) * FV);
    WS_ALLOC(va.parent, sizeof(uint32_t) * FV);
    WS_ALLOC(va.csize, sizeof(uint32_t) * FV);
    WS_ALLOC(va.cbox, sizeof(int32_t) * 6 * FV);
    WS_ALLOC(va.cclose, sizeof(uint32_t) * FV);
    WS_ALLOC(va.bb, sizeof(uint32_t) * FV);
    WS_ALLOC(d_labels, sizeof(uint32_t) * FV);
    WS_ALLOC(d_table, sizeof(ClusterRec) * FV);
    WS_ALLOC(d_cand, sizeof(CandMember) * FV);
    // (the frame kernel's code list: every wave of a frame's workgroup appends to a segment of its own - whole rounds of the input pass)
    fs.keys_cap = (pt_cap + IN_SEG_ALIGN - 1u) / IN_SEG_ALIGN * IN_SEG_ALIGN;
    WS_ALLOC(d_ptrank, sizeof(uint32_t) * static_cast<size_t>(F) * fs.keys_cap);
    WS_ALLOC(sa.extras, sizeof(uint32_t) * static_cast<size_t>(F) * pt_cap);
    WS_ALLOC(sa.counts, sizeof(uint32_t) * 2 * F);
    sa.keys = d_ptrank;
    WS_ALLOC(d_stage, sizeof(float) * 5 * static_cast<size_t>(F) * pt_cap);
    WS_ALLOC(d_packed, sizeof(PackedFrame) * F);
    WS_ALLOC(d_lite, sizeof(PackedLite) * F);
    WS_ALLOC(d_tailc, sizeof(vtd::TailCluster) * vtd::TP_MAXC * static_cast<size_t>(F));
    WS_ALLOC(d_dets, sizeof(vtd::FrameDets) * F);
    WS_ALLOC(d_job_be, sizeof(uint32_t) * 2 * F);
    WS_ALLOC(d_members_big, sizeof(CandMemberX) * static_cast<size_t>(std::max<uint32_t>(vox_cap, 1)));
    WS_ALLOC(fs.rowT, sizeof(unsigned long long) * 4 * FR_ROWS_MAX * static_cast<size_t>(F));
    WS_ALLOC(fs.rowQ, sizeof(uint32_t) * 4 * FR_ROWS_MAX * static_cast<size_t>(F));
    WS_ALLOC(fs.bmin, sizeof(uint32_t) * LB_MAX * static_cast<size_t>(F));
    WS_ALLOC(fs.nodeA, sizeof(unsigned long long) * 4 * LB_MAX * static_cast<size_t>(F));
    WS_ALLOC(fs.bbsave, sizeof(unsigned long long) * FR_BB64 * static_cast<size_t>(F));
    WS_ALLOC(fs.frag, sizeof(float4) * static_cast<size_t>(F) * std::max<uint32_t>(pt_cap, 1));
#undef WS_ALLOC
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h_packed), sizeof(PackedFrame) * F)) != hipSuccess)
      return e;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h_lite), sizeof(PackedLite) * F)) != hipSuccess)
      return e;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h_dets), sizeof(vtd::FrameDets) * F, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
      return e;
    std::memset(h_dets, 0, sizeof(vtd::FrameDets) * F);
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h_dets_dev), h_dets, 0)) != hipSuccess)
      return e;
    if ((e = h_args.assign(F, FrameArgs{})) != hipSuccess)
      return e;
    if ((e = hipEventCreateWithFlags(&ev_done, hipEventDisableTiming)) != hipSuccess)
      return e;
    if ((e = hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming)) != hipSuccess)
      return e;
    if ((e = hipEventCreateWithFlags(&ev_h2d, hipEventDisableTiming)) != hipSuccess)
      return e;
    if ((e = hipEventCreateWithFlags(&ev_key, hipEventDisableTiming)) != hipSuccess)
      return e;
    if ((e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)) != hipSuccess)
      return e;
    // the memsets above run on the null stream, the kernels on non-blocking streams: wait for the fills to land
    return hipDeviceSynchronize();
  }
};

struct ExploreBufs
{
  uint32_t F = 0;
  size_t jobs_cap = 0, members_cap = 0;
  unsigned long long* d_overlay = nullptr;
  uint32_t *d_stack = nullptr, *d_explored = nullptr, *d_touched = nullptr, *d_ovl_list = nullptr, *d_ovl_count = nullptr, *d_job_begin = nullptr;
  uint32_t* d_visited = nullptr;  // per frame slot: visited bits of the running flood fill (all-zero between fills)
  vc::ExploreJob* d_jobs = nullptr;
  vc::ExploreResult* d_results = nullptr;
  int* d_members = nullptr;
};

struct HostCluster
{
  ClusterRec rec;
  int cclass = VOFOD_CLASS_NONE;
  bool evaluated = false;
  vt::Boxes boxes{};
  float obb_size = NAN;
};

struct vofod_handle
{
  unsigned long long *d_prof_slab = nullptr, *d_prof_ccl = nullptr;  // stamp buffers of the VOFOD_LDS_PROF diagnostics
  // A frame of a batch held more pure-far bricks than the close-first frame kernel takes (a cold map: nothing is "close"): the
  // batch ran again with the full clustering, and so do the following ones - until the map has gained background voxels
  // (nVoxelsOver well above the count at the overflow) or was reset.
  bool cf_off = false;
  uint64_t cf_off_bg = 0;
  bool bgcount_stale = false;  // k_finalize_far patched the nVoxelsOver counters on the device: n_bg_voxels is older than they are
  bool lds_ccl_off = false;  // a frame of the batch just collected overflowed the LDS kernels: the NEXT launch (its re-run) takes the global-memory kernels, then the flag drops
  std::mutex mtx;
  vofod_static_params sp{};
  vofod_dyn_params dp{};
  std::string err;
  Prof prof;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream_tail = nullptr;  // tail (k_explore) of collected async batches
  hipStream_t stream_key = nullptr;   // staged pipeline: streaming kernels of all submitted batches, lowest priority
  // staged pipeline: the frame kernels of the submitted batches take the frame streams in turn.  A stream runs its kernels one
  // after the other, and a frame kernel lasts from its first workgroup's start to its LAST workgroup's end - about two workgroup
  // durations once the starts are staggered: with two streams a CU whose workgroup ended early had nothing to run until the other
  // stream's previous kernel had drained completely (round 5: ~15 % of the CU time; the stamps of tools/phase_table.sh show it).
  static constexpr int MAX_FRAME_STREAMS = 8;
  hipStream_t stream_frames[MAX_FRAME_STREAMS] = {};
  int n_frame_streams = 0;
  hipStream_t stream_frame = nullptr;  // == stream_frames[0]
  int frame_toggle = 0;
  // Batches in flight.  Slots are allocated on first use.  Large batches gain nothing beyond four; small ones (whole chains side
  // by side, tails included) gain up to eight PROVIDED their streams do not share hardware queues - the runtime deals a process's
  // streams onto four by default (GPU_MAX_HW_QUEUES): 142 k frames/s at 32 frames with four queues, 204-289 k with 16.
  static constexpr int MAX_INFLIGHT = 8;
  hipStream_t chain_stream[MAX_INFLIGHT] = {};  // [0] == stream; in-flight batches run their chains on separate streams and overlap on the device
  Workspace wsx[MAX_INFLIGHT - 1];              // workspaces of tickets 1..7 (ticket 0 uses ws)
  Workspace* slot(int t) { return t == 0 ? &ws : &wsx[t - 1]; }
  ExploreBufs explore_slot[MAX_INFLIGHT];  // flood-fill buffers of small submitted batches: their tails run on the tickets' own streams

  float exclude_center[3], oparea_center[3];
  uint64_t background_min_sufficient_pts = 0;

  MapGeom mg{};
  vt::Geom hg{};
  float *d_map = nullptr, *d_flags = nullptr, *d_ray = nullptr;
  unsigned long long* d_mapbits = nullptr;
  unsigned long long* d_mapclose = nullptr;  // d_mapbits dilated by hasCloseTo's stencil (k_dilate), valid while gens match
  uint64_t mapbits_gen = 0, mapclose_gen = ~0ull;
  float mapclose_dist = -1.0f;
  unsigned long long* d_counter = nullptr;  // scratch words
  unsigned long long *d_bgcount = nullptr, *h_bgcount = nullptr;  // MB_SLOTS partial nVoxelsOver counters (64 B apart); host pinned copy
  bool bgcount_fresh = false;
  hipEvent_t ev_stagger = nullptr;  // the streaming kernels of the batch submitted last have finished
  bool ev_stagger_set = false;
  hipEvent_t ev_explore = nullptr;  // the shared flood-fill buffers are free again (device tails of batches on different streams take turns)
  hipEvent_t ev_bgcount = nullptr;  // recorded behind the device-to-host copy of the background count: waited for before the count is consumed
  unsigned long long* h_counter = nullptr;  // pinned
  bool mapbits_valid = false;
  float mapbits_thr = 0;
  uint64_t n_bg_voxels = 0;

  float *d_lut_dirs = nullptr, *d_lut_offs = nullptr;
  uint8_t* d_mask = nullptr;

  Workspace ws, aux, sepws;
  ExploreBufs explore;
  std::unique_ptr<vt::Pool> pool;
  struct ClusterTables
  {
    bool valid = false;
    float leaf[3] = {0, 0, 0}, tol = 0, cmax = 0;
    ClusterParams cp{};
    int n_rows = 0;
    StencilRow* d_rows = nullptr;
    bool brick_ok = false;
    BrickParams bp{};
    BrickOff* d_boffs = nullptr;
    unsigned long long *d_sure = nullptr, *d_amb = nullptr;
    int8_t* d_pair = nullptr;  // [64*64] stencil index of offset(o2) - offset(o1), -1 when outside the forward stencil
    LbTables* d_lbtab = nullptr;  // row / octant tables of the LDS clustering (k_frame_lds)
    bool lds_ok = false;
  } ctab[2];
  int ctab_next = 0;
  struct CloseTables
  {
    bool valid = false;
    float max_dist = 0;
    int n_rows = 0;
  } closetab;
  bool ray_dirty = false;
  StencilRow* d_rows = nullptr;
  CloseRow* d_crows = nullptr;
  float* d_boxstage = nullptr;
  size_t boxstage_cap = 0;
  uint64_t* d_idxstage = nullptr;
  size_t idxstage_cap = 0;
  std::vector<CandMemberX> h_members_big;

  bool background_pts_sufficient = false, sure_background_sufficient = false;
  int detection_its = 0;
  uint32_t last_detection_id = 0;
  bool raycast_pending = false;
  int raycast_start_its = 0;
  vr::SepState sep;
  bool sep_pending = false;
  int sep_start_its = 0;
};

namespace
{

// ------------------------------------------------------------------ tables built on the host

// Decides, for a lattice offset (di,dj,dk), whether two voxel centres that far apart are within the tolerance:
// 1 certainly (nominal squared distance below tol^2 by more than the worst float evaluation error),
// 0 certainly not, 2 on the boundary -> the kernels evaluate FLANN's float expression on the actual centres.
struct EdgeClassifier
{
  double leaf[3], r2, delta;
  float r2f;
  EdgeClassifier(const float leaf_[3], float tol, float cmax)
  {
    r2f = tol * tol;
    r2 = r2f;
    for (int a = 0; a < 3; a++)
      leaf[a] = leaf_[a];
    delta = 2.0 * ulp32(cmax);  // error bound of one centre coordinate (two roundings)
  }
  int operator()(int di, int dj, int dk) const
  {
    const double ex = std::abs(di) * leaf[0], ey = std::abs(dj) * leaf[1], ez = std::abs(dk) * leaf[2];
    const double D = ex * ex + ey * ey + ez * ez;
    const double eps = 2.0 * delta + 3.0 * ulp32(static_cast<float>(std::max({ex, ey, ez, 1e-30})));
    const double E = 2.0 * eps * (ex + ey + ez) + 3.0 * eps * eps + 8.0 * D * 1.2e-7;
    if (D + E < r2)
      return 1;
    if (D - E >= r2)
      return 0;
    return 2;
  }
};

// Half stencil of the Euclidean predicate d2 < tol^2 on a lattice of pitch `leaf` (voxel-level kernel).
int build_cluster_stencil(const float leaf[3], float tol, float cmax, std::vector<StencilRow>& rows, ClusterParams& cp)
{
  rows.clear();
  int R[3];
  for (int a = 0; a < 3; a++)
  {
    R[a] = static_cast<int>(std::ceil(static_cast<double>(tol) / leaf[a])) + 1;
    if (R[a] > MAX_R)
      return VOFOD_ERR_INVALID_ARG;
  }
  const EdgeClassifier classify(leaf, tol, cmax);
  cp.row_gap = -1;
  for (int dk = 0; dk <= R[2]; dk++)
    for (int dj = (dk == 0 ? 0 : -R[1]); dj <= R[1]; dj++)
    {
      StencilRow row{};
      row.dj = static_cast<int16_t>(dj);
      row.dk = static_cast<int16_t>(dk);
      row.r_sure = -1;
      row.r_max = -1;
      row.amb = 0;
      bool sure_run = true;
      for (int di = 0; di <= R[0]; di++)
      {
        const int c = classify(di, dj, dk);
        if (c == 1 && sure_run)
          row.r_sure = static_cast<int16_t>(di);
        else
          sure_run = false;
        if (c == 1 && !sure_run)
          row.amb |= 1u << di;  // cannot happen for a monotone predicate; evaluate in float to stay safe
        if (c == 2)
          row.amb |= 1u << di;
        if (c != 0)
          row.r_max = static_cast<int16_t>(di);
      }
      if (row.r_max < 0)
        continue;
      if (dj == 0 && dk == 0)
      {
        if (row.r_max < 1)
          continue;
        cp.row_gap = std::max<int>(row.r_sure, 0);
      }
      rows.push_back(row);
    }
  if (rows.size() > MAX_STENCIL_ROWS)
    return VOFOD_ERR_INVALID_ARG;
  cp.n_rows = static_cast<int>(rows.size());
  cp.r2 = classify.r2f;
  if (cp.row_gap < 0)
    cp.row_gap = 0;
  return VOFOD_OK;
}

// Brick-level tables (kernels_brick.h).  Returns false when a 4x4x4 brick is not a clique for this tolerance.
bool build_brick_tables(const float leaf[3], float tol, float cmax, std::vector<BrickOff>& offs, std::vector<unsigned long long>& sure, std::vector<unsigned long long>& amb)
{
  offs.clear();
  sure.clear();
  amb.clear();
  const EdgeClassifier classify(leaf, tol, cmax);
  if (classify(3, 3, 3) != 1)
    return false;
  int Rb[3];
  for (int a = 0; a < 3; a++)
  {
    const int r_max = static_cast<int>(std::ceil(static_cast<double>(tol) / leaf[a])) + 1;
    Rb[a] = (r_max + 3 + 3) / 4;
  }
  for (int bz = 0; bz <= Rb[2]; bz++)
    for (int by = (bz == 0 ? 0 : -Rb[1]); by <= Rb[1]; by++)
      for (int bx = ((bz == 0 && by == 0) ? 1 : -Rb[0]); bx <= Rb[0]; bx++)
      {
        unsigned long long ms[64], ma[64];
        bool any = false, any_amb = false;
        for (int p = 0; p < 64; p++)
        {
          ms[p] = ma[p] = 0;
          const int px = p & 3, py = (p >> 2) & 3, pz = p >> 4;
          for (int q = 0; q < 64; q++)
          {
            const int qx = q & 3, qy = (q >> 2) & 3, qz = q >> 4;
            const int c = classify(4 * bx + qx - px, 4 * by + qy - py, 4 * bz + qz - pz);
            if (c == 1)
              ms[p] |= 1ull << q;
            else if (c == 2)
              ma[p] |= 1ull << q;
          }
          any |= (ms[p] | ma[p]) != 0;
          any_amb |= ma[p] != 0;
        }
        if (!any)
          continue;
        BrickOff bo{};
        bo.dx = static_cast<int8_t>(bx);
        bo.dy = static_cast<int8_t>(by);
        bo.dz = static_cast<int8_t>(bz);
        bo.has_amb = static_cast<uint8_t>(any_amb);
        for (int p = 0; p < 64; p++)
        {
          if (ms[p] | ma[p])
            bo.ua |= 1ull << p;
          bo.ub |= ms[p] | ma[p];
        }
        offs.push_back(bo);
        sure.insert(sure.end(), ms, ms + 64);
        amb.insert(amb.end(), ma, ma + 64);
      }
  return true;
}

// Tables of the LDS clustering (k_frame_lds) derived from the brick stencil: the offsets grouped into (dy,dz) rows, and per offset two
// 8x8 bit matrices over 2x2x2 octants: sure8 (every voxel pair of the two octants is certainly within the tolerance)
// and maybe8 (some pair is, certainly or on the boundary).
bool build_lds_tables(const EdgeClassifier& classify, const std::vector<BrickOff>& offs, const std::vector<unsigned long long>& sure, const std::vector<unsigned long long>& amb,
                      LbTables& t)
{
  std::memset(&t, 0, sizeof(t));
  // ball tables: every relative voxel offset the predicate can accept must lie within LB_RV per axis
  for (int c = LB_RV + 1; c <= 4 * 4 + 3; c++)
    if (classify(c, 0, 0) != 0 || classify(0, c, 0) != 0 || classify(0, 0, c) != 0)
      return false;
  for (int dz = -LB_RV; dz <= LB_RV; dz++)
    for (int dy = -LB_RV; dy <= LB_RV; dy++)
      for (int dx = -LB_RV; dx <= LB_RV; dx++)
      {
        const int c = classify(dx, dy, dz);
        if (c == 1)
          t.ball_sure[dz + LB_RV][dy + LB_RV] |= static_cast<uint16_t>(1u << (dx + LB_RV));
        else if (c == 2)
          t.ball_amb[dz + LB_RV][dy + LB_RV] |= static_cast<uint16_t>(1u << (dx + LB_RV));
      }
  if (offs.empty() || offs.size() > LB_MAX_OFF)
    return false;
  int R = 0;
  for (const auto& o : offs)
    R = std::max(R, std::abs(static_cast<int>(o.dx)));
  if (2 * R + 1 > LB_WIN)
    return false;
  t.R = R;
  for (size_t o = 0; o < offs.size(); o++)
  {
    int row = -1;
    for (int r = 0; r < t.n_rows; r++)
      if (t.rows[r].dy == offs[o].dy && t.rows[r].dz == offs[o].dz)
        row = r;
    if (row < 0)
    {
      if (t.n_rows == LB_MAX_ROWS)
        return false;
      row = t.n_rows++;
      t.rows[row].dy = offs[o].dy;
      t.rows[row].dz = offs[o].dz;
      for (auto& x : t.rows[row].o)
        x = -1;
    }
    if (std::abs(static_cast<int>(offs[o].dx)) <= 1 && std::abs(static_cast<int>(offs[o].dy)) <= 1 && std::abs(static_cast<int>(offs[o].dz)) <= 1)
      t.near_mask |= 1ull << o;
    if (std::abs(static_cast<int>(offs[o].dx)) + std::abs(static_cast<int>(offs[o].dy)) + std::abs(static_cast<int>(offs[o].dz)) == 1)
      t.axis_mask |= 1ull << o;
    const int s = offs[o].dx + R;
    t.rows[row].valid |= static_cast<uint8_t>(1u << s);
    t.rows[row].o[s] = static_cast<int8_t>(o);
    auto oct = [](int p) { return ((p & 3) >> 1) | ((((p >> 2) & 3) >> 1) << 1) | (((p >> 4) >> 1) << 2); };
    unsigned long long all = ~0ull, any = 0ull;
    bool seen[64] = {};
    for (int p = 0; p < 64; p++)
      for (int q = 0; q < 64; q++)
      {
        const int bit = oct(p) * 8 + oct(q);
        const bool is_sure = (sure[o * 64 + p] >> q) & 1ull, is_amb = (amb[o * 64 + p] >> q) & 1ull;
        if (!is_sure)
          all &= ~(1ull << bit);
        if (is_sure || is_amb)
          any |= 1ull << bit;
        seen[bit] = true;
      }
    (void)seen;
    t.oct[2 * o] = all;
    t.oct[2 * o + 1] = any;
  }
  return true;
}

// Rows of hasCloseTo's half-open cube (voxel_map.cpp:380-393), nearest rows first.
int build_close_rows(float max_dist, float vs_inv, std::vector<CloseRow>& rows)
{
  rows.clear();
  const float max_dist_idx = max_dist * vs_inv;
  const int d = static_cast<int>(std::ceil(max_dist_idx));
  if (d > MAX_R || d < 0)
    return VOFOD_ERR_INVALID_ARG;
  // int(sqrt(n2)) <= max_dist_idx  <=>  n2 <= n2max, found by direct evaluation of the reference's test
  for (int dy = -d; dy < d; dy++)
    for (int dz = -d; dz < d; dz++)
    {
      int lo = 1, hi = -1;
      for (int dx = -d; dx < d; dx++)
      {
        const int n2 = dx * dx + dy * dy + dz * dz;
        const int norm = static_cast<int>(std::sqrt(static_cast<double>(n2)));
        if (static_cast<float>(norm) <= max_dist_idx)
        {
          if (hi < lo)
            lo = dx;
          hi = dx;
        }
      }
      if (hi >= lo)
        rows.push_back(CloseRow{static_cast<int16_t>(dy), static_cast<int16_t>(dz), static_cast<int16_t>(lo), static_cast<int16_t>(hi)});
    }
  std::stable_sort(rows.begin(), rows.end(), [](const CloseRow& a, const CloseRow& b) { return a.dy * a.dy + a.dz * a.dz < b.dy * b.dy + b.dz * b.dz; });
  if (rows.size() > MAX_STENCIL_ROWS)
    return VOFOD_ERR_INVALID_ARG;
  return VOFOD_OK;
}

void fill_grid_params(vofod_handle* h, GridParams& g, const float leaf[3], bool align, const float align_center[3], const Workspace& ws)
{
  std::memset(&g, 0, sizeof(g));
  for (int a = 0; a < 3; a++)
  {
    g.leaf[a] = leaf[a];
    g.inv[a] = 1.0f / leaf[a];
    g.aco[a] = 0;
    if (align)
    {
      float aco = std::fmod(align_center[a] - leaf[a] / 2, leaf[a]);  // voxel_grid_weighted.cpp:86-97
      if (aco < 0)
        aco += leaf[a];
      g.aco[a] = aco;
    }
    g.ex_max[a] = h->exclude_center[a] + h->sp.exclude_size[a] / 2;  // vofod_nodelet.cpp:626-629
    g.ex_min[a] = h->exclude_center[a] - h->sp.exclude_size[a] / 2;
    g.op_max[a] = h->oparea_center[a] + h->sp.oparea_size[a] / 2;  // :645-648
    g.op_min[a] = h->oparea_center[a] - h->sp.oparea_size[a] / 2;
  }
  g.align = align;
  g.words_cap = ws.words_cap;
  g.vox_cap = ws.vox_cap;
  g.n_frames = 1;
  g.sparse_prefix = 0;
}

int ensure_boxstage(vofod_handle* h, size_t n)
{
  if (n <= h->boxstage_cap)
    return VOFOD_OK;
  if (h->d_boxstage)
    (void)hipFree(h->d_boxstage);
  h->boxstage_cap = std::max<size_t>(n, 1u << 20);
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_boxstage), h->boxstage_cap * sizeof(float)));
  return VOFOD_OK;
}

// read the map cells [lo, lo+n) clipped to the map into a host box
int read_box(vofod_handle* h, const float* d_map, int lo[3], int hi[3], vt::Box& box)
{
  for (int a = 0; a < 3; a++)
  {
    lo[a] = std::max(lo[a], 0);
    hi[a] = std::min(hi[a], h->hg.s[a] - 1);
    box.lo[a] = lo[a];
    box.n[a] = std::max(hi[a] - lo[a] + 1, 0);
  }
  const size_t n = static_cast<size_t>(box.n[0]) * box.n[1] * box.n[2];
  box.v.resize(n);
  if (n == 0)
    return VOFOD_OK;
  const int r = ensure_boxstage(h, n);
  if (r != VOFOD_OK)
    return r;
  KLAUNCH(h, k_read_box, dim3((n + 255) / 256), dim3(256), d_map, h->mg, lo[0], lo[1], lo[2], box.n[0], box.n[1], box.n[2], h->d_boxstage);
  HIPCHK(hipMemcpyAsync(box.v.data(), h->d_boxstage, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VOFOD_OK;
}

int scatter_set(vofod_handle* h, float* d_map, const std::vector<uint64_t>& idx, float value)
{
  if (idx.empty())
    return VOFOD_OK;
  if (idx.size() > h->idxstage_cap)
  {
    if (h->d_idxstage)
      (void)hipFree(h->d_idxstage);
    h->idxstage_cap = std::max<size_t>(idx.size(), 1u << 16);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_idxstage), h->idxstage_cap * sizeof(uint64_t)));
  }
  HIPCHK(hipMemcpyAsync(h->d_idxstage, idx.data(), idx.size() * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  KLAUNCH(h, k_scatter_set, dim3((idx.size() + 255) / 256), dim3(256), d_map, h->d_idxstage, static_cast<uint32_t>(idx.size()), value);
  HIPCHK(hipStreamSynchronize(h->stream));
  return VOFOD_OK;
}

int fill_map(vofod_handle* h, float* p, float v)
{
  KLAUNCH(h, k_fill, dim3(2048), dim3(256), p, h->mg.n, v);
  HIPCHK(hipGetLastError());
  return VOFOD_OK;
}

// reset() vofod_nodelet.cpp:1610-1632
int do_reset(vofod_handle* h)
{
  int r;
  if ((r = fill_map(h, h->d_map, h->sp.score_init)) != VOFOD_OK)
    return r;
  if ((r = fill_map(h, h->d_flags, 0.0f)) != VOFOD_OK)
    return r;
  if ((r = fill_map(h, h->d_ray, 0.0f)) != VOFOD_OK)
    return r;
  HIPCHK(hipStreamSynchronize(h->stream));
  h->detection_its = 0;
  h->raycast_pending = false;
  h->sep_pending = false;
  h->mapbits_valid = false;
  return VOFOD_OK;
}

// stage the columns of one cloud into the workspace if they live on the host; fill FrameArgs
int stage_cloud(vofod_handle* h, Workspace& ws, uint32_t f, const void* x, const void* y, const void* z, const void* intensity, const void* range, size_t stride,
                size_t n, int memspace, uint32_t flags, const float* tf)
{
  FrameArgs& a = ws.h_args[f];
  std::memset(&a, 0, sizeof(a));
  a.n = static_cast<uint32_t>(n);
  a.flags = flags;
  if (tf)
    std::memcpy(a.tf, tf, sizeof(float) * 12);
  if (memspace == VOFOD_MEM_DEVICE)
  {
    a.x = static_cast<const char*>(x);
    a.y = static_cast<const char*>(y);
    a.z = static_cast<const char*>(z);
    a.intensity = static_cast<const char*>(intensity);
    a.stride = stride;
    return VOFOD_OK;
  }
  // An array of structs in host memory (stride > 4, the three coordinates inside one struct: pcl::PointCloud<ouster_ros::Point>,
  // 48 bytes per point, what the nodelet holds - include/vofod/point_types.h) crosses the link with ONE copy of the whole block;
  // the kernels then read the columns in place at the struct's stride (k_key1<false>).  Round 3 gathered every column on the
  // host (three passes over the cloud and a synchronisation per column: ~1 ms per frame).  4 x the bytes of packed columns over
  // PCIe: the link's ceiling for this layout is ~8 k frames/s of OS1-128.
  // The kernels dereference floats in place: only structs whose stride and member offsets are multiples of 4 take this path, a
  // packed / odd layout is gathered below.  The staging block is sized by the FIRST frame of a batch (or by
  // vofod_reserve's caller, which passes through here with frame 0 too); a later frame of the same batch whose struct is
  // larger than the block's pitch allows is gathered as well - growing the block there would free it under the earlier frames'
  // copies and pointers.
  if (stride > 4 && stride % 4 == 0 && !intensity && !range && n > 0)
  {
    const char* lo = std::min({static_cast<const char*>(x), static_cast<const char*>(y), static_cast<const char*>(z)});
    const char* hi = std::max({static_cast<const char*>(x), static_cast<const char*>(y), static_cast<const char*>(z)});
    const bool aligned = (static_cast<const char*>(x) - lo) % 4 == 0 && (static_cast<const char*>(y) - lo) % 4 == 0 && (static_cast<const char*>(z) - lo) % 4 == 0;
    if (aligned && static_cast<size_t>(hi - lo) + 4 <= stride)
    {
      const size_t block = (n - 1) * stride + static_cast<size_t>(hi - lo) + 4;
      if (f == 0 && (block > ws.aos_pitch || !ws.d_stage_aos))
      {
        const size_t pitch = (ws.pt_cap * stride + 255) & ~static_cast<size_t>(255);
        HIPCHK(hipStreamSynchronize(h->stream));
        if (ws.d_stage_aos)
          (void)hipFree(ws.d_stage_aos);
        ws.d_stage_aos = nullptr;
        ws.stage_aos_bytes = 0;
        ws.aos_pitch = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&ws.d_stage_aos), pitch * ws.F));
        ws.stage_aos_bytes = pitch * ws.F;
        ws.aos_pitch = pitch;
      }
      if (ws.d_stage_aos && block <= ws.aos_pitch)
      {
        char* dst = ws.d_stage_aos + static_cast<size_t>(f) * ws.aos_pitch;
        HIPCHK(hipMemcpyAsync(dst, lo, block, hipMemcpyHostToDevice, h->stream));
        a.x = dst + (static_cast<const char*>(x) - lo);
        a.y = dst + (static_cast<const char*>(y) - lo);
        a.z = dst + (static_cast<const char*>(z) - lo);
        a.intensity = nullptr;
        a.stride = stride;
        return VOFOD_OK;
      }
    }
  }
  float* base = ws.d_stage + static_cast<size_t>(f) * ws.pt_cap * 5;
  const void* cols[5] = {x, y, z, intensity, range};
  std::vector<float> tmp;
  for (int c = 0; c < 5; c++)
  {
    if (!cols[c])
      continue;
    const void* src = cols[c];
    if (stride != 4)
    {
      tmp.resize(n);
      for (size_t i = 0; i < n; i++)
        std::memcpy(&tmp[i], static_cast<const char*>(cols[c]) + i * stride, 4);
      src = tmp.data();
    }
    HIPCHK(hipMemcpyAsync(base + static_cast<size_t>(c) * ws.pt_cap, src, n * 4, hipMemcpyHostToDevice, h->stream));
    if (stride != 4)
      HIPCHK(hipStreamSynchronize(h->stream));  // tmp is reused
  }
  a.x = reinterpret_cast<const char*>(base);
  a.y = reinterpret_cast<const char*>(base + ws.pt_cap);
  a.z = reinterpret_cast<const char*>(base + 2 * static_cast<size_t>(ws.pt_cap));
  a.intensity = intensity ? reinterpret_cast<const char*>(base + 3 * static_cast<size_t>(ws.pt_cap)) : nullptr;
  a.stride = 4;
  return VOFOD_OK;
}

// 1-D grid of a per-frame kernel: n_frames x gx blocks (see frame_block in kernels_voxelize.h)
inline dim3 fgrid(const GridParams& g, uint32_t gx) { return dim3(g.n_frames * gx); }
inline uint32_t emit_split(uint32_t n_frames) { return n_frames <= 16 ? EMIT_SPLIT : 1u; }

// Reference lattice of the single-pass input (kernels_frame.h, k_key1): what voxel_grid_weighted.cpp:72-106 yields for a cloud
// whose minimum is the operation area's corner, and the band around cell boundaries inside which the frame's own offset may
// round a point into the neighbouring cell.  Bound: |q - O| <= 2c (c = largest |coordinate|), the subtraction and the product
// each round by 2^-24 relative -> 2 * 2c * inv * 2^-23 cells per offset; the two offsets differ from whole cells by their own
// roundings (2 * c * 2^-23 * inv) and by leaf * inv != 1 (<= 2^-22 per cell of distance).  eps is four times the sum.
RefLattice fill_ref_lattice(const GridParams& g)
{
  RefLattice rl{};
  float cmax = 0.0f, dmax = 0.0f, inv_max = 0.0f;
  for (int a = 0; a < 3; a++)
  {
    const int min_b = static_cast<int>(std::floor(g.op_min[a] * g.inv[a]));
    float offset = static_cast<float>(min_b) * g.leaf[a];
    if (g.align)
      offset = offset - g.aco[a];
    rl.off[a] = offset;
    rl.dims[a] = static_cast<int>(std::floor((g.op_max[a] - offset) * g.inv[a])) + 2;
    cmax = std::max(cmax, std::max(std::fabs(g.op_min[a]), std::fabs(g.op_max[a])) + g.leaf[a]);
    dmax = std::max(dmax, static_cast<float>(rl.dims[a]));
    inv_max = std::max(inv_max, g.inv[a]);
  }
  const float u = 1.1920929e-7f;  // 2^-23
  const float e1 = 2.0f * (2.0f * cmax) * inv_max * u, e2 = 2.0f * cmax * u * inv_max + dmax * 2.0f * u;
  rl.eps = 4.0f * (e1 + e2) + 1e-4f;
  // (eps < 0.05: at most ~30 % of the points are fragile, and three words of the frame's side list hold each of them)
  // the frame kernel's bricks are bricks of this lattice: its brick bitmap has to fit LDS (LB_BITWORDS), its brick rows the
  // per-frame row tables, the brick coordinates the fields of a code (9 + 9 + 6 bits)
  for (int a = 0; a < 3; a++)
    rl.nb[a] = (rl.dims[a] + 3) / 4;
  const bool fits = rl.dims[0] > 0 && rl.dims[1] > 0 && rl.dims[2] > 0 && rl.nb[0] <= 512 && rl.nb[1] <= 512 && rl.nb[2] <= FR_MAX_NBZ &&
                    static_cast<long long>(rl.nb[0]) * rl.nb[1] * rl.nb[2] <= static_cast<long long>(LB_BITWORDS) * 32 && static_cast<uint32_t>(rl.nb[1]) * rl.nb[2] <= FR_ROWS_MAX;
  rl.on = rl.eps < 0.05f && fits;
  return rl;
}

// kernel chain K1-K6 over frames [0,n): bbox -> lattice -> occupancy bitmap -> ranks -> weighted cloud
int launch_voxelize(vofod_handle* h, Workspace& ws, GridParams& g, uint32_t n, uint32_t max_pts, bool want_ptrank, bool two_phase, const BrickParams* bricks = nullptr,
                    bool lean_hint = false)
{
  BrickParams bpv{};
  if (bricks)
    bpv = *bricks;
  bpv.bricks_cap = ws.bricks_cap;
  ws.bricks_preset = bricks != nullptr;
  g.n_frames = n;
  // lean emission: the LDS clustering kernel will follow and initialises the per-root slots itself (see plan_lds_ccl)
  ws.lean_emit = lean_hint && !bricks && !two_phase;
  ws.slab_bitmap = false;
  HIPCHK(hipMemcpyAsync(ws.d_args, ws.h_args.data(), sizeof(FrameArgs) * n, hipMemcpyHostToDevice, h->stream));
  const uint32_t gx = std::max(1u, std::min((max_pts + 255u) / 256u, 1024u));
  const uint32_t gb = std::max(1u, std::min((max_pts + 2047u) / 2048u, 1024u));  // 8 points per thread: few header atomics
  // Batches whose clustering will run inside LDS (plan_lds_ccl): the brick-first frame kernel (kernels_frame.h) reads the input
  // itself (round 5; rounds 2-4: a streaming kernel k_key1 in front), builds the voxel records and clusters them on the same LDS
  // image - launched by launch_cluster.  It needs the batch's reference lattice (bricks of the operation area's lattice in the
  // LDS bitmap); where that does not fit, the general kernels below take the batch.
  ws.ref_lattice = RefLattice{};
  bool frame_plan = ws.lean_emit && n >= 4 && !want_ptrank && !two_phase;
  if (frame_plan)
  {
    ws.ref_lattice = fill_ref_lattice(g);
    frame_plan = ws.ref_lattice.on != 0;
  }
  if (!frame_plan)
    ws.lean_emit = false;  // the general emission kernels initialise every per-voxel slot; the global clustering kernels follow
  const uint32_t lean_bit = ws.lean_emit ? 0x80000000u : 0u;
  KLAUNCH(h, k_init_hdr, dim3(n), dim3(64), ws.d_hdrs, static_cast<uint32_t*>(nullptr));
  ws.frame_fused = false;
  if (frame_plan)
  {
    bool packed = true;
    for (uint32_t f = 0; f < n && packed; f++)
    {
      const FrameArgs& a = ws.h_args[f];
      packed = a.stride == 4 && (a.n & 3u) == 0 && a.n >= 4 && ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.z)) & 15u) == 0;
    }
    ws.in_packed = packed;
    if (!h->ev_stagger)
      HIPCHK(hipEventCreateWithFlags(&h->ev_stagger, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->ev_stagger, h->stream));  // the next batch's chain may start now
    h->ev_stagger_set = true;
    ws.frame_fused = true;
    HIPCHK(hipGetLastError());
    return VOFOD_OK;
  }
  KLAUNCH(h, k_bbox, fgrid(g, gb), dim3(256), ws.d_args, g, ws.d_hdrs);
  KLAUNCH(h, k_grid, dim3(n), dim3(64), g, ws.d_hdrs);
  if (two_phase)
    return VOFOD_OK;  // caller inspects the lattice size before the bitmap is touched
  // Lattices of at most SLAB_MAX slabs of 1 Mi cells: the bitmap is built slab by slab in LDS (kernels_slab.h).
  // VOFOD_SLABS=0 keeps the global-atomic kernels (also used for the counted grid, which needs every point's rank).
  const bool slabs_on = !switch_off("VOFOD_SLABS");
  constexpr uint32_t SLAB_MAX = 32;
  const uint32_t n_slabs = (ws.words_cap + SLAB_WORDS64 - 1) / SLAB_WORDS64;
  if (slabs_on && n >= 4 && !want_ptrank && !bricks && n_slabs <= SLAB_MAX)  // a single frame is served faster by the whole chip through the global bitmap
  {
    ws.slab_bitmap = true;
    if (!ws.bitmap_clean && !g.sparse_prefix)  // voxel-level clustering: neighbour windows run into the words past the lattice, they must read as zero
      HIPCHK(hipMemsetAsync(ws.d_bitmaps, 0, sizeof(unsigned long long) * ws.F * (static_cast<size_t>(ws.words_cap) + 2), h->stream));
    HIPCHK(hipMemsetAsync(ws.sa.counts, 0, sizeof(uint32_t) * 2 * n, h->stream));
    const uint32_t gk = std::max(1u, (max_pts + KEY_THREADS * KEY_PPT - 1) / (KEY_THREADS * KEY_PPT));
    KLAUNCH(h, k_key, fgrid(g, gk), dim3(KEY_THREADS), ws.d_args, g, ws.d_hdrs, ws.sa, ws.pt_cap);
    // VOFOD_SLAB_EMIT=0 keeps the separate emission kernels for every batch size
    const bool slab_emit_on = !switch_off("VOFOD_SLAB_EMIT");
    ws.bitmap_clean = false;
    unsigned long long*& d_prof_se = h->d_prof_slab;  // VOFOD_LDS_PROF=1 (diagnostics): per-handle stamp buffer
    if (!d_prof_se && std::getenv("VOFOD_LDS_PROF"))
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&d_prof_se), sizeof(unsigned long long) * 16 * 4096));
    if (slab_emit_on && n >= 128)
    {
      // batches that fill the chip: one workgroup per frame walks the slabs in order and emits the voxel records itself
      KLAUNCH(h, k_slab_emit, dim3(n), dim3(SLAB_THREADS), g, ws.d_hdrs, ws.sa, ws.pt_cap, ws.d_bitmaps, ws.d_wprefix, ws.va, 1u | lean_bit, d_prof_se);
      if (d_prof_se)
      {
        unsigned long long t[16];
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(t, d_prof_se, sizeof(t), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[k_slab_emit] keys %.1f | zero %.1f mark %.1f count+scan %.1f out+emit %.1f us (all slabs of frame 0)\n", t[0] * 0.01, t[1] * 0.01, t[2] * 0.01, t[3] * 0.01, t[4] * 0.01);
        std::vector<unsigned long long> all(16 * n);
        HIPCHK(hipMemcpy(all.data(), d_prof_se, sizeof(unsigned long long) * 16 * n, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        double dmin = 1e9, dmax = 0, dsum = 0, smax = 0;
        for (uint32_t f = 0; f < n; f++)
        {
          t0 = std::min(t0, all[16 * f + 5]);
          t1 = std::max(t1, all[16 * f + 6]);
        }
        for (uint32_t f = 0; f < n; f++)
        {
          const double d = (all[16 * f + 6] - all[16 * f + 5]) * 0.01;
          dmin = std::min(dmin, d);
          dmax = std::max(dmax, d);
          dsum += d;
          smax = std::max(smax, (all[16 * f + 5] - t0) * 0.01);
        }
        {
          std::vector<std::pair<double, uint32_t>> byd;
          for (uint32_t f = 0; f < n; f++)
            byd.push_back({(all[16 * f + 6] - all[16 * f + 5]) * 0.01, f});
          std::sort(byd.begin(), byd.end());
          for (size_t q : {size_t(0), byd.size() / 4, byd.size() / 2, 3 * byd.size() / 4, byd.size() - 1})
          {
            const uint32_t f = byd[q].second;
            std::fprintf(stderr, "[k_slab_emit]   frame %u: %.1f us, keys %llu, zero %.1f mark %.1f count %.1f emit %.1f\n", f, byd[q].first, all[16 * f + 7], all[16 * f + 1] * 0.01, all[16 * f + 2] * 0.01,
                         all[16 * f + 3] * 0.01, all[16 * f + 4] * 0.01);
          }
        }
        std::fprintf(stderr, "[k_slab_emit] %u workgroups: span %.1f us, per-workgroup min %.1f mean %.1f max %.1f us, latest start +%.1f us\n", n, (t1 - t0) * 0.01, dmin, dsum / n, dmax, smax);
      }
    }
    else
    {
      // workgroups per frame: all slabs in parallel for small batches, fewer workgroups walking several slabs otherwise
      const uint32_t slab_g = n <= 32 ? n_slabs : std::max(1u, std::min(n_slabs, 512u / n));
      KLAUNCH(h, k_slab, fgrid(g, slab_g), dim3(SLAB_THREADS), g, ws.d_hdrs, ws.sa, ws.pt_cap, ws.d_bitmaps, ws.d_blocksums, ws.nblk_cap);
      KLAUNCH(h, k_scan_b, dim3(n), dim3(1024), g, ws.d_hdrs, ws.d_blocksums, ws.nblk_cap);
      KLAUNCH(h, k_emit, fgrid(g, ws.nblk_cap * emit_split(n)), dim3(256), g, ws.d_hdrs, ws.d_bitmaps, ws.d_blocksums, ws.nblk_cap, ws.d_wprefix, ws.va, bpv, ws.ba, 0, 1u | lean_bit, emit_split(n));
    }
    // after k_slab_emit only the extras beyond its LDS staging area are left (none on ordinary scans)
    KLAUNCH(h, k_count_extras, fgrid(g, (slab_emit_on && n >= 128) ? 4 : 24), dim3(256), g, ws.d_hdrs, ws.sa, ws.pt_cap, ws.d_bitmaps, ws.d_wprefix, ws.va);
    HIPCHK(hipGetLastError());
    return VOFOD_OK;
  }
  if (!ws.bitmap_clean)
    HIPCHK(hipMemsetAsync(ws.d_bitmaps, 0, sizeof(unsigned long long) * ws.F * (static_cast<size_t>(ws.words_cap) + 2), h->stream));
  ws.bitmap_clean = false;  // set again by whoever runs the clearing pass
  KLAUNCH(h, k_setbits, fgrid(g, gx), dim3(256), ws.d_args, g, ws.d_hdrs, ws.d_bitmaps);
  KLAUNCH(h, k_scan_a, fgrid(g, ws.nblk_cap), dim3(256), g, ws.d_hdrs, ws.d_bitmaps, ws.d_blocksums, ws.nblk_cap);
  KLAUNCH(h, k_scan_b, dim3(n), dim3(1024), g, ws.d_hdrs, ws.d_blocksums, ws.nblk_cap);
  KLAUNCH(h, k_emit, fgrid(g, ws.nblk_cap * emit_split(n)), dim3(256), g, ws.d_hdrs, ws.d_bitmaps, ws.d_blocksums, ws.nblk_cap, ws.d_wprefix, ws.va, bpv, ws.ba, bricks ? 1 : 0, lean_bit, emit_split(n));
  KLAUNCH(h, k_count, fgrid(g, gx), dim3(256), ws.d_args, g, ws.d_hdrs, ws.d_bitmaps, ws.d_wprefix, ws.va, want_ptrank ? ws.d_ptrank : nullptr, ws.pt_cap);
  HIPCHK(hipGetLastError());
  return VOFOD_OK;
}

int launch_voxelize_rest(vofod_handle* h, Workspace& ws, const GridParams& g, uint32_t n, uint32_t max_pts, bool want_ptrank)
{
  BrickParams bpv{};
  bpv.bricks_cap = ws.bricks_cap;
  const BrickParams* bricks = nullptr;
  ws.bricks_preset = false;
  const uint32_t gx = std::max(1u, std::min((max_pts + 255u) / 256u, 1024u));
  HIPCHK(hipMemsetAsync(ws.d_bitmaps, 0, sizeof(unsigned long long) * ws.F * (static_cast<size_t>(ws.words_cap) + 2), h->stream));
  ws.bitmap_clean = false;
  KLAUNCH(h, k_setbits, fgrid(g, gx), dim3(256), ws.d_args, g, ws.d_hdrs, ws.d_bitmaps);
  KLAUNCH(h, k_scan_a, fgrid(g, ws.nblk_cap), dim3(256), g, ws.d_hdrs, ws.d_bitmaps, ws.d_blocksums, ws.nblk_cap);
  KLAUNCH(h, k_scan_b, dim3(n), dim3(1024), g, ws.d_hdrs, ws.d_blocksums, ws.nblk_cap);
  KLAUNCH(h, k_emit, fgrid(g, ws.nblk_cap * emit_split(n)), dim3(256), g, ws.d_hdrs, ws.d_bitmaps, ws.d_blocksums, ws.nblk_cap, ws.d_wprefix, ws.va, bpv, ws.ba, bricks ? 1 : 0, 0u, emit_split(n));
  KLAUNCH(h, k_count, fgrid(g, gx), dim3(256), ws.d_args, g, ws.d_hdrs, ws.d_bitmaps, ws.d_wprefix, ws.va, want_ptrank ? ws.d_ptrank : nullptr, ws.pt_cap);
  HIPCHK(hipGetLastError());
  return VOFOD_OK;
}

// Stencil / brick tables of a clustering problem, cached per (leaf, tolerance, coordinate bound).
int cluster_tables(vofod_handle* h, const GridParams& g, float tol, float cmax, vofod_handle::ClusterTables** out)
{
  vofod_handle::ClusterTables* ct = nullptr;
  for (auto& c : h->ctab)
    if (c.valid && c.tol == tol && c.cmax == cmax && c.leaf[0] == g.leaf[0] && c.leaf[1] == g.leaf[1] && c.leaf[2] == g.leaf[2])
      ct = &c;
  if (!ct)
  {
    ct = &h->ctab[h->ctab_next];
    h->ctab_next ^= 1;
    // the slot's device tables are rewritten in place below: kernels of batches in flight (non-blocking streams) may still
    // read them.  A new tolerance is a rare event; the device is simply drained first.
    HIPCHK(hipDeviceSynchronize());
    ct->valid = false;
    ct->lds_ok = false;
    std::vector<StencilRow> rows;
    const int r = build_cluster_stencil(g.leaf, tol, cmax, rows, ct->cp);
    if (r != VOFOD_OK)
    {
      h->err = "cluster tolerance / leaf ratio exceeds the 63-bit neighbour window";
      return r;
    }
    std::vector<BrickOff> offs;
    std::vector<unsigned long long> sure, amb;
    ct->brick_ok = build_brick_tables(g.leaf, tol, cmax, offs, sure, amb);
    for (void* p : {static_cast<void*>(ct->d_rows), static_cast<void*>(ct->d_boffs), static_cast<void*>(ct->d_sure), static_cast<void*>(ct->d_amb)})
      if (p)
        (void)hipFree(p);
    ct->d_rows = nullptr;
    ct->d_boffs = nullptr;
    ct->d_sure = ct->d_amb = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&ct->d_rows), sizeof(StencilRow) * std::max<size_t>(rows.size(), 1)));
    HIPCHK(hipMemcpy(ct->d_rows, rows.data(), sizeof(StencilRow) * rows.size(), hipMemcpyHostToDevice));
    ct->n_rows = static_cast<int>(rows.size());
    if (ct->brick_ok)
    {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&ct->d_boffs), sizeof(BrickOff) * std::max<size_t>(offs.size(), 1)));
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&ct->d_sure), sizeof(unsigned long long) * std::max<size_t>(sure.size(), 1)));
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&ct->d_amb), sizeof(unsigned long long) * std::max<size_t>(amb.size(), 1)));
      HIPCHK(hipMemcpy(ct->d_boffs, offs.data(), sizeof(BrickOff) * offs.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(ct->d_sure, sure.data(), sizeof(unsigned long long) * sure.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(ct->d_amb, amb.data(), sizeof(unsigned long long) * amb.size(), hipMemcpyHostToDevice));
      ct->bp.n_off = static_cast<int>(offs.size());
      ct->bp.r2 = ct->cp.r2;
      if (ct->d_pair)
        (void)hipFree(ct->d_pair);
      ct->d_pair = nullptr;
      if (offs.size() <= 64)
      {
        std::vector<int8_t> pair(64 * 64, -1);
        for (size_t o1 = 0; o1 < offs.size(); o1++)
          for (size_t o2 = 0; o2 < offs.size(); o2++)
          {
            const int dx = offs[o2].dx - offs[o1].dx, dy = offs[o2].dy - offs[o1].dy, dz = offs[o2].dz - offs[o1].dz;
            for (size_t o3 = 0; o3 < offs.size(); o3++)
              if (offs[o3].dx == dx && offs[o3].dy == dy && offs[o3].dz == dz)
                pair[o1 * 64 + o2] = static_cast<int8_t>(o3);
          }
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&ct->d_pair), pair.size()));
        HIPCHK(hipMemcpy(ct->d_pair, pair.data(), pair.size(), hipMemcpyHostToDevice));
      }
      LbTables lt;
      ct->lds_ok = build_lds_tables(EdgeClassifier(g.leaf, tol, cmax), offs, sure, amb, lt);
      if (ct->lds_ok)
      {
        if (!ct->d_lbtab)
          HIPCHK(hipMalloc(reinterpret_cast<void**>(&ct->d_lbtab), sizeof(LbTables)));
        HIPCHK(hipMemcpy(ct->d_lbtab, &lt, sizeof(LbTables), hipMemcpyHostToDevice));
      }
    }
    ct->tol = tol;
    ct->cmax = cmax;
    for (int a = 0; a < 3; a++)
      ct->leaf[a] = g.leaf[a];
    ct->valid = true;
  }
  *out = ct;
  return VOFOD_OK;
}

// brick-level clustering is used when a 4x4x4 brick is a clique for the tolerance; VOFOD_CCL=voxel forces the
// voxel-level kernels (the tests compare both families against the oracle)
bool want_bricks(const vofod_handle::ClusterTables* ct, const Workspace& ws)
{
  const char* force = std::getenv("VOFOD_CCL");
  if (force && std::strcmp(force, "voxel") == 0)
    return false;
  return ct->brick_ok && ws.bricks_cap > 0;
}

// Will launch_cluster run the LDS-resident brick clustering (kernels_brick_lds.h) for this batch?  k_emit then leaves the
// per-root statistics slots to that kernel.  VOFOD_BRICK_LDS=0 keeps the global-memory kernels.
bool plan_lds_ccl(const vofod_handle* h, const vofod_handle::ClusterTables* ct, const Workspace& ws, bool allow_lds)
{
  const bool lds_on = !switch_off("VOFOD_BRICK_LDS");
  return allow_lds && lds_on && want_bricks(ct, ws) && ct->lds_ok && !h->lds_ccl_off;
}

// VOFOD_LDS_PROF (diagnostics): phase durations of the frame kernel's workgroups from the 100 MHz wall clock stamps in slots
// [s0, s0 + cnt) of the handle's stamp buffer.  =1: after every launch (the stream is synchronised: one batch at a time); =2: when the
// batch is collected (batches in flight keep their own slots: the phases as they last in the pipeline).
int print_frame_prof(vofod_handle* h, uint32_t s0, uint32_t cnt, bool sync)
{
  unsigned long long* d_prof = h->d_prof_ccl;
  std::vector<unsigned long long> t(32 * cnt);
  if (sync)
    HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(t.data(), d_prof + 32 * static_cast<size_t>(s0), sizeof(unsigned long long) * 32 * cnt, hipMemcpyDeviceToHost));
  if (const char* rawf = std::getenv("VOFOD_LDS_PROF_RAW"))
  {
    // raw mode: nothing is formatted while the pipeline runs (the JSON table below costs the host ~100 us per launch: 840 k instead
    // of 950 k frames/s) - every frame's start, end and CU are kept and written to the named file when the process ends
    // (tools/cu_gaps.py lays the frames of every CU end to end)
    struct RawLog
    {
      std::vector<unsigned long long> v;  // launch, start, end, hw id per frame
      std::string path;
      uint64_t launch = 0;
      ~RawLog()
      {
        if (FILE* fp = std::fopen(path.c_str(), "w"))
        {
          for (size_t i = 0; i + 3 < v.size(); i += 4)
            std::fprintf(fp, "%llu %.2f %.2f %u %u %u %u\n", v[i], v[i + 1] * 0.01, v[i + 2] * 0.01, static_cast<unsigned>(v[i + 3] >> 32) & 15u, static_cast<unsigned>(v[i + 3] >> 13) & 7u,
                         static_cast<unsigned>(v[i + 3] >> 12) & 1u, static_cast<unsigned>(v[i + 3] >> 8) & 15u);
          std::fclose(fp);
        }
      }
    };
    static RawLog log;
    log.path = rawf;
    for (uint32_t f = 0; f < cnt; f++)
      if (t[32 * f + 13])
      {
        const unsigned long long rec[4] = {log.launch, t[32 * f], t[32 * f + 13], t[32 * f + 21]};
        log.v.insert(log.v.end(), rec, rec + 4);
      }
    log.launch++;
    HIPCHK(hipMemsetAsync(d_prof + 32 * static_cast<size_t>(s0), 0, sizeof(unsigned long long) * 32 * cnt, h->stream));
    return VOFOD_OK;
  }
  static const char* names[13] = {"bits", "prefix", "words", "rank-a/b", "rank-c", "emit", "extras", "probe", "adjacent", "far", "exact", "minima+stats", "labels"};  // (rank-a/b includes the counting pass: stamp 14 splits them)
  std::vector<std::pair<double, uint32_t>> byd;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (uint32_t f = 0; f < cnt; f++)
  {
    if (!t[32 * f + 13])
      continue;
    byd.push_back({(t[32 * f + 13] - t[32 * f]) * 0.01, f});
    t0 = std::min(t0, t[32 * f]);
    t1 = std::max(t1, t[32 * f + 13]);
  }
  std::sort(byd.begin(), byd.end());
  double mean = 0;
  for (auto& b : byd)
    mean += b.first / byd.size();
  for (size_t q : {size_t(0), byd.size() / 2, byd.size() - 1})
  {
    if (byd.empty())
      break;
    const uint32_t f = byd[q].second;
    if (t[32 * f + 31] == ~0ull)
    {
      // a close-first frame: its own phases behind the emission
      auto us = [&](int a, int b) { return (t[32 * f + b] - t[32 * f + a]) * 0.01; };
      std::fprintf(stderr,
                   "[k_frame_lds_far] frame %u: %.1f us | keys %llu V %llu bricks %llu pure-far bricks %llu far clusters %llu candidate members %llu | input+fragile %.1f prefix %.1f words %.1f count %.1f closebits "
                   "%.1f rank-a/b %.1f rank-c %.1f emit %.1f extras %.1f near+far %.1f open %.1f stats %.1f table %.1f members %.1f\n",
                   f, byd[q].first, t[32 * f + 27], t[32 * f + 29], t[32 * f + 26], t[32 * f + 24], t[32 * f + 25], t[32 * f + 30], us(0, 1), us(1, 2), us(2, 3), us(3, 15), us(15, 14), us(14, 4), us(4, 5),
                   us(5, 6), us(6, 7), us(7, 8), us(8, 9), us(9, 10), us(10, 11), us(11, 13));
      continue;
    }
    std::fprintf(stderr, "[k_frame_lds] frame %u: %.1f us | keys %llu V %llu bricks %llu extras %llu hits %llu open %llu surviving %llu |", f, byd[q].first, t[32 * f + 27], t[32 * f + 29],
                 t[32 * f + 26], t[32 * f + 28], t[32 * f + 24], t[32 * f + 25], t[32 * f + 30]);
    for (int i = 0; i < 13; i++)
      std::fprintf(stderr, " %s %.1f", names[i], (t[32 * f + i + 1] - t[32 * f + i]) * 0.01);
    std::fprintf(stderr, " (count %.1f of rank-a/b; %llu bricks outside the largest component, %llu far hits)\n", (t[32 * f + 14] - t[32 * f + 3]) * 0.01, t[32 * f + 31] & 0xffffffffull,
                 t[32 * f + 31] >> 32);
  }
  if (!byd.empty())
    std::fprintf(stderr, "[k_frame_lds] %zu workgroups: span %.1f us, mean %.1f us\n", byd.size(), (t1 - t0) * 0.01, mean);
  // VOFOD_LDS_PROF_JSON=<file>: the phase table of the close-first frames of this launch (mean / median / max over the
  // frames, us), one JSON object per launch appended to the file - profiles/r05_frame_phases.json is made of these
  if (const char* jf = std::getenv("VOFOD_LDS_PROF_JSON"); jf && !byd.empty() && t[32 * byd[0].second + 31] == ~0ull)
  {
    static const int cuts[][2] = {{0, 16}, {16, 17}, {17, 18}, {18, 19}, {19, 20}, {20, 1}, {1, 2}, {2, 3}, {3, 15}, {15, 14}, {14, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {8, 9}, {9, 10}, {10, 11}, {11, 13}, {0, 13}};
    static const char* cnames[] = {"input", "input_wait", "grid", "grid_wait", "fragile", "fragile_wait", "prefix", "words", "count", "closebits", "rank_ab", "rank_c", "emit", "extras_restore", "cf_edges", "cf_open", "cf_stats", "cf_table", "cf_members", "total"};
    if (FILE* fp = std::fopen(jf, "a"))
    {
      double busy = 0;
        for (auto& b : byd)
          busy += b.first;
        std::fprintf(fp, "{\"frames\": %zu, \"span_us\": %.1f, \"t0_abs_us\": %.1f, \"busy_us_sum\": %.1f, \"phases\": {", byd.size(), (t1 - t0) * 0.01, t0 * 0.01, busy);
      for (size_t c = 0; c < sizeof(cuts) / sizeof(cuts[0]); c++)
      {
        std::vector<double> v;
        for (auto& b : byd)
          v.push_back((t[32 * b.second + cuts[c][1]] - t[32 * b.second + cuts[c][0]]) * 0.01);
        std::sort(v.begin(), v.end());
        double m = 0;
        for (double x : v)
          m += x / v.size();
        std::fprintf(fp, "%s\"%s\": {\"mean\": %.2f, \"median\": %.2f, \"p90\": %.2f, \"max\": %.2f}", c ? ", " : "", cnames[c], m, v[v.size() / 2], v[v.size() * 9 / 10], v.back());
      }
      // where in time the frames' workgroups started and ended relative to the first start (one CU per frame: late starts = a busy chip)
      std::vector<double> st, en;
      for (auto& b : byd)
      {
        st.push_back((t[32 * b.second] - t0) * 0.01);
        en.push_back((t[32 * b.second + 13] - t0) * 0.01);
      }
      std::sort(st.begin(), st.end());
      std::sort(en.begin(), en.end());
      std::fprintf(fp, "}, \"start_us\": {\"median\": %.1f, \"max\": %.1f}, \"end_us\": {\"median\": %.1f, \"max\": %.1f}", st[st.size() / 2], st.back(), en[en.size() / 2], en.back());
      {
        // the classification tail of the same frames (k_tail_far's stamps): boxes + gates, flood fills, records; and how long after
        // the frame's own end its tail started
        std::vector<double> bx, ex, fi, to, lag;
        for (auto& b : byd)
        {
          const unsigned long long a0 = t[32 * b.second + 22], pk = t[32 * b.second + 23];
          if (!a0)
            continue;
          const double d1 = (pk & 0xfffffull) * 0.01, d2 = ((pk >> 20) & 0xfffffull) * 0.01, d3 = ((pk >> 40) & 0xfffffull) * 0.01;
          bx.push_back(d1);
          ex.push_back(d2 - d1);
          fi.push_back(d3 - d2);
          to.push_back(d3);
          lag.push_back((static_cast<double>(a0) - static_cast<double>(t[32 * b.second + 13])) * 0.01);
        }
        if (!to.empty())
        {
          auto q = [&](std::vector<double>& v, const char* name, bool last) {
            std::sort(v.begin(), v.end());
            double m = 0;
            for (double x : v)
              m += x / v.size();
            std::fprintf(fp, "\"%s\": {\"mean\": %.2f, \"median\": %.2f, \"p90\": %.2f, \"max\": %.2f}%s", name, m, v[v.size() / 2], v[v.size() * 9 / 10], v.back(), last ? "" : ", ");
          };
          std::fprintf(fp, ", \"tail\": {");
          q(bx, "boxes_gates", false);
          q(ex, "flood_fills", false);
          q(fi, "records", false);
          q(to, "total", false);
          q(lag, "start_after_frame_end", true);
          std::fprintf(fp, "}");
        }
      }
      std::fprintf(fp, "}\n");
      std::fclose(fp);
    }
  }
  HIPCHK(hipMemset(d_prof + 32 * static_cast<size_t>(s0), 0, sizeof(unsigned long long) * 32 * cnt));
          return VOFOD_OK;
  }

// K7: Euclidean clustering of the frames in `ws`
int launch_cluster(vofod_handle* h, Workspace& ws, const GridParams& g, uint32_t n, float tol, float cmax, bool allow_lds = false, const unsigned long long* mapclose = nullptr,
                   const UpdateParams* up_tables = nullptr)
{
  ws.finalize_fused = false;
  ws.closefar_fused = false;
  ws.far_ran = false;
  vofod_handle::ClusterTables* ct = nullptr;
  const int rt = cluster_tables(h, g, tol, cmax, &ct);
  if (rt != VOFOD_OK)
    return rt;
  const uint32_t gv = (ws.vox_cap + 255u) / 256u;
  if (want_bricks(ct, ws))
  {
    BrickParams bp = ct->bp;
    bp.bricks_cap = ws.bricks_cap;
    // Batches of independent frames: the whole brick graph of a frame is clustered inside one workgroup's LDS
    // (kernels_brick_lds.h).  VOFOD_BRICK_LDS=0 keeps the global-memory kernels.
    const uint32_t lb_limit = std::getenv("VOFOD_LDS_MAX_BRICKS") ? std::min<uint32_t>(LB_MAX, std::atoi(std::getenv("VOFOD_LDS_MAX_BRICKS"))) : LB_MAX;
    if (ws.lean_emit)
    {
      (void)allow_lds;
      ws.lean_emit = false;
      unsigned long long*& d_prof_all = h->d_prof_ccl;
      if (!d_prof_all && std::getenv("VOFOD_LDS_PROF"))
      {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&d_prof_all), sizeof(unsigned long long) * 32 * 4096));
        HIPCHK(hipMemset(d_prof_all, 0, sizeof(unsigned long long) * 32 * 4096));
      }
      // VOFOD_LDS_PROF=2: a submitted batch keeps the stamps in its ticket's slots and prints them when it is collected - the
      // phases as they last with the other batches' kernels beside them (=1 synchronises behind every launch)
      ws.prof_deferred = false;
      uint32_t prof_slot0 = 0;
      if (d_prof_all && std::atoi(std::getenv("VOFOD_LDS_PROF") ? std::getenv("VOFOD_LDS_PROF") : "0") == 2 && ws.F <= 256)
        for (int t = 0; t < vofod_handle::MAX_INFLIGHT; t++)
          if (&ws == h->slot(t))
          {
            ws.prof_deferred = true;
            prof_slot0 = 256u * static_cast<uint32_t>(t);
          }
      ws.prof_slot0 = prof_slot0;
      unsigned long long* d_prof = d_prof_all ? d_prof_all + 32 * static_cast<size_t>(prof_slot0) : nullptr;
      if (ws.frame_fused)
      {
        ws.frame_fused = false;
        ws.far_ran = up_tables && mapclose && ws.close_first;
        // (four instantiations: close first or not, packed 16-byte column loads or strided ones; the profile names the algorithmic
        // variant only: k_frame_lds_far / k_frame_lds_full)
#define VOFOD_FRAME_LAUNCH(label, kern, UP, WT, CF)                                                                                                                                          \
  KLAUNCH_AS(h, label, kern, dim3(n), dim3(FR_THREADS), g, bp, ct->d_lbtab, ws.d_hdrs, ws.sa, ws.pt_cap, ws.va, ws.d_labels, lb_limit, reinterpret_cast<uint32_t*>(ws.d_table), ws.fs, h->mg, mapclose, \
          h->d_mapbits, h->d_crows, h->closetab.n_rows, UP, ws.d_table, ws.d_cand, WT, d_prof, ws.ref_lattice, ws.d_args, CF)
        const UpdateParams up_none{};
        if (ws.far_ran)  // read-only batches: cluster the far voxels only (the close-first instantiation)
        {
          if (ws.in_packed)
            VOFOD_FRAME_LAUNCH("k_frame_lds_far", k_frame_lds_far_p, *up_tables, 1, ws.close_first);
          else
            VOFOD_FRAME_LAUNCH("k_frame_lds_far", k_frame_lds_far, *up_tables, 1, ws.close_first);
        }
        else
        {
          if (ws.in_packed)
            VOFOD_FRAME_LAUNCH("k_frame_lds_full", k_frame_lds_full_p, up_tables ? *up_tables : up_none, (up_tables && mapclose) ? 1 : 0, 0);
          else
            VOFOD_FRAME_LAUNCH("k_frame_lds_full", k_frame_lds_full, up_tables ? *up_tables : up_none, (up_tables && mapclose) ? 1 : 0, 0);
        }
#undef VOFOD_FRAME_LAUNCH
        ws.finalize_fused = up_tables && mapclose;
        if (d_prof && !ws.prof_deferred)
          if (const int pr = print_frame_prof(h, 0, n, true); pr != VOFOD_OK)
            return pr;
        ws.closefar_fused = mapclose != nullptr;
        HIPCHK(hipGetLastError());
        return VOFOD_OK;
      }
      // (lean_emit without the frame kernel cannot happen: launch_voxelize clears it when it does not plan the frame path)
      h->err = "internal: lean emission without the frame kernel";
      return VOFOD_ERR_DEVICE;
    }
    if (!ws.bricks_preset)
      KLAUNCH(h, k_brick_set, fgrid(g, gv), dim3(256), g, bp, ws.d_hdrs, ws.va, ws.ba);
    ws.bricks_preset = false;
    // the stencil's pairs as per-brick connectivity masks + transitive reduction; a stencil beyond 64 offsets (or without the
    // pair tables) takes the fused probe + union kernel.  (Round 1 also had "masks + batched hooking" behind VOFOD_BRICK_MODE=2:
    // CAS storms, 470 us against 206 us - removed in round 4.)
    if (bp.n_off <= 64 && ct->d_pair)
    {
      // ws.d_table is free until k_finalize: it holds the per-brick connectivity masks (list order) in between
      unsigned long long* conn = reinterpret_cast<unsigned long long*>(ws.d_table);
      KLAUNCH(h, k_brick_conn, fgrid(g, gv * CONN_LANES), dim3(256), g, bp, ct->d_boffs, ct->d_sure, ct->d_amb, ws.d_hdrs, ws.ba, conn, ws.d_bconn);
      KLAUNCH(h, k_brick_link_tr, fgrid(g, gv), dim3(256), g, bp, ct->d_boffs, ct->d_pair, ws.d_hdrs, ws.ba, conn, ws.d_bconn);
    }
    else
      KLAUNCH(h, k_brick_union<1>, fgrid(g, gv), dim3(256), g, bp, ct->d_boffs, ct->d_sure, ct->d_amb, ws.d_hdrs, ws.ba);
    KLAUNCH(h, k_brick_root, fgrid(g, gv), dim3(256), g, bp, ws.d_hdrs, ws.ba, ws.d_bitmaps, ws.d_wprefix);
    KLAUNCH(h, k_flatten<1>, fgrid(g, gv), dim3(256), g, ws.d_hdrs, ws.va, ws.d_labels, ws.ba, ws.bricks_cap, h->mg, mapclose, h->d_mapbits, h->d_crows, h->closetab.n_rows);
    KLAUNCH(h, k_brick_clear, fgrid(g, gv), dim3(256), g, bp, ws.d_hdrs, ws.ba);
  }
  else
  {
    KLAUNCH(h, k_union<2>, fgrid(g, gv), dim3(256), g, ct->cp, ct->d_rows, ws.d_hdrs, ws.d_bitmaps, ws.d_wprefix, ws.va);
    KLAUNCH(h, k_flatten<0>, fgrid(g, gv), dim3(256), g, ws.d_hdrs, ws.va, ws.d_labels, ws.ba, 0u, h->mg, mapclose, h->d_mapbits, h->d_crows, h->closetab.n_rows);
  }
  ws.closefar_fused = mapclose != nullptr;  // k_flatten answered hasCloseTo through the dilated image
  HIPCHK(hipGetLastError());
  return VOFOD_OK;
}

// nVoxelsOver + occupancy image of the map, cached while the map and the threshold are unchanged
// nVoxelsOver (:715) from the patched counters of a sensor stream - fetched only when it can still change something (the
// background latch not yet set, debug output): a round trip of ~10 us per scan otherwise
int refresh_bgcount(vofod_handle* h)
{
  if (!h->bgcount_stale)
    return VOFOD_OK;
  h->bgcount_stale = false;
  if (!h->mapbits_valid)
    return VOFOD_OK;  // (the image is about to be rebuilt, the count with it)
  HIPCHK(hipMemcpy(h->h_bgcount, h->d_bgcount, sizeof(unsigned long long) * 8 * MB_SLOTS, hipMemcpyDeviceToHost));
  uint64_t t = 0;
  for (int i = 0; i < MB_SLOTS; i++)
    t += h->h_bgcount[8 * i];
  h->n_bg_voxels = t;
  h->bgcount_fresh = false;
  return VOFOD_OK;
}

int ensure_mapbits(vofod_handle* h, float thr)
{
  if (h->mapbits_valid && h->mapbits_thr == thr)
    return VOFOD_OK;
  h->bgcount_stale = false;  // (rebuilt below: image and count)
  // the image is shared with every batch in flight (their chains read it from streams of their own): let them finish first
  for (int t = 0; t < vofod_handle::MAX_INFLIGHT; t++)
    if (h->slot(t)->pending && h->slot(t)->ev_done)
      HIPCHK(hipEventSynchronize(h->slot(t)->ev_done));
  HIPCHK(hipMemsetAsync(h->d_bgcount, 0, sizeof(unsigned long long) * 8 * MB_SLOTS, h->stream));
  KLAUNCH(h, k_mapbits, dim3(1024), dim3(256), h->d_map, h->mg.n, thr, h->d_mapbits, h->d_bgcount);
  HIPCHK(hipMemcpyAsync(h->h_bgcount, h->d_bgcount, sizeof(unsigned long long) * 8 * MB_SLOTS, hipMemcpyDeviceToHost, h->stream));
  if (!h->ev_bgcount)
    HIPCHK(hipEventCreateWithFlags(&h->ev_bgcount, hipEventDisableTiming));
  HIPCHK(hipEventRecord(h->ev_bgcount, h->stream));
  h->bgcount_fresh = true;
  h->mapbits_gen++;
  h->mapbits_valid = true;
  h->mapbits_thr = thr;
  return VOFOD_OK;  // h_counter is valid after the next stream sync
}

// the dilated occupancy image for read-only batches (see k_dilate); call after ensure_mapbits and the close-row upload
int ensure_mapclose(vofod_handle* h, const CloseParams& cpar)
{
  if (h->mapclose_gen == h->mapbits_gen && h->mapclose_dist == h->closetab.max_dist)
    return VOFOD_OK;
  const size_t words = (h->mg.n + 63) / 64 + 2;
  HIPCHK(hipMemsetAsync(h->d_mapclose, 0, words * sizeof(unsigned long long), h->stream));
  const uint64_t chunks = static_cast<uint64_t>(h->mg.sz) * h->mg.sy * ((h->mg.sx + 63) / 64);
  KLAUNCH(h, k_dilate, dim3(static_cast<uint32_t>((chunks + 255) / 256)), dim3(256), h->mg, cpar, h->d_crows, h->d_mapbits, h->d_mapclose);
  HIPCHK(hipStreamSynchronize(h->stream));  // other chains read it from their own streams
  h->mapclose_gen = h->mapbits_gen;
  h->mapclose_dist = h->closetab.max_dist;
  return VOFOD_OK;
}

float map_cmax(const vofod_handle* h)
{
  float c = 0;
  for (int a = 0; a < 3; a++)
  {
    c = std::max(c, std::fabs(h->mg.off[a]));
    c = std::max(c, std::fabs(h->mg.off[a] + h->mg.vs * (a == 0 ? h->mg.sx : a == 1 ? h->mg.sy : h->mg.sz)));
  }
  return c + 2 * h->mg.vs;
}

int raycast_begin_locked(vofod_handle* h, const vofod_scan* scan, const float tf[12]);
int raycast_finish_locked(vofod_handle* h);

int ensure_explore(vofod_handle* h, ExploreBufs& eb, uint32_t F, size_t n_jobs, size_t n_members)
{
  const size_t ovl_words = (h->mg.n + 63) / 64;
  if (eb.F < F)
  {
    for (void* p : {static_cast<void*>(eb.d_overlay), static_cast<void*>(eb.d_stack), static_cast<void*>(eb.d_explored), static_cast<void*>(eb.d_touched),
                    static_cast<void*>(eb.d_ovl_list), static_cast<void*>(eb.d_ovl_count), static_cast<void*>(eb.d_job_begin), static_cast<void*>(eb.d_visited)})
      if (p)
        (void)hipFree(p);
    eb.F = F;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_overlay), sizeof(unsigned long long) * ovl_words * F));
    HIPCHK(hipMemset(eb.d_overlay, 0, sizeof(unsigned long long) * ovl_words * F));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_stack), sizeof(uint32_t) * vc::EX_CELLS * F));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_explored), sizeof(uint32_t) * vc::EX_CELLS * F));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_touched), sizeof(uint32_t) * vc::EX_CELLS * F));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_ovl_list), sizeof(uint32_t) * vc::EX_CELLS * F));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_ovl_count), sizeof(uint32_t) * F));
    HIPCHK(hipMemset(eb.d_ovl_count, 0, sizeof(uint32_t) * F));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_job_begin), sizeof(uint32_t) * (F + 1)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_visited), sizeof(uint32_t) * vc::EX_WORDS * F));
    HIPCHK(hipMemset(eb.d_visited, 0, sizeof(uint32_t) * vc::EX_WORDS * F));
    HIPCHK(hipDeviceSynchronize());  // null-stream memsets vs non-blocking streams
  }
  if (eb.jobs_cap < n_jobs)
  {
    if (eb.d_jobs)
      (void)hipFree(eb.d_jobs);
    if (eb.d_results)
      (void)hipFree(eb.d_results);
    eb.jobs_cap = std::max<size_t>(n_jobs * 2, 1024);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_jobs), sizeof(vc::ExploreJob) * eb.jobs_cap));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_results), sizeof(vc::ExploreResult) * eb.jobs_cap));
  }
  if (eb.members_cap < n_members)
  {
    if (eb.d_members)
      (void)hipFree(eb.d_members);
    eb.members_cap = std::max<size_t>(n_members * 2, 1 << 16);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eb.d_members), sizeof(int) * 3 * eb.members_cap));
  }
  return VOFOD_OK;
}

// Fallback of k_explore for one frame: the same flood fills and sums on read-back boxes of the map.
int host_explore_frame(vofod_handle* h, std::vector<HostCluster>& cl, const vt::MemberIndex& by_root, const std::vector<int>& job_of,
                       const std::vector<vc::ExploreJob>& jobs, std::vector<vc::ExploreResult>& results, bool no_update, float thr_frontiers, float thr_new, const vofod_dyn_params& dp)
{
  std::vector<uint64_t> pending;  // voxels this scan's classification turned into frontiers (:1712-1715)
  auto unlin = [&](uint64_t li, int i3[3]) {
    i3[0] = static_cast<int>(li % h->hg.s[0]);
    i3[1] = static_cast<int>((li / h->hg.s[0]) % h->hg.s[1]);
    i3[2] = static_cast<int>(li / (static_cast<uint64_t>(h->hg.s[0]) * h->hg.s[1]));
  };
  for (size_t ci = 0; ci < cl.size(); ci++)
  {
    const int ji = job_of[ci];
    if (ji < 0)
      continue;
    const vc::ExploreJob& job = jobs[ji];
    const vt::MemberSpan mem = by_root.of(cl[ci].rec.root);
    const int R = job.R;
    int lo[3], hi[3];
    for (int a = 0; a < 3; a++)
    {
      lo[a] = INT32_MAX;
      hi[a] = INT32_MIN;
    }
    for (const vt::Member& m : mem)
    {
      int o[3];
      h->hg.coordToIdx(m.p, o);
      for (int a = 0; a < 3; a++)
      {
        lo[a] = std::min(lo[a], o[a] - R - 1);
        hi[a] = std::max(hi[a], o[a] + R + 1);
      }
    }
    vt::Box box;
    int r = read_box(h, h->d_map, lo, hi, box);
    if (r != VOFOD_OK)
      return r;
    for (const uint64_t li : pending)
    {
      int i3[3];
      unlin(li, i3);
      if (box.has(i3))
        box.v[box.at(i3)] = thr_frontiers;
    }
    bool is_floating = true;
    std::vector<uint64_t> explored;
    for (const vt::Member& m : mem)
    {
      if (vt::explore_to_ground(h->hg, box, m.p, thr_frontiers, thr_new, static_cast<float>(R), explored))
      {
        is_floating = false;
        break;
      }
      for (const uint64_t li : explored)
      {
        int i3[3];
        unlin(li, i3);
        box.v[box.at(i3)] = thr_frontiers;
        pending.push_back(li);
      }
    }
    results[ji].floating = is_floating;
  }
  if (!no_update && !pending.empty())
  {
    const int r = scatter_set(h, h->d_map, pending, thr_frontiers);
    if (r != VOFOD_OK)
      return r;
    h->mapbits_valid = false;
  }
  // uncertainty sums (:851-865) after every frontier write of the frame
  for (size_t ci = 0; ci < cl.size(); ci++)
  {
    const int ji = job_of[ci];
    if (ji < 0 || !results[ji].floating)
      continue;
    const vc::ExploreJob& job = jobs[ji];
    const vt::MemberSpan mem = by_root.of(cl[ci].rec.root);
    int mn[3] = {job.box_lo[0], job.box_lo[1], job.box_lo[2]}, mx[3] = {job.box_hi[0], job.box_hi[1], job.box_hi[2]};
    vt::Box sub;
    const int r = read_box(h, h->d_map, mn, mx, sub);
    if (r != VOFOD_OK)
      return r;
    if (no_update)
      for (const uint64_t li : pending)
      {
        int i3[3];
        unlin(li, i3);
        if (sub.has(i3))
          sub.v[sub.at(i3)] = thr_frontiers;
      }
    const float ray = static_cast<float>(dp.voxel_map__scores__ray);
    for (const vt::Member& m : mem)
    {
      int o[3];
      h->hg.coordToIdx(m.p, o);
      if (sub.has(o))
        sub.v[sub.at(o)] = ray;
    }
    double u = 0.0;
    for (const float val : sub.v)
      u += 1.0 - val / dp.voxel_map__scores__ray;
    results[ji].conf_sum = u;
  }
  return VOFOD_OK;
}

// The body of processMsg (vofod_nodelet.cpp:926-965) for n frames.
enum FramesPhase { FRAMES_SYNC = 0, FRAMES_LAUNCH = 1, FRAMES_COLLECT = 2 };

int process_frames(vofod_handle* h, Workspace& ws, FramesPhase phase, const vofod_scan* scans, const float* tfs, uint32_t n, int flags, vofod_detection* out, size_t cap,
                   uint32_t* n_out_per_frame, size_t* n_out, vofod_scan_debug* dbg)
{
  const vofod_static_params& sp = h->sp;
  // A submitted batch is classified with the dynamic parameters of its submission: vofod_set_dynamic_params may be called
  // between submit and collect (DetectionParams.cfg semantics: "between any two calls"), and the collect half (position sigma,
  // min_points, the host fall-back tail) as well as the re-run of an overflowed batch must not mix the two parameter sets.
  if (phase == FRAMES_LAUNCH && !ws.rerun)
    ws.job_dp = h->dp;
  const vofod_dyn_params& dp = (phase == FRAMES_SYNC) ? h->dp : ws.job_dp;
  if (phase == FRAMES_COLLECT)
  {
    n = ws.job_n;
    tfs = ws.job_tfs.data();
    flags = VOFOD_SCAN_NO_MAP_UPDATE;
  }
  if (n == 0)
  {
    if (n_out)
      *n_out = 0;
    return VOFOD_OK;
  }
  if (n > ws.F)
  {
    h->err = "batch larger than max_batch_frames";
    return VOFOD_ERR_CAPACITY;
  }
  const size_t npts = static_cast<size_t>(sp.sensor_hrays) * sp.sensor_vrays;
  if (phase != FRAMES_COLLECT)
    for (uint32_t f = 0; f < n; f++)
    {
      const vofod_scan& s = scans[f];
      if (!s.x || !s.y || !s.z)
        return VOFOD_ERR_INVALID_ARG;
      if (static_cast<size_t>(s.width) * s.height != npts)  // :895-899
        return VOFOD_ERR_SIZE_MISMATCH;
    }
  const bool no_update = flags & VOFOD_SCAN_NO_MAP_UPDATE;
  // re-run of the batch that overflowed the LDS kernels: the flag is consumed here, whatever path this call takes (an early
  // error return must not leave it standing for a different batch), and only a launch of the very same job reuses its inputs
  bool rerun = false;
  if (phase != FRAMES_COLLECT)
  {
    rerun = ws.rerun && ws.job_n == n;
    ws.rerun = false;
  }
  int ret = VOFOD_OK;
  bool rc_done = false;  // ++its and the raycast role of VOFOD_SCAN_AUTO_RAYCAST have run ahead of the device tail
  auto t0 = clk::now();
  static const bool trace = std::getenv("VOFOD_TRACE") != nullptr;
  double tr_launch = 0, tr_sync1 = 0, tr_prep = 0, tr_explore = 0, tr_a = 0, tr_b = 0, tr_c = 0;
  struct DbgEvents  // destroyed on every return path
  {
    hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    ~DbgEvents()
    {
      for (auto& x : e)
        if (x)
          (void)hipEventDestroy(x);
    }
  } dbg_events;
  hipEvent_t* ev = dbg_events.e;
  if (dbg)
    for (int i = 0; i < 5; i++)
      HIPCHK(hipEventCreate(&ev[i]));

  GridParams g;
  int r = VOFOD_OK;
  const float thr_new = static_cast<float>(dp.voxel_map__thresholds__new_obstacles);
  struct ChainStream
  {
    vofod_handle* h;
    hipStream_t saved;
    ChainStream(vofod_handle* h_, hipStream_t st) : h(h_), saved(h_->stream)
    {
      if (st)
        h->stream = st;
    }
    ~ChainStream() { h->stream = saved; }
  };
  // In-flight batches on streams of their own overlap their kernel chains: a gain while a batch leaves CUs idle (32 / 64 /
  // 128 frames: +31 / +56 / +16 %), a loss once one batch's kernels fill the chip (256 frames: -6 %, co-running chains only
  // slow each other down).
  // Round 2: the chains are staggered.  A batch's streaming kernels (bounding box, brick codes: HBM bound, a few waves per CU)
  // start when the previous batch's have finished, i.e. while that batch's frame kernel (LDS bound, one workgroup per CU)
  // runs: the two phases of consecutive batches share the chip instead of taking turns.
  // (Rounds 2-3 kept VOFOD_TWO_CHAINS / VOFOD_STAGGER / VOFOD_PIPE / VOFOD_FRAME_STREAMS to switch these schemes off: lost
  // experiments, removed in round 4 with their code paths - DESIGN 5.3 has the measurements.)
  constexpr bool two_chains = true, stagger_on = true;
  // Submitted batches run as a three-stage pipeline on three streams: the streaming kernels (bounding box + brick codes:
  // vector-instruction bound, a few waves per CU) of every batch on a low-priority stream, the frame kernels (one 156 KB
  // workgroup per CU, latency bound) on a second one, the classification tails on a third (high priority).  The streaming
  // kernels of batch k+1 then fill the issue slots the frame kernel of batch k leaves idle, and when both are ready at the
  // same moment the frame kernel's workgroups are placed first (a CU full of streaming waves has no room for one).
  hipStream_t my_stream = nullptr;
  bool staged = false;
  if (two_chains && phase == FRAMES_LAUNCH)
  {
    // (a batch that leaves most CUs idle - fewer frames than half the CUs - gains more from whole chains running side by
    // side: its frame kernel shares the chip with the frame kernels of the other batches in flight)
    if (!h->stream_key || !h->stream_frame || n < 128u)
    {
      for (int t = 1; t < vofod_handle::MAX_INFLIGHT; t++)
        if (&ws == h->slot(t))
        {
          if (!h->chain_stream[t])
            HIPCHK(hipStreamCreateWithFlags(&h->chain_stream[t], hipStreamNonBlocking));
          my_stream = h->chain_stream[t];
        }
    }
    else
    {
      my_stream = h->stream_key;
      staged = true;
    }
  }
  if (phase != FRAMES_COLLECT)
  {
  ChainStream chain_guard(h, my_stream);
  if (two_chains && stagger_on && phase == FRAMES_LAUNCH && h->ev_stagger_set)
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_stagger, 0));  // the previous batch's streaming kernels are through
  if (two_chains && phase == FRAMES_LAUNCH && !(h->mapbits_valid && h->mapbits_thr == thr_new))
  {
    // the occupancy image is shared by both chains: make sure it is complete before a second stream reads it
    r = ensure_mapbits(h, thr_new);
    if (r != VOFOD_OK)
      return r;
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  // ---- stage inputs, K1-K6 (filterAndTransform :621-684)
  // (the re-run of a batch that overflowed the LDS kernels keeps the frame arguments and the staged columns of its first
  // launch: the caller's host buffers need not outlive vofod_batch_submit)
  // Host-resident batches whose frames are packed x | y | z columns at a constant pitch (one pcl / numpy block per batch, or
  // per-frame blocks of one arena) cross PCIe with ONE 2-D copy command instead of three per frame: at 256 frames the 768
  // copy calls alone cost more host time than the whole device chain.  The copy is enqueued on the streaming stage's stream:
  // it overlaps the frame kernel and the tail of the batches in front.  (Pinned memory makes it asynchronous; a pageable
  // source is staged by the runtime and blocks the call.)
  bool staged_2d = false;
  if (!rerun && n >= 2 && scans[0].memspace == VOFOD_MEM_HOST)
  {
    const char* x0 = static_cast<const char*>(scans[0].x);
    const ptrdiff_t pitch = static_cast<const char*>(scans[1].x) - x0;
    bool ok = pitch >= static_cast<ptrdiff_t>(npts * 12);
    for (uint32_t f = 0; f < n && ok; f++)
    {
      const vofod_scan& s = scans[f];
      const char* xf = x0 + static_cast<ptrdiff_t>(f) * pitch;
      ok = s.memspace == VOFOD_MEM_HOST && s.stride_bytes == 4 && static_cast<const char*>(s.x) == xf && static_cast<const char*>(s.y) == xf + npts * 4 &&
           static_cast<const char*>(s.z) == xf + npts * 8;
    }
    if (ok && ws.pt_cap == npts)
    {
      HIPCHK(hipMemcpy2DAsync(ws.d_stage, sizeof(float) * 5 * ws.pt_cap, x0, static_cast<size_t>(pitch), npts * 12, n, hipMemcpyHostToDevice, h->stream));
      for (uint32_t f = 0; f < n; f++)
      {
        float* base = ws.d_stage + static_cast<size_t>(f) * ws.pt_cap * 5;
        const int r = stage_cloud(h, ws, f, base, base + ws.pt_cap, base + 2 * static_cast<size_t>(ws.pt_cap), nullptr, nullptr, 4, npts, VOFOD_MEM_DEVICE, FA_SCAN, tfs + 12 * f);
        if (r != VOFOD_OK)
          return r;
      }
      staged_2d = true;
    }
  }
  bool host_copies = staged_2d;
  for (uint32_t f = 0; f < n && !rerun && !staged_2d; f++)
  {
    const vofod_scan& s = scans[f];
    host_copies |= s.memspace != VOFOD_MEM_DEVICE;
    const int r = stage_cloud(h, ws, f, s.x, s.y, s.z, nullptr, nullptr, s.stride_bytes, npts, s.memspace, FA_SCAN, tfs + 12 * f);
    if (r != VOFOD_OK)
      return r;
  }
  // include/vofod.h promises that the scans' host buffers need not outlive vofod_batch_submit.  A copy from page-locked memory
  // is truly asynchronous - and on the streaming stage's stream it is queued behind the previous batch's streaming kernels:
  // it may start long after the call has returned, when a caller that refills its arena has already overwritten it.
  // The event marks the end of the batch's copies; the submitting call returns behind it (below).
  if (host_copies && phase == FRAMES_LAUNCH && ws.ev_h2d)
    HIPCHK(hipEventRecord(ws.ev_h2d, h->stream));
  else
    host_copies = false;
  // an error return between here and the wait below must not leave copies from the caller's buffers in flight
  struct H2DGuard
  {
    hipEvent_t ev;
    bool armed;
    ~H2DGuard()
    {
      if (armed)
        (void)hipEventSynchronize(ev);
    }
  } h2d_guard{ws.ev_h2d, host_copies};
  const float leaf[3] = {sp.voxel_size, sp.voxel_size, sp.voxel_size};
  const int zero[3] = {0, 0, 0};
  float align_center[3];
  h->hg.idxToCoord(zero, align_center);  // :664
  fill_grid_params(h, g, leaf, true, align_center, ws);
  // ---- tables and map images of K8/K9 (findCloseFarClusters :703-750): independent of the frames, prepared first
  r = ensure_mapbits(h, thr_new);
  if (r != VOFOD_OK)
    return r;
  // (a sensor stream's patched nVoxelsOver counters - k_finalize_far of the PREVIOUS scan - are fetched here, before this scan's
  // own update is enqueued, and only while the count can still change something: the background latch, the debug output)
  if (!h->background_pts_sufficient || dbg)
    if (const int rb = refresh_bgcount(h); rb != VOFOD_OK)
      return rb;
  if (!h->closetab.valid || h->closetab.max_dist != static_cast<float>(dp.ground_points_max_distance))
  {
    std::vector<CloseRow> crows;
    r = build_close_rows(static_cast<float>(dp.ground_points_max_distance), h->mg.vs_inv, crows);
    if (r != VOFOD_OK)
    {
      h->err = "ground_points_max_distance / voxel_size exceeds the 63-bit window";
      return r;
    }
    HIPCHK(hipMemcpy(h->d_crows, crows.data(), sizeof(CloseRow) * crows.size(), hipMemcpyHostToDevice));
    h->closetab.valid = true;
    h->closetab.max_dist = static_cast<float>(dp.ground_points_max_distance);
    h->closetab.n_rows = static_cast<int>(crows.size());
  }
  CloseParams cpar{h->closetab.n_rows, thr_new};
  // read-only batches: the map's dilated image answers hasCloseTo with one bit per voxel (VOFOD_DILATE=0: stencil sweep)
  const bool dilate_on = !switch_off("VOFOD_DILATE");
  const bool use_dilated = dilate_on && no_update && n >= 4;
  if (use_dilated)
  {
    r = ensure_mapclose(h, cpar);
    if (r != VOFOD_OK)
      return r;
  }
  if (dbg)
    HIPCHK(hipEventRecord(ev[0], h->stream));
  {
    // the clustering family is known up front, so the emission kernel can register the voxels in their bricks
    vofod_handle::ClusterTables* ct = nullptr;
    r = cluster_tables(h, g, static_cast<float>(dp.ground_points_max_distance), map_cmax(h), &ct);
    if (r != VOFOD_OK)
      return r;
    g.sparse_prefix = want_bricks(ct, ws) ? 1u : 0u;
    // (Fusing the brick registration into k_emit was measured slower - 194 us vs 96 + 50 us for 32 frames: the returning
    // atomicOr sits inside the load-balanced emission loop - so it stays a kernel of its own.)
    const bool lds_plan = plan_lds_ccl(h, ct, ws, no_update && n >= 4);
    h->lds_ccl_off = false;  // one-shot: only the re-run of the batch that overflowed stays off the LDS kernels
    r = launch_voxelize(h, ws, g, n, static_cast<uint32_t>(npts), false, false, nullptr, lds_plan);
  }
  if (r != VOFOD_OK)
    return r;
  if (dbg)
    HIPCHK(hipEventRecord(ev[1], h->stream));
  if (staged)
  {
    HIPCHK(hipEventRecord(ws.ev_key, h->stream));
    // Consecutive frame kernels alternate between two streams: frame kernel k+1 waits neither for the LAST workgroup of frame
    // kernel k (its workgroups take the CUs as those of k retire: frames last 260-350 us) nor for the 14 us launch hand-off
    // behind it.  Measured +2..4 % (541 / 533 / 533 k against 519 / 510 / 534 k frames/s, alternating runs on one box); the
    // pipeline's pace is then set by k_key1, which runs as a guest of the frame kernels all the time (one wave per SIMD beside a
    // frame workgroup: ~440 us per batch).
    h->frame_toggle = (h->frame_toggle + 1) % std::max(h->n_frame_streams, 1);
    h->stream = h->stream_frames[h->frame_toggle];
    HIPCHK(hipStreamWaitEvent(h->stream, ws.ev_key, 0));
  }

  // ---- K7 clusterCloud :932
  UpdateParams up{};
  up.score_point = static_cast<float>(dp.voxel_map__scores__point);
  up.score_unknown = static_cast<float>(dp.voxel_map__scores__unknown);
  up.min_points = dp.classification__min_points;
  up.cand_max_extent = static_cast<float>(dp.classification__max_size * (1.0 + 1e-4) + 1e-3 * sp.voxel_size);
  up.no_update = no_update;
  // k_slab rewrites the bitmap densely and the brick clustering reads it at set bits only: no need to clean it after use;
  // then the LDS clustering kernel can write the cluster table and the candidate list itself (nothing left for k_finalize)
  const bool frame_path = ws.frame_fused;  // brick-first frame kernel: the global occupancy bitmaps are not touched at all
  const bool bitmap_was_clean = ws.bitmap_clean;
  const bool keep_dirty = frame_path || (ws.slab_bitmap && g.sparse_prefix);
  // Read-only batches cluster close first (k_frame_lds): only the far clusters are ever used (vofod_nodelet.cpp:946-963).  The
  // debug view of ALL clusters keeps the full clustering; dbg[0].far_only asks for the production path's view instead.
  // VOFOD_CLOSE_FIRST=0: the full clustering everywhere.
  const bool close_first_on = !switch_off("VOFOD_CLOSE_FIRST");
  const bool dbg_far_only = dbg && dbg[0].far_only;
  if (h->cf_off && (h->n_bg_voxels > h->cf_off_bg + h->cf_off_bg / 4 + 1000 || h->n_bg_voxels < h->cf_off_bg))
    h->cf_off = false;  // the map has changed a lot since: try the close-first kernel again
  ws.close_first = (close_first_on && use_dilated && !h->cf_off) ? (dbg ? (dbg_far_only ? 2 : 0) : 1) : 0;
  const uint32_t gv = (ws.vox_cap + 255u) / 256u;
  // A single map-updating scan without debug output (the reference's own mode) clusters close first on the general path
  // (kernels_far.h): close bits from hasCloseTo's stencil with every voxel as its own cluster, then edges and unions around the
  // far voxels only - instead of the six brick kernels over the whole frame.
  const bool dtail_wanted = !switch_off("VOFOD_DEVICE_TAIL");
  bool far_single = close_first_on && !h->cf_off && !dbg && !no_update && n == 1 && phase == FRAMES_SYNC && dtail_wanted && !frame_path && !keep_dirty;
  if (far_single)
  {
    vofod_handle::ClusterTables* ct = nullptr;
    r = cluster_tables(h, g, static_cast<float>(dp.ground_points_max_distance), map_cmax(h), &ct);
    if (r != VOFOD_OK)
      return r;
    far_single = ct->n_rows > 0 && ws.vox_cap >= FAR_MAX;
    if (far_single)
    {
      uint32_t* far_list = ws.d_labels;  // (labels are not written on this path)
      KLAUNCH(h, k_closefar, fgrid(g, gv), dim3(256), g, h->mg, cpar, h->d_crows, ws.d_hdrs, h->d_mapbits, ws.va, static_cast<const uint32_t*>(nullptr), static_cast<const unsigned long long*>(nullptr), ws.d_cand,
              ws.d_hdrs);
      KLAUNCH(h, k_closefar_sweep, fgrid(g, (ws.vox_cap * 16u + 255u) / 256u), dim3(256), g, h->mg, cpar, h->d_crows, ws.d_hdrs, h->d_mapbits, ws.va, ws.d_cand, far_list, ws.d_hdrs, ws.vox_cap);
      const uint32_t items = FAR_MAX * 2u * static_cast<uint32_t>(ct->n_rows);
      KLAUNCH(h, k_far_edges, dim3((items + 255u) / 256u), dim3(256), g, ct->cp, ct->d_rows, ws.d_hdrs, ws.d_bitmaps, ws.d_wprefix, ws.va, far_list);
      KLAUNCH(h, k_far_final, dim3(1), dim3(1024), g, ws.d_hdrs, ws.va, far_list, up, ws.d_table, ws.d_cand);
      // The occupancy image is patched where the update flips a bit (valid while one voxel of the scan falls into one map cell -
      // the aligned lattice - and the flood fills' frontier value is no background value: their writes then flip nothing).
      ws.mapbits_patched = g.align && static_cast<float>(dp.voxel_map__thresholds__frontiers) <= thr_new;
      KLAUNCH(h, k_finalize_far, dim3(gv), dim3(256), g, h->mg, up, ws.d_hdrs, ws.va, h->d_map, h->d_flags, ws.d_bitmaps, ws.mapbits_patched ? h->d_mapbits : nullptr, h->d_bgcount, thr_new);
      ws.closefar_fused = true;
      ws.finalize_fused = true;
    }
  }
  if (!far_single)
  {
  r = launch_cluster(h, ws, g, n, static_cast<float>(dp.ground_points_max_distance), map_cmax(h), no_update && n >= 4, use_dilated ? h->d_mapclose : nullptr,
                     (keep_dirty && no_update) ? &up : nullptr);
  if (r != VOFOD_OK)
    return r;
  }
  if (dbg)
    HIPCHK(hipEventRecord(ev[2], h->stream));

  // ---- K8/K9 findCloseFarClusters :703-750 (tables and images were prepared before the chain was enqueued)
  if (!ws.closefar_fused)
  {
    // few frames: the undecided voxels go through a list to a sweep kernel with 16 lanes per voxel (ws.d_cand is free until
    // k_finalize); many frames: swept in place
    const bool split = n < 8 && !use_dilated;
    KLAUNCH(h, k_closefar, fgrid(g, gv), dim3(256), g, h->mg, cpar, h->d_crows, ws.d_hdrs, h->d_mapbits, ws.va, ws.d_labels, use_dilated ? h->d_mapclose : nullptr,
            split ? ws.d_cand : nullptr, ws.d_hdrs);
    if (split)
      KLAUNCH(h, k_closefar_sweep, fgrid(g, (ws.vox_cap * 16u + 255u) / 256u), dim3(256), g, h->mg, cpar, h->d_crows, ws.d_hdrs, h->d_mapbits, ws.va, ws.d_cand);
  }
  ws.closefar_fused = false;
  if (dbg)
    HIPCHK(hipEventRecord(ev[3], h->stream));

  // ---- K10 updateVMaps :943-950 + cluster table + candidate members
  if (!ws.finalize_fused)
    KLAUNCH(h, k_finalize, fgrid(g, gv), dim3(256), g, h->mg, up, ws.d_hdrs, ws.va, ws.d_labels, h->d_map, h->d_flags, ws.d_table, ws.d_cand, keep_dirty ? nullptr : ws.d_bitmaps);
  ws.bitmap_clean = frame_path ? bitmap_was_clean : !keep_dirty;
  ws.finalize_fused = false;
  const bool lite_on = !switch_off("VOFOD_LITE");
  const bool dtail_on = !switch_off("VOFOD_DEVICE_TAIL");
  // (round 4: a single map-updating scan - the reference's own mode - takes the device tail too: no cluster table down, explore
  // jobs up, results down between the kernels; the flood fills then write their frontiers to the map itself, vofod_nodelet.cpp:1712-1715)
  const bool single_update = !no_update && n == 1 && phase == FRAMES_SYNC;
  ws.dtail = dtail_on && !dbg && ((n >= 4 && no_update) || single_update);
  ws.lite = !ws.dtail && lite_on && !dbg && n >= 4 && no_update;
  hipStream_t tail_stream_used = h->stream;  // where the device tail's last operation was enqueued
  if (ws.dtail && single_update && (flags & VOFOD_SCAN_AUTO_RAYCAST))
  {
    // VOFOD_SCAN_AUTO_RAYCAST applies the pending raycast update (or starts a pass) between ++its and the classification
    // (:949-963), i.e. between the kernels enqueued so far and the tail: the scan's status comes back first (a scan that has to
    // run again, or failed, must not have moved the raycast state), then the raycast role, then the tail kernels.
    HIPCHK(hipMemcpyAsync(&ws.h_packed[0].hdr, ws.d_hdrs, sizeof(FrameHdr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (ws.h_packed[0].hdr.status == VOFOD_OK)
    {
      h->detection_its++;  // :949
      if (h->raycast_pending)
        raycast_finish_locked(h);
      else
        raycast_begin_locked(h, &scans[0], tfs);
      rc_done = true;
    }
  }
  if (ws.dtail)
  {
    // Classification tail on the device (kernels_tail.h): boxes + gates, flood fills, detection records; nothing comes back
    // but the records.  The background latch (:716-721) is needed now, not at collect time.
    if (h->bgcount_fresh)
    {
      if (h->ev_bgcount)
        HIPCHK(hipEventSynchronize(h->ev_bgcount));
      uint64_t t = 0;
      for (int i = 0; i < MB_SLOTS; i++)
        t += h->h_bgcount[8 * i];
      h->n_bg_voxels = t;
      h->bgcount_fresh = false;
    }
    if (h->n_bg_voxels > h->background_min_sufficient_pts)
      h->background_pts_sufficient = true;
    // A small submitted batch (fewer frames than half the CUs: its whole chain runs on the ticket's stream) keeps flood-fill
    // buffers of its own and its tail on that stream: the tails of the batches in flight overlap.  On the shared tail stream
    // they took turns - k_tail_prep + k_explore + k_tail_finish last 250-280 us in the company of other batches' frame kernels
    // (120 us alone), and that turn WAS the pace of 32-frame batches (device timeline, tools/trace32.sh).
    int own_tail = -1;
    if (phase == FRAMES_LAUNCH && two_chains && !staged && n < 128u)
      for (int t = 0; t < vofod_handle::MAX_INFLIGHT; t++)
        if (&ws == h->slot(t))
          own_tail = t;
    ExploreBufs& eb = own_tail >= 0 ? h->explore_slot[own_tail] : h->explore;
    {
      const uint32_t ebF = own_tail >= 0 ? std::max<uint32_t>(eb.F, (n + 31u) & ~31u) : h->ws.F;
      r = ensure_explore(h, eb, ebF, static_cast<size_t>(ebF) * vtd::TP_MAXC, static_cast<size_t>(ebF) * vtd::TP_MAXM);
    }
    if (r != VOFOD_OK)
      return r;
    // A submitted batch runs its tail on the handle's tail stream: one wave per frame does the flood
    // fills (latency bound, ~0.1 ms), which overlaps with the streaming kernels of the next batch instead of delaying them.
    // The tail stream also serialises the tails of the batches in flight on the shared flood-fill buffers.
    hipStream_t chain_stream = h->stream;
    struct TailStream
    {
      vofod_handle* h;
      hipStream_t saved;
      ~TailStream() { h->stream = saved; }
    } tail_guard{h, h->stream};
    if (!h->ev_explore)
      HIPCHK(hipEventCreateWithFlags(&h->ev_explore, hipEventDisableTiming));
    // (large batches: the tails take turns on the tail stream, underneath the frame kernel of the next batch)
    if (own_tail >= 0)
      ;  // (own buffers, own stream: nothing to wait for)
    else if (phase == FRAMES_LAUNCH && h->stream_tail)
    {
      HIPCHK(hipEventRecord(ws.ev_packed, chain_stream));  // the cluster tables of this batch are complete
      h->stream = h->stream_tail;
      HIPCHK(hipStreamWaitEvent(h->stream, ws.ev_packed, 0));
    }
    else
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_explore, 0));  // (a synchronous call: wait for the tails of batches in flight)
    vtd::TailParams tp{};
    tp.min_points = dp.classification__min_points;
    tp.max_distance = dp.classification__max_distance;
    tp.max_size = dp.classification__max_size;
    tp.max_explore = dp.classification__max_explore_distance;
    tp.voxel_size = sp.voxel_size;
    tp.latches = (h->background_pts_sufficient && h->sure_background_sufficient) ? 1 : 0;
    vc::ExploreParams ep{};
    ep.thr_unknown = static_cast<float>(dp.voxel_map__thresholds__frontiers);
    ep.thr_ground = thr_new;
    ep.frontier_value = static_cast<float>(dp.voxel_map__thresholds__frontiers);
    ep.ray_score = dp.voxel_map__scores__ray;
    ep.no_update = no_update ? 1 : 0;
    ep.stack_cap = vc::EX_CELLS;
    // the records (135 KB) go straight into the pinned host slots from the last tail kernel: no copy command on any stream (see k_tail_finish)
    static const bool diag_no_tail = std::getenv("VOFOD_DIAG_NO_TAIL") != nullptr;  // (diagnostics: timing of the pipeline without the tail kernel - results are wrong)
    if (diag_no_tail && n >= 128u)
      ;
    else if ((ws.far_ran && ws.close_first == 1) || far_single)
      // close-first frames: ordered lists from the frame kernel (or k_far_final), the whole tail in one kernel of one wave per frame
      KLAUNCH(h, vtd::k_tail_far, dim3((n + vtd::TAIL_WPB - 1) / vtd::TAIL_WPB), dim3(64 * vtd::TAIL_WPB), g, ws.d_hdrs, ws.d_args, ws.d_table, ws.d_cand, ws.va, h->mg, tp, ep, eb.d_jobs, eb.d_members, h->d_map, eb.d_overlay, eb.d_stack, eb.d_explored, eb.d_touched,
              eb.d_ovl_list, eb.d_ovl_count, eb.d_results, eb.d_visited, ws.d_dets, ws.h_dets_dev, far_single ? ws.d_tailc : static_cast<vtd::TailCluster*>(nullptr),
              (ws.prof_deferred && h->d_prof_ccl && !far_single) ? h->d_prof_ccl + 32 * static_cast<size_t>(ws.prof_slot0) : static_cast<unsigned long long*>(nullptr));
    else
    {
      KLAUNCH(h, vtd::k_tail_prep, dim3(n), dim3(vtd::TP_THREADS), g, ws.d_hdrs, ws.d_args, ws.d_table, ws.d_cand, ws.va, h->mg, tp, eb.d_jobs, ws.d_job_be, ws.d_job_be + ws.F, eb.d_members, ws.d_tailc,
              ws.d_dets);
      KLAUNCH(h, vc::k_explore, dim3(n), dim3(64), ep, h->mg, eb.d_jobs, ws.d_job_be, ws.d_job_be + ws.F, eb.d_members, h->d_map, eb.d_overlay, eb.d_stack, eb.d_explored, eb.d_touched, eb.d_ovl_list,
              eb.d_ovl_count, eb.d_results, eb.d_visited);
      KLAUNCH(h, vtd::k_tail_finish, dim3(n), dim3(64), ws.d_tailc, eb.d_results, ws.d_dets, ws.h_dets_dev);
    }
    if (own_tail < 0)
      HIPCHK(hipEventRecord(h->ev_explore, h->stream));  // the shared flood-fill buffers are free again
    tail_stream_used = h->stream;
  }
  else if (ws.lite)
  {
    KLAUNCH(h, k_pack_lite, dim3(n), dim3(256), g, ws.d_hdrs, ws.d_table, ws.d_cand, ws.va, ws.d_lite);
    if (phase == FRAMES_LAUNCH)
    {
      HIPCHK(hipEventRecord(ws.ev_packed, h->stream));
      HIPCHK(hipStreamWaitEvent(ws.copy_stream, ws.ev_packed, 0));
      HIPCHK(hipMemcpyAsync(ws.h_lite, ws.d_lite, sizeof(PackedLite) * n, hipMemcpyDeviceToHost, ws.copy_stream));
    }
    else
      HIPCHK(hipMemcpyAsync(ws.h_lite, ws.d_lite, sizeof(PackedLite) * n, hipMemcpyDeviceToHost, h->stream));
  }
  else
  {
    KLAUNCH(h, k_pack, fgrid(g, (std::max(SPEC_C, SPEC_M) + 255) / 256), dim3(256), g, ws.d_hdrs, ws.d_table, ws.d_cand, ws.va, ws.d_packed);
    HIPCHK(hipMemcpyAsync(ws.h_packed, ws.d_packed, sizeof(PackedFrame) * n, hipMemcpyDeviceToHost, h->stream));
  }
  if (dbg)
    HIPCHK(hipEventRecord(ev[4], h->stream));
  tr_launch = ms_since(t0);
  if (phase == FRAMES_LAUNCH)
  {
    h2d_guard.armed = false;
    if (host_copies)
      HIPCHK(hipEventSynchronize(ws.ev_h2d));  // (the whole chain is enqueued by now: the device works while the host waits for the link)
    HIPCHK(hipEventRecord(ws.ev_done, ws.dtail ? tail_stream_used : ws.lite ? ws.copy_stream : h->stream));
    ws.pending = true;
    ws.job_n = n;
    ws.job_g = g;
    if (tfs != ws.job_tfs.data())
      ws.job_tfs.assign(tfs, tfs + 12 * static_cast<size_t>(n));
    if (scans != ws.job_scans.data())
      ws.job_scans.assign(scans, scans + n);
    return VOFOD_OK;
  }
  }  // launch part
  else
    g = ws.job_g;
  if (phase == FRAMES_COLLECT)
  {
    HIPCHK(hipEventSynchronize(ws.ev_done));
    ws.pending = false;
    if (ws.prof_deferred && h->d_prof_ccl)
    {
      ws.prof_deferred = false;
      if (const int pr = print_frame_prof(h, ws.prof_slot0, n, false); pr != VOFOD_OK)
        return pr;
    }
  }
  else
  {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (ws.prof_deferred && h->d_prof_ccl)
    {
      ws.prof_deferred = false;
      if (const int pr = print_frame_prof(h, ws.prof_slot0, n, false); pr != VOFOD_OK)
        return pr;
    }
  }
  tr_sync1 = ms_since(t0);
  if (ws.lite)
    for (uint32_t f = 0; f < n; f++)
      ws.h_packed[f].hdr = ws.h_lite[f].hdr;  // the tail below reads the frames through the packed slots
  if (ws.dtail)
    for (uint32_t f = 0; f < n; f++)
      ws.h_packed[f].hdr.status = ws.h_dets[f].status;
  for (uint32_t f = 0; f < n; f++)
    if (ws.h_packed[f].hdr.status == CCL_RETRY_STATUS)
    {
      // a frame held more bricks than the LDS clustering kernel takes: nothing of this batch was used (batches never
      // update the map); the caller runs it again on the global-memory kernels
      h->lds_ccl_off = true;
      ws.rerun = true;
      ws.job_n = n;
      ws.bitmap_clean = false;
      return CCL_RETRY_STATUS;
    }
  if (!no_update)
  {
    bool keep = false;
    if (ws.mapbits_patched)
    {
      // (the image was patched by k_finalize_far - unless the scan has to run again or left the map: then it is rebuilt)
      keep = ws.h_packed[0].hdr.status == VOFOD_OK && h->mapbits_valid;
      ws.mapbits_patched = false;
    }
    if (keep)
    {
      h->bgcount_stale = true;  // (the counters on the device are newer than the host's sum: fetched when somebody needs it)
      h->mapbits_gen++;         // (the dilated image of the batches is of an older state)
    }
    else
      h->mapbits_valid = false;
  }

  if (h->bgcount_fresh)
  {
    if (h->ev_bgcount)
      HIPCHK(hipEventSynchronize(h->ev_bgcount));  // the copy may have been enqueued on another chain's stream
    uint64_t t = 0;
    for (int i = 0; i < MB_SLOTS; i++)
      t += h->h_bgcount[8 * i];
    h->n_bg_voxels = t;
    h->bgcount_fresh = false;
  }
  // (n_bg_voxels is the count findCloseFarClusters saw, i.e. of the map BEFORE this call's update; counters patched by this very
  // call - bgcount_stale set above - belong to the next one and are not fetched here)
  if (h->n_bg_voxels > h->background_min_sufficient_pts)  // :716-721
    h->background_pts_sufficient = true;
  for (uint32_t f = 0; f < n; f++)
    if (ws.h_packed[f].hdr.status == CF_RETRY_STATUS)
    {
      // a frame with more pure-far bricks than the close-first kernel takes: nothing of this batch was used; the caller runs
      // it again (same descriptors, staged columns kept) with the full clustering
      h->cf_off = true;
      h->cf_off_bg = h->n_bg_voxels;
      ws.rerun = true;
      ws.job_n = n;
      return CCL_RETRY_STATUS;
    }
  if (!no_update && !rc_done)
  {
    h->detection_its++;  // :949
    if (flags & VOFOD_SCAN_AUTO_RAYCAST)
    {
      if (h->raycast_pending)
        raycast_finish_locked(h);
      else
        raycast_begin_locked(h, &scans[0], tfs);
    }
  }
  double dev_ms[5] = {0, 0, 0, 0, 0};
  if (dbg)
  {
    for (int i = 0; i < 4; i++)
    {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
      dev_ms[i] = ms;
    }
  }

  if (ws.dtail)
  {
    // ---- the tail ran on the device: extractDetections' record (:848-877) from the raw detections, frame by frame
    uint32_t fb = 0;
    for (uint32_t f = 0; f < n; f++)
      fb |= ws.h_dets[f].fallback;
    if (!fb)
    {
      size_t total = 0;
      if (phase == FRAMES_COLLECT && out)
      {
        // an output array too small for this batch: nothing is consumed - the ticket stays pending, ids are not handed
        // out, *n_out tells the size to come back with
        size_t need = 0;
        for (uint32_t f = 0; f < n; f++)
          need += ws.h_dets[f].n;
        if (need > cap)
        {
          ws.pending = true;
          *n_out = need;
          return VOFOD_ERR_CAPACITY;
        }
      }
      for (uint32_t f = 0; f < n; f++)
      {
        const vtd::FrameDets& D = ws.h_dets[f];
        if (D.status != VOFOD_OK)
          ret = D.status;
        const float* tf = tfs + 12 * f;
        const float tpos[3] = {tf[3], tf[7], tf[11]};
        for (uint32_t i = 0; i < D.n; i++)
        {
          const vtd::DetRaw& R = D.d[i];
          vofod_detection det{};
          const float d[3] = {tpos[0] - R.center[0], tpos[1] - R.center[1], tpos[2] - R.center[2]};
          const double det_dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
          det.id = h->last_detection_id++;
          det.frame = f;
          det.n_points = R.n_points;
          const float cov = static_cast<float>(std::sqrt(det_dist) * dp.output__position_sigma);
          for (int q = 0; q < 3; q++)
            det.covariance[4 * q] = cov;
          const double u = R.conf_sum / R.n_points;  // :860-865
          det.confidence = static_cast<float>(1.0 / std::exp(u));
          const double vray_res = sp.sensor_vfov / static_cast<double>(sp.sensor_vrays);
          const double hray_res = 2 * M_PI / static_cast<double>(sp.sensor_hrays);
          det.detection_probability = std::min(std::atan(1.0 / det_dist) / (vray_res * dp.classification__min_points), 1.0) * std::min(std::atan(1.0 / det_dist) / hray_res, 1.0);
          for (int a = 0; a < 3; a++)
            det.position[a] = R.center[a];
          if (out && total < cap)
            out[total] = det;
          total++;
        }
        if (n_out_per_frame)
          n_out_per_frame[f] = D.n;
      }
      if (trace)
        std::fprintf(stderr, "[vofod trace] n=%u device tail: sync %.3f end %.3f ms, %zu detections\n", n, tr_sync1, ms_since(t0), total);
      *n_out = total;
      if (total > cap)
        ret = VOFOD_ERR_CAPACITY;
      return ret;
    }
    if (!no_update && (fb & (vtd::TAIL_FB_DETS | vtd::TAIL_FB_EXPLORE)))
    {
      // A map-updating scan whose flood fills have already written their frontiers to the map: the tail cannot be run again.
      // More detections than the record slots hold (TP_MAXD per frame): everything needed is on the device - the clusters in
      // canonical order (d_tailc) and their explore results.  (A work list overflow cannot happen for radii the device accepts.)
      if (fb & vtd::TAIL_FB_EXPLORE)
      {
        h->err = "device tail: flood-fill work list overflow";
        return VOFOD_ERR_DEVICE;
      }
      std::vector<vtd::TailCluster> tc(vtd::TP_MAXC);
      std::vector<vc::ExploreResult> res(vtd::TP_MAXC);
      HIPCHK(hipMemcpy(tc.data(), ws.d_tailc, sizeof(vtd::TailCluster) * vtd::TP_MAXC, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(res.data(), h->explore.d_results, sizeof(vc::ExploreResult) * vtd::TP_MAXC, hipMemcpyDeviceToHost));  // (frame 0: result slots 0..TP_MAXC-1)
      size_t total = 0;
      const float* tf = tfs;
      for (int c = 0; c < vtd::TP_MAXC; c++)
      {
        if (tc[c].job < 0 || tc[c].job >= vtd::TP_MAXC || !res[tc[c].job].floating)
          continue;
        vofod_detection det{};
        const float d[3] = {tf[3] - tc[c].obb_center[0], tf[7] - tc[c].obb_center[1], tf[11] - tc[c].obb_center[2]};
        const double det_dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        det.id = h->last_detection_id++;
        det.frame = 0;
        det.n_points = tc[c].n_members;
        const float cov = static_cast<float>(std::sqrt(det_dist) * dp.output__position_sigma);
        for (int q = 0; q < 3; q++)
          det.covariance[4 * q] = cov;
        const double u = res[tc[c].job].conf_sum / tc[c].n_members;  // :860-865
        det.confidence = static_cast<float>(1.0 / std::exp(u));
        const double vray_res = sp.sensor_vfov / static_cast<double>(sp.sensor_vrays);
        const double hray_res = 2 * M_PI / static_cast<double>(sp.sensor_hrays);
        det.detection_probability = std::min(std::atan(1.0 / det_dist) / (vray_res * dp.classification__min_points), 1.0) * std::min(std::atan(1.0 / det_dist) / hray_res, 1.0);
        for (int a = 0; a < 3; a++)
          det.position[a] = tc[c].obb_center[a];
        if (out && total < cap)
          out[total] = det;
        total++;
      }
      if (n_out_per_frame)
        n_out_per_frame[0] = static_cast<uint32_t>(total);
      *n_out = total;
      return total > cap ? VOFOD_ERR_CAPACITY : ret;
    }
    // a frame exceeded a capacity of the device tail: the host tail redoes the batch from the full tables
    // (capacities of k_tail_prep - members, clusters, radius: no flood fill has run yet, also on a map-updating scan)
    ws.dtail = false;
    KLAUNCH(h, k_pack, fgrid(g, (std::max(SPEC_C, SPEC_M) + 255) / 256), dim3(256), g, ws.d_hdrs, ws.d_table, ws.d_cand, ws.va, ws.d_packed);
    HIPCHK(hipMemcpyAsync(ws.h_packed, ws.d_packed, sizeof(PackedFrame) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  // ---- tail: classifyClusters :961 + extractDetections :963.
  // Host: canonical cluster order, OBB + gates of the few candidate clusters.  Device (k_explore): the flood
  // fills and uncertainty sums, one wave per frame, jobs of a frame in the reference's order.
  const auto t_tail = clk::now();
  const float thr_frontiers = static_cast<float>(dp.voxel_map__thresholds__frontiers);
  struct FrameTail
  {
    std::vector<HostCluster> cl;
    vt::MemberIndex by_root;
    std::vector<int> job_of;  // per cluster: index into jobs or -1
    bool host_fallback = false;
  };
  std::vector<FrameTail> tails(n);
  std::vector<vc::ExploreJob> jobs;
  std::vector<uint32_t> job_begin(n + 1, 0);
  std::vector<int> job_members;
  const bool latches = h->background_pts_sufficient && h->sure_background_sufficient;
  // phase A (serial): frames whose tables overflowed the speculative read-back fetch the rest
  std::vector<std::vector<ClusterRec>> recs_big(n);
  std::vector<std::vector<CandMemberX>> members_big(n);
  std::vector<uint8_t> big_recs(n, 0), big_members(n, 0);
  for (uint32_t f = 0; f < n; f++)
  {
    FrameHdr& hdr = ws.h_packed[f].hdr;
    if (hdr.status != VOFOD_OK)
      ret = hdr.status;
    if (ws.lite)
    {
      // lite read-back: only the candidate clusters' records came back; the header's C becomes their number
      const PackedLite& L = ws.h_lite[f];
      PackedFrame& pf = ws.h_packed[f];
      if (L.n_recs <= LITE_C)
      {
        std::memcpy(pf.table, L.recs, sizeof(ClusterRec) * L.n_recs);
        static_assert(LITE_C <= SPEC_C && LITE_M <= SPEC_M, "the lite lists are unpacked into the packed slot");
      }
      else
      {
        // more candidate clusters than the lite slot holds: fetch the frame's whole table, keep the candidates
        std::vector<ClusterRec> all(hdr.C);
        HIPCHK(hipMemcpy(all.data(), ws.d_table + static_cast<size_t>(f) * ws.vox_cap, sizeof(ClusterRec) * hdr.C, hipMemcpyDeviceToHost));
        for (const ClusterRec& r : all)
          if (r.cand && !r.close)
            recs_big[f].push_back(r);
        big_recs[f] = 1;
      }
      hdr.C = L.n_recs;
      if (L.n_members <= LITE_M)
        std::memcpy(pf.members, L.members, sizeof(CandMemberX) * L.n_members);
      else
        big_members[f] = 1;
    }
    else
    {
      big_recs[f] = hdr.C > SPEC_C;
      big_members[f] = hdr.n_cand > SPEC_M;
      if (big_recs[f])
      {
        recs_big[f].resize(hdr.C);
        HIPCHK(hipMemcpy(recs_big[f].data(), ws.d_table + static_cast<size_t>(f) * ws.vox_cap, sizeof(ClusterRec) * hdr.C, hipMemcpyDeviceToHost));
      }
    }
    if (big_members[f])
    {
      members_big[f].resize(hdr.n_cand);
      CandMemberX* d_tmp = static_cast<CandMemberX*>(ws.d_members_big);  // n_cand <= V <= vox_cap: sized with the workspace
      KLAUNCH(h, k_gather_members, dim3((hdr.n_cand + 255) / 256), dim3(256), g, f, hdr.n_cand, ws.d_cand, ws.va, d_tmp);
      HIPCHK(hipMemcpyAsync(members_big[f].data(), d_tmp, sizeof(CandMemberX) * hdr.n_cand, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    }
  }
  // phase B (parallel over frames): canonical order, member index, boxes and gates, the frame's explore jobs
  std::vector<std::vector<vc::ExploreJob>> jobs_f(n);
  std::vector<std::vector<int>> members_f(n);
  auto prep_frame = [&](uint32_t f) {
    FrameTail& T = tails[f];
    const PackedFrame& pf = ws.h_packed[f];
    const FrameHdr& hdr = pf.hdr;
    const ClusterRec* recs = big_recs[f] ? recs_big[f].data() : pf.table;
    const CandMemberX* members = big_members[f] ? members_big[f].data() : pf.members;
    // canonical order: size desc, smallest member asc
    T.cl.resize(hdr.C);
    for (uint32_t c = 0; c < hdr.C; c++)
      T.cl[c].rec = recs[c];
    std::sort(T.cl.begin(), T.cl.end(), [](const HostCluster& a, const HostCluster& b) {
      if (a.rec.size != b.rec.size)
        return a.rec.size > b.rec.size;
      return a.rec.root < b.rec.root;
    });
    {
      std::vector<std::pair<uint64_t, vt::Member>> tmp(hdr.n_cand);
      for (uint32_t i = 0; i < hdr.n_cand; i++)
      {
        const CandMemberX& m = members[i];
        tmp[i] = {(static_cast<uint64_t>(m.root) << 32) | m.v, vt::Member{m.v, {m.x, m.y, m.z}, m.count}};
      }
      T.by_root.build(tmp);
    }
    T.job_of.assign(hdr.C, -1);
    const float* tf = tfs + 12 * f;
    const float tpos[3] = {tf[3], tf[7], tf[11]};
    std::vector<vc::ExploreJob>& jl = jobs_f[f];
    std::vector<int>& ml = members_f[f];
    // classify_cluster :1648-1690: boxes and gates
    for (uint32_t ci = 0; ci < hdr.C; ci++)
    {
      HostCluster& c = T.cl[ci];
      if (c.rec.close)
        continue;
      c.cclass = VOFOD_CLASS_INVALID;
      if (!c.rec.cand)
        continue;  // fails min_points or cannot pass max_size (device-side gate)
      const vt::MemberSpan mem = T.by_root.of(c.rec.root);
      c.boxes = vt::boxes_of(mem);
      c.evaluated = true;
      if (static_cast<int>(mem.size()) < dp.classification__min_points)
        continue;
      {
        const float d[3] = {tpos[0] - c.boxes.obb_center[0], tpos[1] - c.boxes.obb_center[1], tpos[2] - c.boxes.obb_center[2]};
        const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (dist > dp.classification__max_distance)
          continue;
      }
      {
        const float d[3] = {c.boxes.obb_max[0] - c.boxes.obb_min[0], c.boxes.obb_max[1] - c.boxes.obb_min[1], c.boxes.obb_max[2] - c.boxes.obb_min[2]};
        c.obb_size = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (c.obb_size > dp.classification__max_size)
          continue;
      }
      if (!latches)  // :1694, :1719-1722
      {
        c.cclass = VOFOD_CLASS_UNKNOWN;
        continue;
      }
      vc::ExploreJob job{};
      job.frame = f;
      job.n_members = static_cast<uint32_t>(mem.size());
      job.member_off = static_cast<uint32_t>(ml.size() / 3);  // rebased when the frames are concatenated
      job.R = static_cast<int>((c.obb_size + dp.classification__max_explore_distance) / sp.voxel_size);  // :1696
      for (const vt::Member& m : mem)
      {
        int o[3];
        h->hg.coordToIdx(m.p, o);
        ml.insert(ml.end(), o, o + 3);
      }
      int mn[3], mx[3];  // getSubmapCopy(aabb, inflate 2) voxel_map.cpp:550-559
      h->hg.coordToIdx(c.boxes.aabb_min, mn);
      h->hg.coordToIdx(c.boxes.aabb_max, mx);
      for (int a = 0; a < 3; a++)
      {
        job.box_lo[a] = std::clamp(mn[a] - 2, 0, h->hg.s[a] - 1);
        job.box_hi[a] = std::clamp(mx[a] + 2, 0, h->hg.s[a] - 1);
      }
      if (job.R > vc::EX_MAX_R || job.R < 0)
        T.host_fallback = true;
      T.job_of[ci] = static_cast<int>(jl.size());  // rebased below
      jl.push_back(job);
    }
    if (jl.size() > vc::EX_MAX_JOBS)
      T.host_fallback = true;
  };
  h->pool->parallel_for(n, prep_frame);
  // phase C (serial): concatenate the frames' job lists in frame order
  for (uint32_t f = 0; f < n; f++)
  {
    job_begin[f] = static_cast<uint32_t>(jobs.size());
    const uint32_t jbase = static_cast<uint32_t>(jobs.size()), mbase = static_cast<uint32_t>(job_members.size() / 3);
    for (vc::ExploreJob j : jobs_f[f])
    {
      j.member_off += mbase;
      j.result_slot = static_cast<uint32_t>(jobs.size());
      jobs.push_back(j);
    }
    job_members.insert(job_members.end(), members_f[f].begin(), members_f[f].end());
    for (int& ji : tails[f].job_of)
      if (ji >= 0)
        ji += static_cast<int>(jbase);
  }
  job_begin[n] = static_cast<uint32_t>(jobs.size());

  tr_prep = ms_since(t0);
  std::vector<vc::ExploreResult> results(jobs.size());
  const bool force_host = std::getenv("VOFOD_EXPLORE") && std::strcmp(std::getenv("VOFOD_EXPLORE"), "host") == 0;  // tests exercise the fallback
  bool any_host = force_host;
  for (const FrameTail& T : tails)
    any_host |= T.host_fallback;
  if (!jobs.empty() && !any_host)
  {
    r = ensure_explore(h, h->explore, h->ws.F, jobs.size(), job_members.size() / 3);
    if (r != VOFOD_OK)
      return r;
    // a collected async batch runs its tail on a second stream so that it does not queue behind the next batch's chain
    struct StreamSwap
    {
      vofod_handle* h;
      hipStream_t saved;
      StreamSwap(vofod_handle* h_, bool on) : h(h_), saved(h_->stream)
      {
        if (on)
          h->stream = h->stream_tail;
      }
      ~StreamSwap() { h->stream = saved; }
    } swap_guard(h, phase == FRAMES_COLLECT);
    ExploreBufs& eb = h->explore;
    if (h->ev_explore)
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_explore, 0));  // a device tail in flight may still use the flood-fill buffers
    HIPCHK(hipMemcpyAsync(eb.d_jobs, jobs.data(), sizeof(vc::ExploreJob) * jobs.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(eb.d_job_begin, job_begin.data(), sizeof(uint32_t) * (n + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(eb.d_members, job_members.data(), sizeof(int) * job_members.size(), hipMemcpyHostToDevice, h->stream));
    vc::ExploreParams ep{};
    ep.thr_unknown = thr_frontiers;
    ep.thr_ground = thr_new;
    ep.frontier_value = thr_frontiers;
    ep.ray_score = dp.voxel_map__scores__ray;
    ep.no_update = no_update;
    ep.stack_cap = vc::EX_CELLS;
    KLAUNCH(h, vc::k_explore, dim3(n), dim3(64), ep, h->mg, eb.d_jobs, eb.d_job_begin, eb.d_job_begin + 1, eb.d_members, h->d_map, eb.d_overlay, eb.d_stack, eb.d_explored, eb.d_touched,
            eb.d_ovl_list, eb.d_ovl_count, eb.d_results, eb.d_visited);
    HIPCHK(hipMemcpyAsync(results.data(), eb.d_results, sizeof(vc::ExploreResult) * jobs.size(), hipMemcpyDeviceToHost, h->stream));
    if (h->ev_explore)
      HIPCHK(hipEventRecord(h->ev_explore, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!no_update)
      h->mapbits_valid = false;
  }

  tr_explore = ms_since(t0);
  size_t total = 0;
  for (uint32_t f = 0; f < n; f++)
  {
    FrameTail& T = tails[f];
    const FrameHdr& hdr = ws.h_packed[f].hdr;
    const float* tf = tfs + 12 * f;
    const float tpos[3] = {tf[3], tf[7], tf[11]};
    uint32_t n_det_frame = 0;
    if (any_host && job_begin[f + 1] > job_begin[f])
    {
      // fallback (Manhattan radius or job count beyond the device kernel's limits): sequential host path over read-back boxes
      r = host_explore_frame(h, T.cl, T.by_root, T.job_of, jobs, results, no_update, thr_frontiers, thr_new, dp);
      if (r != VOFOD_OK)
        return r;
    }
    for (uint32_t ci = 0; ci < hdr.C; ci++)
    {
      HostCluster& c = T.cl[ci];
      const int ji = T.job_of[ci];
      if (ji < 0)
        continue;
      c.cclass = results[ji].floating ? VOFOD_CLASS_MAV : VOFOD_CLASS_UNKNOWN;
    }
    // extractDetections :834-879
    for (uint32_t ci = 0; ci < hdr.C; ci++)
    {
      HostCluster& c = T.cl[ci];
      if (c.rec.close || c.cclass != VOFOD_CLASS_MAV)
        continue;
      const vt::MemberSpan mem = T.by_root.of(c.rec.root);
      vofod_detection det{};
      const float d[3] = {tpos[0] - c.boxes.obb_center[0], tpos[1] - c.boxes.obb_center[1], tpos[2] - c.boxes.obb_center[2]};
      const double det_dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      det.id = h->last_detection_id++;
      det.frame = f;
      det.n_points = mem.size();
      const float cov = static_cast<float>(std::sqrt(det_dist) * dp.output__position_sigma);
      for (int q = 0; q < 3; q++)
        det.covariance[4 * q] = cov;
      const double u = results[T.job_of[ci]].conf_sum / mem.size();  // :860-865
      det.confidence = static_cast<float>(1.0 / std::exp(u));
      const double vray_res = sp.sensor_vfov / static_cast<double>(sp.sensor_vrays);
      const double hray_res = 2 * M_PI / static_cast<double>(sp.sensor_hrays);
      det.detection_probability = std::min(std::atan(1.0 / det_dist) / (vray_res * dp.classification__min_points), 1.0) * std::min(std::atan(1.0 / det_dist) / hray_res, 1.0);
      for (int a = 0; a < 3; a++)
        det.position[a] = c.boxes.obb_center[a];
      if (out && total < cap)
        out[total] = det;
      total++;
      n_det_frame++;
    }
    if (n_out_per_frame)
      n_out_per_frame[f] = n_det_frame;

    if (dbg)
    {
      vofod_scan_debug& d = dbg[f];
      d.n_input_after_crop = hdr.n_in;
      d.n_bg_voxels = h->n_bg_voxels;
      d.background_pts_sufficient = h->background_pts_sufficient;
      d.sure_background_sufficient = h->sure_background_sufficient;
      // far-only view (dbg[0].far_only): what the production path of a read-only batch computes - the far clusters, and labels
      // for their voxels only.  A frame that went through the full clustering all the same (no dilated image, more pure-far
      // bricks than the close-first path takes, VOFOD_CLOSE_FIRST=0) is cut down to that view here.
      const bool far_view = dbg[0].far_only != 0;
      uint32_t n_shown = hdr.C;
      if (far_view && !hdr.far_only)
      {
        n_shown = 0;
        for (uint32_t c = 0; c < hdr.C; c++)
          n_shown += T.cl[c].rec.close ? 0u : 1u;
      }
      d.n_weighted = hdr.V;
      d.n_clusters = n_shown;
      if ((d.weighted || d.labels) && d.weighted_cap < hdr.V)
        ret = VOFOD_ERR_CAPACITY;
      else
      {
        if (d.weighted && hdr.V)
          HIPCHK(hipMemcpy(d.weighted, ws.va.pts + static_cast<size_t>(f) * ws.vox_cap, sizeof(float4) * hdr.V, hipMemcpyDeviceToHost));
        if (d.labels && hdr.V)
        {
          HIPCHK(hipMemcpy(d.labels, ws.d_labels + static_cast<size_t>(f) * ws.vox_cap, sizeof(uint32_t) * hdr.V, hipMemcpyDeviceToHost));
          if (far_view && !hdr.far_only)
          {
            std::vector<uint32_t> far_roots;
            for (uint32_t c = 0; c < hdr.C; c++)
              if (!T.cl[c].rec.close)
                far_roots.push_back(T.cl[c].rec.root);
            std::sort(far_roots.begin(), far_roots.end());
            for (uint32_t v = 0; v < hdr.V; v++)
              if (!std::binary_search(far_roots.begin(), far_roots.end(), d.labels[v]))
                d.labels[v] = CF_LABEL_NONE;
          }
        }
      }
      if (d.clusters)
      {
        if (d.clusters_cap < n_shown)
          ret = VOFOD_ERR_CAPACITY;
        else
          for (uint32_t c = 0, c_out = 0; c < hdr.C; c++)
          {
            const HostCluster& hc = T.cl[c];
            if (far_view && hc.rec.close)
              continue;
            vofod_cluster_info& ci = d.clusters[c_out++];
            ci.first_member = hc.rec.root;
            ci.n_points = hc.rec.size;
            ci.is_close = hc.rec.close;
            ci.cclass = hc.cclass;
            for (int a = 0; a < 3; a++)
            {
              ci.aabb_min[a] = (static_cast<float>(hc.rec.imin[a]) + 0.5f) * g.leaf[a] + hdr.offset[a];
              ci.aabb_max[a] = (static_cast<float>(hc.rec.imax[a]) + 0.5f) * g.leaf[a] + hdr.offset[a];
              ci.obb_center[a] = hc.evaluated ? hc.boxes.obb_center[a] : NAN;
            }
            ci.obb_size = hc.obb_size;
          }
      }
      d.stage_ms[0] = dev_ms[0];
      d.stage_ms[1] = dev_ms[1];
      d.stage_ms[2] = dev_ms[2];
      d.stage_ms[3] = dev_ms[3];
      d.stage_ms[4] = ms_since(t_tail);
      d.stage_ms[5] = ms_since(t0);
    }
  }
  if (trace)
  {
    size_t sumC = 0, sumCand = 0, sumEval = 0;
    for (uint32_t f = 0; f < n; f++)
    {
      sumC += ws.h_packed[f].hdr.C;
      sumCand += ws.h_packed[f].hdr.n_cand;
      for (const auto& c : tails[f].cl)
        sumEval += c.evaluated;
    }
    std::fprintf(stderr, "[vofod trace] n=%u launch %.3f sync1 %.3f prep %.3f explore %.3f end %.3f ms jobs %zu C %zu cand_members %zu evaluated %zu | sort %.3f gates %.3f\n", n, tr_launch, tr_sync1,
                 tr_prep, tr_explore, ms_since(t0), jobs.size(), sumC, sumCand, sumEval, tr_a, tr_b);
    (void)tr_c;
  }
  *n_out = total;
  if (total > cap)
    ret = VOFOD_ERR_CAPACITY;
  return ret;
}

}  // namespace

#include "driver_aux.h"
#include "collective.h"
