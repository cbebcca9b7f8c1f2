// RCCL all-gather of the batched mode's detection records behind the C-ABI (include/vofod.h): every rank
// contributes frames_per_rank fixed-size slots (d_max 128-byte vofod_detection records + an 8-byte count word per frame),
// one ncclAllGather over xGMI returns all ranks' slots.  RCCL is loaded at run time (dlopen): the library does not
// depend on it unless the collective is used.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vofod.h"

namespace vcoll
{

// the few RCCL entry points used (signatures of rccl.h; ncclUniqueId = 128 opaque bytes, ncclChar = 0, ncclSuccess = 0)
struct UniqueId
{
  char internal[128];
};
using Comm = void*;
struct Api
{
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* so = nullptr;
  std::string err;
  bool load()
  {
    if (so)
      return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((so = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
        break;
    if (!so)
    {
      err = std::string("RCCL not found: ") + dlerror();
      return false;
    }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(so, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(so, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(so, "ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(so, "ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(so, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather)
    {
      err = "RCCL library lacks an expected entry point";
      return false;
    }
    return true;
  }
};

inline Api& api()
{
  static Api a;
  return a;
}

// one mutex for loading the library and for the process-wide error string
inline std::mutex& load_mutex()
{
  static std::mutex m;
  return m;
}

}  // namespace vcoll

struct vofod_comm
{
  vcoll::Comm comm = nullptr;
  int rank = 0, n_ranks = 1, device = 0;
  hipStream_t stream = nullptr;
  char *d_send = nullptr, *d_recv = nullptr, *h_stage = nullptr;  // h_stage: pinned, send slot followed by the receive area
  size_t cap_bytes = 0;  // per-rank payload the buffers are sized for
  std::mutex mtx;
  std::string err;
};

extern "C" {

int vofod_comm_unique_id(uint8_t id[VOFOD_COMM_ID_BYTES])
{
  if (!id)
    return VOFOD_ERR_INVALID_ARG;
  static_assert(VOFOD_COMM_ID_BYTES == sizeof(vcoll::UniqueId), "ncclUniqueId is 128 bytes");
  std::scoped_lock lck(vcoll::load_mutex());
  if (!vcoll::api().load())
    return VOFOD_ERR_DEVICE;
  vcoll::UniqueId u;
  if (vcoll::api().GetUniqueId(&u) != 0)
    return VOFOD_ERR_DEVICE;
  std::memcpy(id, u.internal, sizeof(u.internal));
  return VOFOD_OK;
}

int vofod_comm_create(const uint8_t id[VOFOD_COMM_ID_BYTES], int32_t rank, int32_t n_ranks, int32_t device, vofod_comm** out)
{
  if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return VOFOD_ERR_INVALID_ARG;
  *out = nullptr;
  {
    std::scoped_lock lck(vcoll::load_mutex());
    if (!vcoll::api().load())
      return VOFOD_ERR_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess)
  {
    std::scoped_lock lck(vcoll::load_mutex());
    vcoll::api().err = "hipSetDevice failed";
    return VOFOD_ERR_DEVICE;
  }
  auto* c = new vofod_comm;
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->device = device;
  vcoll::UniqueId u;
  std::memcpy(u.internal, id, sizeof(u.internal));
  const bool comm_ok = vcoll::api().CommInitRank(&c->comm, n_ranks, u, rank) == 0;
  if (!comm_ok)
    c->comm = nullptr;
  if (!comm_ok || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
  {
    {
      std::scoped_lock lck(vcoll::load_mutex());
      vcoll::api().err = comm_ok ? "hipStreamCreate failed" : "ncclCommInitRank failed";
    }
    c->stream = nullptr;
    vofod_comm_destroy(c);  // releases the communicator when only the stream failed
    return VOFOD_ERR_DEVICE;
  }
  *out = c;
  return VOFOD_OK;
}

void vofod_comm_destroy(vofod_comm* c)
{
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  if (c->comm)
    (void)vcoll::api().CommDestroy(c->comm);
  if (c->d_send)
    (void)hipFree(c->d_send);
  if (c->d_recv)
    (void)hipFree(c->d_recv);
  if (c->h_stage)
    (void)hipHostFree(c->h_stage);
  if (c->stream)
    (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* vofod_comm_last_error(vofod_comm* c) { return c ? c->err.c_str() : vcoll::api().err.c_str(); }

// The exchange's wire format, as two plain host functions (no device, no communicator: unit-tested on the CPU for 2 and 8
// ranks, byte for byte against vofod_amd/dist.py's pack_detections; a caller with a transport of its own - MPI, gloo - can use
// them directly).  Slot of a frame: d_max records of 128 bytes (the frame's first detections in order, zero padded), then the
// frame's TRUE count as a 64-bit word (D_max * 128 + 8 bytes).
size_t vofod_detection_slot_bytes(size_t d_max) { return d_max * sizeof(vofod_detection) + 8; }

int vofod_pack_detection_slots(const vofod_detection* local, const uint32_t* n_per_frame, size_t frames, size_t d_max, void* slots)
{
  if (!n_per_frame || !slots || d_max == 0)
    return VOFOD_ERR_INVALID_ARG;
  static_assert(sizeof(vofod_detection) == 128, "Detection.msg record: 128 bytes");
  const size_t slot = vofod_detection_slot_bytes(d_max);
  char* out = static_cast<char*>(slots);
  std::memset(out, 0, frames * slot);
  size_t next = 0;
  for (size_t f = 0; f < frames; f++)
  {
    char* s = out + f * slot;
    const uint32_t cnt = n_per_frame[f];
    const uint32_t keep = static_cast<uint32_t>(std::min<size_t>(cnt, d_max));
    if (keep)
    {
      if (!local)
        return VOFOD_ERR_INVALID_ARG;
      std::memcpy(s, local + next, keep * sizeof(vofod_detection));
    }
    std::memcpy(s + d_max * sizeof(vofod_detection), &cnt, 4);
    next += cnt;
  }
  return VOFOD_OK;
}

int vofod_unpack_detection_slots(const void* slots, size_t frames_total, size_t d_max, vofod_detection* all, uint32_t* all_counts)
{
  if (!slots || !all || !all_counts || d_max == 0)
    return VOFOD_ERR_INVALID_ARG;
  const size_t slot = vofod_detection_slot_bytes(d_max);
  const char* in = static_cast<const char*>(slots);
  for (size_t q = 0; q < frames_total; q++)
  {
    const char* s = in + q * slot;
    std::memcpy(all + q * d_max, s, d_max * sizeof(vofod_detection));
    std::memcpy(all_counts + q, s + d_max * sizeof(vofod_detection), 4);
  }
  return VOFOD_OK;
}

int vofod_allgather_detections(vofod_comm* c, const vofod_detection* local, const uint32_t* n_per_frame, size_t frames_per_rank, size_t d_max, vofod_detection* all, uint32_t* all_counts)
{
  if (!c || !n_per_frame || !all || !all_counts || d_max == 0)
    return VOFOD_ERR_INVALID_ARG;
  if (!local)  // allowed only when this rank has no detection at all
    for (size_t f = 0; f < frames_per_rank; f++)
      if (n_per_frame[f])
        return VOFOD_ERR_INVALID_ARG;
  std::scoped_lock lck(c->mtx);
  if (hipSetDevice(c->device) != hipSuccess)
    return VOFOD_ERR_DEVICE;
  static_assert(sizeof(vofod_detection) == 128, "Detection.msg record: 128 bytes");
  const size_t slot = vofod_detection_slot_bytes(d_max);  // records + count word per frame
  const size_t bytes = frames_per_rank * slot;
  if (bytes == 0)
    return VOFOD_OK;
#define COLLCHK(expr)                                             \
  do                                                              \
  {                                                               \
    if ((expr) != hipSuccess)                                     \
    {                                                             \
      c->err = std::string(#expr) + " failed";                    \
      return VOFOD_ERR_DEVICE;                                    \
    }                                                             \
  } while (0)
  if (bytes > c->cap_bytes)
  {
    if (c->d_send)
      (void)hipFree(c->d_send);
    if (c->d_recv)
      (void)hipFree(c->d_recv);
    if (c->h_stage)
      (void)hipHostFree(c->h_stage);
    c->d_send = c->d_recv = c->h_stage = nullptr;
    c->cap_bytes = 0;
    COLLCHK(hipMalloc(reinterpret_cast<void**>(&c->d_send), bytes));
    COLLCHK(hipMalloc(reinterpret_cast<void**>(&c->d_recv), bytes * c->n_ranks));
    COLLCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), bytes * (c->n_ranks + 1)));
    c->cap_bytes = bytes;
  }
  // pack: the frame's detections (in order) at the head of its slot, the count behind them
  if (const int pr = vofod_pack_detection_slots(local, n_per_frame, frames_per_rank, d_max, c->h_stage); pr != VOFOD_OK)
    return pr;
  COLLCHK(hipMemcpyAsync(c->d_send, c->h_stage, bytes, hipMemcpyHostToDevice, c->stream));
  if (const int r = vcoll::api().AllGather(c->d_send, c->d_recv, bytes, 0 /* ncclChar */, c->comm, c->stream); r != 0)
  {
    c->err = std::string("ncclAllGather: ") + (vcoll::api().GetErrorString ? vcoll::api().GetErrorString(r) : "error");
    return VOFOD_ERR_DEVICE;
  }
  char* h_recv = c->h_stage + bytes;
  COLLCHK(hipMemcpyAsync(h_recv, c->d_recv, bytes * c->n_ranks, hipMemcpyDeviceToHost, c->stream));
  COLLCHK(hipStreamSynchronize(c->stream));
#undef COLLCHK
  if (const int ur = vofod_unpack_detection_slots(h_recv, static_cast<size_t>(c->n_ranks) * frames_per_rank, d_max, all, all_counts); ur != VOFOD_OK)
    return ur;
  return VOFOD_OK;
}

}  // extern "C"
