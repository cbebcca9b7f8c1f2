"""Batched many-scan mode across the GPUs of one node.

Independent frames are sharded in contiguous blocks over the ranks (one process per GPU), every rank runs
`vofod_process_batch` on its own frames against its own replica of the voxel map, and the only collective is
one all-gather of fixed-size detection records at the end (RCCL over xGMI for CUDA tensors; gloo on CPU in the
tests).  The payload is tiny (D_MAX x 128 B + 8 B per frame), so the call is latency bound.
"""
from __future__ import annotations

import numpy as np

from . import capi

D_MAX = 16                      # detection records carried per frame
REC_F64 = capi.DETECTION.itemsize // 8   # a 128-byte vofod_detection viewed as 16 float64 words
FRAME_F64 = D_MAX * REC_F64 + 1          # + the frame's detection count


def shard_frames(n_frames: int, world: int, rank: int) -> range:
    """Contiguous block of frame indices owned by `rank` (frame f -> rank f // ceil(n/world))."""
    per = -(-n_frames // world)
    return range(min(rank * per, n_frames), min((rank + 1) * per, n_frames))


def pack_detections(dets: np.ndarray, per_frame: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
    """[n_frames, FRAME_F64] float64: up to D_MAX records per frame followed by the true count."""
    n = len(per_frame)
    if out is None:
        out = np.zeros((n, FRAME_F64), dtype=np.float64)
    else:
        out[:] = 0
    per_frame = np.asarray(per_frame)
    out.view(np.uint64)[:, -1] = per_frame  # the count as an integer word: the same bytes as vofod_pack_detection_slots (collective.h) writes
    if len(dets):
        raw = np.ascontiguousarray(dets).view(np.float64).reshape(-1, REC_F64)
        start = np.cumsum(per_frame) - per_frame  # first record of every frame
        for f in np.flatnonzero(per_frame):       # frames with detections are few: no loop over the batch
            m = min(int(per_frame[f]), D_MAX)
            out[f, : m * REC_F64] = raw[start[f] : start[f] + m].reshape(-1)
    return out


def unpack_detections(packed: np.ndarray, frame_offset: int = 0) -> np.ndarray:
    """Inverse of pack_detections for one rank's block; `frame` fields are rebased by frame_offset."""
    recs = []
    for f in range(packed.shape[0]):
        m = min(int(np.ascontiguousarray(packed).view(np.uint64)[f, -1]), D_MAX)
        if m:
            d = np.frombuffer(np.ascontiguousarray(packed[f, : m * REC_F64]).tobytes(), dtype=capi.DETECTION).copy()
            d["frame"] = frame_offset + f
            recs.append(d)
    return np.concatenate(recs) if recs else np.zeros(0, dtype=capi.DETECTION)


def allgather_detections(local, gathered=None):
    """One all-gather of the packed records.  `local`: torch tensor [frames_per_rank, FRAME_F64] (float64) on
    the device the process group communicates on.  Returns [world, frames_per_rank, FRAME_F64]."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if world == 1:
        gathered[0].copy_(local)
    else:
        # concatenated layout [world * frames, ...] is what every backend accepts; same memory as [world, frames, ...]
        dist.all_gather_into_tensor(gathered.view((world * local.shape[0],) + tuple(local.shape[1:])), local.contiguous())
    return gathered


class CabiComm:
    """The same all-gather through the product's C-ABI (vofod_comm_* / vofod_allgather_detections: RCCL called from
    libvofod_hip.so, what a C++ nodelet host links against).  `bootstrap(id_bytes) -> id_bytes` ships rank 0's 128-byte
    RCCL id to every rank (torch.distributed broadcast, MPI, a socket ...); with one rank nothing is shipped."""

    def __init__(self, lib: capi.Library, rank: int, world: int, device: int, bootstrap=None):
        import ctypes as C

        self.lib, self.rank, self.world = lib, rank, world
        ident = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            st = lib.comm_unique_id(capi.ptr(ident))
            if st != capi.OK:
                raise RuntimeError(f"vofod_comm_unique_id: status {st}: {lib.comm_last_error(None).decode()}")
        if world > 1:
            if bootstrap is None:
                raise ValueError("more than one rank needs a bootstrap function for the communicator id")
            ident = np.ascontiguousarray(bootstrap(ident), dtype=np.uint8)
        self.h = C.c_void_p()
        st = lib.comm_create(capi.ptr(ident), rank, world, device, C.byref(self.h))
        if st != capi.OK:
            raise RuntimeError(f"vofod_comm_create: status {st}: {lib.comm_last_error(None).decode()}")

    def allgather(self, dets: np.ndarray, per_frame: np.ndarray, d_max: int = D_MAX):
        """-> (records [world, frames, d_max] of capi.DETECTION, counts [world, frames])"""
        per = np.ascontiguousarray(per_frame, dtype=np.uint32)
        loc = np.ascontiguousarray(dets, dtype=capi.DETECTION)
        n = len(per)
        out = np.zeros((self.world, n, d_max), dtype=capi.DETECTION)
        cnt = np.zeros((self.world, n), dtype=np.uint32)
        st = self.lib.allgather_detections(self.h, capi.ptr(loc) if len(loc) else None, capi.ptr(per), n, d_max, capi.ptr(out), capi.ptr(cnt))
        if st != capi.OK:
            raise RuntimeError(f"vofod_allgather_detections: status {st}: {self.lib.comm_last_error(self.h).decode()}")
        return out, cnt

    def close(self):
        if self.h:
            self.lib.comm_destroy(self.h)
            self.h = None


def torch_bootstrap(device):
    """bootstrap for CabiComm over an initialised torch.distributed process group"""
    import torch
    import torch.distributed as dist

    def ship(ident: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(ident.copy()).to(device)
        dist.broadcast(t, src=0)
        return t.cpu().numpy()

    return ship
