"""The N>1 path on CPU: world_size-2 gloo, frames sharded across ranks, one all-gather of detection records."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _prepared(oracle):
    """every rank's replica of the map: apriori background + a few scans with raycast so the targets float in carved air"""
    from vofod_amd import capi, synth
    from helpers import make_pair

    det, _ = make_pair(oracle, oracle, "os1-128", 0.5)
    scene = synth.make_scene(21, n_targets=3)
    det.load_apriori(synth.apriori_points(scene, 0.5))
    for s in synth.scan_sequence(scene, "os1-128", 5, seed0=300):
        det.process_scan(s.scan, s.tf, flags=capi.SCAN_AUTO_RAYCAST)
    return det, synth.scan_sequence(scene, "os1-128", 6, seed0=310)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from vofod_amd import capi, dist as vdist, synth
    from helpers import make_pair

    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = capi.Library(ROOT / "oracle" / "libvofod_oracle.so", "vofod_oracle_")
    det, scans = _prepared(oracle)
    n_frames = len(scans)
    mine = vdist.shard_frames(n_frames, world, rank)
    dets, per = det.process_batch([scans[f].scan for f in mine], np.stack([scans[f].tf for f in mine]))
    local = torch.from_numpy(vdist.pack_detections(dets, per))
    gathered = vdist.allgather_detections(local).numpy()
    allrec = np.concatenate([vdist.unpack_detections(gathered[r], frame_offset=vdist.shard_frames(n_frames, world, r).start) for r in range(world)])
    np.save(Path(out_dir) / f"rank{rank}.npy", allrec)
    dist.destroy_process_group()


def test_sharded_batch_allgather_equals_single_process(oracle):
    import torch.multiprocessing as mp

    from vofod_amd import capi, dist as vdist, synth
    from helpers import make_pair

    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000)
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r0, r1 = np.load(Path(d) / "rank0.npy"), np.load(Path(d) / "rank1.npy")
    assert r0.tobytes() == r1.tobytes()  # every rank holds the same gathered set
    det, scans = _prepared(oracle)
    dets, per = det.process_batch([s.scan for s in scans], np.stack([s.tf for s in scans]))
    assert len(r0) == len(dets) > 0
    for k in ("frame", "n_points"):
        np.testing.assert_array_equal(r0[k], dets[k])
    np.testing.assert_array_equal(r0["position"], dets["position"])
    np.testing.assert_array_equal(r0["confidence"], dets["confidence"])
    # detection ids are per-handle counters, i.e. per rank in the sharded run
    assert vdist.shard_frames(6, 2, 0) == range(0, 3) and vdist.shard_frames(6, 2, 1) == range(3, 6)
    assert vdist.shard_frames(5, 4, 3) == range(5, 5)


def test_pack_unpack_roundtrip():
    from vofod_amd import capi, dist as vdist

    d = np.zeros(3, dtype=capi.DETECTION)
    d["id"] = [7, 8, 9]
    d["frame"] = [0, 2, 2]
    d["confidence"] = [0.5, 0.25, 1.0]
    d["position"] = np.arange(9).reshape(3, 3)
    packed = vdist.pack_detections(d, np.array([1, 0, 2], dtype=np.uint32))
    assert packed.shape == (3, vdist.FRAME_F64) and packed.view(np.uint64)[:, -1].tolist() == [1, 0, 2]
    back = vdist.unpack_detections(packed)
    assert back.tobytes() == d.tobytes()


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_cabi_slot_pack_unpack_matches_the_python_caller(n_ranks):
    """vofod_allgather_detections' pack / unpack loops (collective.h), as the plain host functions they are: byte for byte the
    payload vofod_amd/dist.py builds, for 2 and 8 ranks, with a frame whose detections exceed d_max (truncated slot, true
    count kept) and ranks without any detection.  The all-gather itself is a concatenation of the ranks' buffers."""
    import ctypes as C

    import vofod_amd
    from vofod_amd import capi, dist as vdist

    lib = vofod_amd.library()  # host functions only: no GPU needed
    rng = np.random.default_rng(n_ranks)
    frames, d_max = 5, vdist.D_MAX
    slot = lib.detection_slot_bytes(d_max)
    assert slot == d_max * 128 + 8 == vdist.FRAME_F64 * 8
    wire, want, want_counts = [], [], []
    for r in range(n_ranks):
        per = rng.integers(0, 4, frames).astype(np.uint32)
        if r == 1:
            per[:] = 0  # a rank without detections
        if r == 0:
            per[2] = d_max + 3  # more than the slot holds
        dets = np.zeros(int(per.sum()), dtype=capi.DETECTION)
        dets["id"] = rng.integers(0, 1 << 31, len(dets))
        dets["frame"] = np.repeat(np.arange(frames), per)
        dets["n_points"] = rng.integers(1, 99, len(dets))
        dets["confidence"] = rng.random(len(dets))
        dets["position"] = rng.normal(size=(len(dets), 3))
        buf = np.zeros(frames * slot, dtype=np.uint8)
        st = lib.pack_detection_slots(capi.ptr(dets) if len(dets) else None, capi.ptr(per), frames, d_max, capi.ptr(buf))
        assert st == capi.OK
        assert buf.tobytes() == vdist.pack_detections(dets, per).tobytes()
        wire.append(buf)
        start = np.cumsum(per) - per
        for f in range(frames):
            blk = np.zeros(d_max, dtype=capi.DETECTION)
            m = min(int(per[f]), d_max)
            blk[:m] = dets[start[f] : start[f] + m]
            want.append(blk)
        want_counts.append(per)
    gathered = np.concatenate(wire)
    out = np.zeros(n_ranks * frames * d_max, dtype=capi.DETECTION)
    counts = np.zeros(n_ranks * frames, dtype=np.uint32)
    assert lib.unpack_detection_slots(capi.ptr(gathered), n_ranks * frames, d_max, capi.ptr(out), capi.ptr(counts)) == capi.OK
    assert out.tobytes() == np.concatenate(want).tobytes()
    np.testing.assert_array_equal(counts, np.concatenate(want_counts))
    assert counts[2] == d_max + 3  # the truncated slot still tells its true count
    # argument checks
    assert lib.pack_detection_slots(None, capi.ptr(np.ones(1, dtype=np.uint32)), 1, d_max, capi.ptr(np.zeros(slot, dtype=np.uint8))) == capi.ERR_INVALID_ARG
    assert lib.unpack_detection_slots(None, 1, d_max, capi.ptr(out), capi.ptr(counts)) == capi.ERR_INVALID_ARG
