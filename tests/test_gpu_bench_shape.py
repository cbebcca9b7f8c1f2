"""GPU parity at the benchmarked shapes: the code path bench.py times is compared with the CPU
oracle on the very frames bench.py generates, and every BASELINE.json configuration gets an oracle comparison at
its full size.  Bit-exact on voxel indices / weights / labels / cluster tables; stated tolerances on OBB-derived
floats, confidences and on the float-atomic raycast accumulation (as in test_gpu_parity.py)."""
import os

import numpy as np
import pytest

from vofod_amd import capi, synth
from vofod_amd.detector import ScanData, cluster, load_cloud, voxel_grid_counted

from helpers import assert_detections_equal, assert_scan_debug_equal, make_pair, sync_maps

pytestmark = pytest.mark.gpu


def _hand_over_map(src, dst):
    """the oracle starts from the map (and the latches) the HIP detector warmed, as bench.py's cpu_baseline does"""
    st = src.status()
    if st.background_pts_sufficient and st.sure_background_sufficient:
        dst.load_apriori(np.zeros((0, 3), dtype=np.float32))  # sets both latches, touches no voxel
    sync_maps(src, dst)


def _compare_batches(ref, dev, scans, tfs, chunk=32, clusters_cap=8192):
    """HIP: one batch (the path under test); oracle: the same frames in chunks (bounded host memory)"""
    n = len(scans)
    db, pb, gb = dev.process_batch(scans, tfs, debug=True, clusters_cap=clusters_cap)
    n_det = 0
    for f0 in range(0, n, chunk):
        f1 = min(f0 + chunk, n)
        da, pa, ga = ref.process_batch(scans[f0:f1], tfs[f0:f1], debug=True, clusters_cap=clusters_cap)
        np.testing.assert_array_equal(pb[f0:f1], pa)
        sel = db[(db["frame"] >= f0) & (db["frame"] < f1)].copy()
        sel["frame"] -= f0
        if len(da):
            sel["id"] += da["id"][0] - sel["id"][0]  # ids run over the whole batch on the HIP side
        assert_detections_equal(da, sel)
        for k, (x, y) in enumerate(zip(ga, gb[f0:f1])):
            try:
                assert_scan_debug_equal(x, y)
            except AssertionError as e:
                raise AssertionError(f"frame {f0 + k}: {e}") from e
        n_det += len(da)
    return gb, n_det, db


def _compare_far_views(ref, dev, scans, tfs, chunk=32, clusters_cap=8192):
    """the far-only debug view of the HIP path (the close-first kernel's own output) against the far part of the oracle's full
    clustering, frame by frame"""
    from helpers import far_view

    n = len(scans)
    db, pb, gb = dev.process_batch(scans, tfs, debug=True, clusters_cap=clusters_cap, far_only=True)
    for f0 in range(0, n, chunk):
        f1 = min(f0 + chunk, n)
        da, pa, ga = ref.process_batch(scans[f0:f1], tfs[f0:f1], debug=True, clusters_cap=clusters_cap)
        np.testing.assert_array_equal(pb[f0:f1], pa)
        for k, (x, y) in enumerate(zip(ga, gb[f0:f1])):
            try:
                assert_scan_debug_equal(far_view(x), y)
            except AssertionError as e:
                raise AssertionError(f"far view, frame {f0 + k}: {e}") from e


@pytest.mark.parametrize("fallback", ["", "VOFOD_CLOSE_FIRST=0", "VOFOD_DEVICE_TAIL=0", "VOFOD_LDS_MAX_BRICKS=4096"])
def test_bench_workload_256_frames_os1_128(oracle, hip, fallback, monkeypatch):
    """configs[3] on one GPU = the bench.py default: 256 x OS1-128 @ 0.25 m, the warmed map, vofod_batch_submit/collect
    (k_frame_lds_far, which reads the input itself, -> k_tail_far), every frame against the oracle - by default and under each production fallback:
    the full clustering (what a cold map takes), the host tail (what a capacity of the device tail falls back to), and frames
    beyond the LDS image (every frame here holds more than 4096 bricks: the batch is run again on the
    global-memory kernels).  The switches are read on every call (vofod_hip.hip switch_off)."""
    if fallback:
        k, v = fallback.split("=")
        monkeypatch.setenv(k, v)
    F = 256
    ref, dev = make_pair(oracle, hip, "os1-128", 0.25, max_batch=F)
    scene = synth.bench_scene()
    synth.warm_map(dev, scene, "os1-128", 24)  # 96 in bench.py; 24 scans leave the same kind of map and keep the test short
    _hand_over_map(dev, ref)
    frames = synth.bench_frames(scene, "os1-128", F)
    scans = [s.scan for s in frames]
    tfs = np.stack([s.tf for s in frames])
    gb, n_det, db_full = _compare_batches(ref, dev, scans, tfs)
    assert min(len(g["weighted"]) for g in gb) > 15_000 and max(g["n_input_after_crop"] for g in gb) > 49_152
    if not fallback:
        _compare_far_views(ref, dev, scans, tfs)
    # the pipelined form bench.py uses: two tickets in flight give the same detections as the synchronous call
    dev.reserve(2)  # (workspaces of both tickets allocated here, not inside the submits)
    dev.lib.profile_enable(dev.h, 1)
    want = dev.process_batch(scans, tfs)
    names = _profiled_kernels(dev.lib, dev)
    dev.lib.profile_enable(dev.h, 0)
    if not os.environ.get("VOFOD_TEST_HARNESS_SELFCHECK"):
        expect = {"": "k_frame_lds_far", "VOFOD_CLOSE_FIRST=0": "k_frame_lds_full", "VOFOD_DEVICE_TAIL=0": "k_pack_lite", "VOFOD_LDS_MAX_BRICKS=4096": "k_brick_root"}[fallback]
        assert expect in names, (fallback, names)
        if fallback == "":
            assert "k_tail_far" in names, names
    # (the calls without debug output read back the lite slots - candidate clusters only; the debug call above the full tables)
    np.testing.assert_array_equal(want[0]["n_points"], db_full["n_points"])
    np.testing.assert_array_equal(want[0]["frame"], db_full["frame"])
    np.testing.assert_array_equal(want[0]["position"], db_full["position"])
    t0 = dev.batch_submit(scans, tfs)
    t1 = dev.batch_submit(scans, tfs)
    g0, g1 = dev.batch_collect(t0), dev.batch_collect(t1)
    for got in (g0, g1):
        np.testing.assert_array_equal(got[1], want[1])
        assert len(got[0]) == len(want[0])
        np.testing.assert_array_equal(got[0]["n_points"], want[0]["n_points"])
        np.testing.assert_array_equal(got[0]["position"], want[0]["position"])

def test_device_tail_at_the_bench_shape_with_many_targets(oracle, hip):
    """The classification tail ON THE DEVICE (k_tail_prep / k_explore / k_tail_finish: the path bench.py times, debug output
    off) at 256 x OS1-128 @ 0.25 m, compared field by field with the oracle's detections - on a scene whose twelve floating
    targets appear after the map was warmed, so that the tail kernels carry dozens of candidate clusters per batch."""
    F = 256
    ref, dev = make_pair(oracle, hip, "os1-128", 0.25, max_batch=F)
    warm_scene = synth.make_scene(synth.BENCH_SCENE_SEED, n_targets=0)
    scene = synth.make_scene(synth.BENCH_SCENE_SEED, n_targets=12)  # same buildings (drawn first from the seed), new targets
    np.testing.assert_array_equal(scene.boxes[: scene.n_static], warm_scene.boxes[: warm_scene.n_static])
    synth.warm_map(dev, warm_scene, "os1-128", 24)
    _hand_over_map(dev, ref)
    st = dev.status()
    assert st.background_pts_sufficient and st.sure_background_sufficient  # both latches: classify_cluster runs its flood fills
    frames = synth.bench_frames(scene, "os1-128", F)
    scans = [s.scan for s in frames]
    tfs = np.stack([s.tf for s in frames])
    want, want_per = [], []
    for f0 in range(0, F, 32):  # the oracle, 32 frames at a time (no debug output: detections only)
        d, per = ref.process_batch(scans[f0 : f0 + 32], tfs[f0 : f0 + 32])
        d = d.copy()
        d["frame"] += f0
        want.append(d)
        want_per.append(per)
    want = np.concatenate(want)
    want_per = np.concatenate(want_per)
    assert len(want) >= 50, len(want)
    want["id"] = np.arange(len(want), dtype=want["id"].dtype)  # ids run over the whole batch on the HIP side
    lib = dev.lib
    lib.profile_enable(dev.h, 1)
    got, got_per = dev.process_batch(scans, tfs)  # synchronous, no debug: device tail
    names = _profiled_kernels(lib, dev)
    lib.profile_enable(dev.h, 0)
    if os.environ.get("VOFOD_DEVICE_TAIL") != "0" and not os.environ.get("VOFOD_TEST_HARNESS_SELFCHECK"):
        # close-first batches: the fused tail; VOFOD_CLOSE_FIRST=0 (full clustering): the three tail kernels of rounds 2-3
        assert "k_tail_far" in names or ("k_tail_prep" in names and "k_explore" in names and "k_tail_finish" in names), names
    id0 = got["id"][0]
    got = got.copy()
    got["id"] -= id0
    np.testing.assert_array_equal(got_per, want_per)
    assert_detections_equal(want, got)
    # the pipelined form (what bench.py times): three tickets in flight
    tickets = [dev.batch_submit(scans, tfs) for _ in range(3)]
    for t in tickets:
        g, per = dev.batch_collect(t)
        g = g.copy()
        g["id"] -= g["id"][0]
        np.testing.assert_array_equal(per, want_per)
        assert_detections_equal(want, g)


def _dense_scan(sensor, seed, extent, empty=False, zlo=-2.0, zhi=1.0):
    h, w, _, _ = synth.SENSORS[sensor]
    n = h * w
    rng = np.random.default_rng(seed)
    if empty:
        x = y = z = np.zeros(n, dtype=np.float32)
    else:
        x = rng.uniform(-extent, extent, n).astype(np.float32)
        y = rng.uniform(-extent, extent, n).astype(np.float32)
        z = rng.uniform(zlo, zhi, n).astype(np.float32)
    return ScanData(x=x, y=y, z=z.copy(), width=w, height=h, stride_bytes=4)


def test_large_os1_128_batch_with_dense_and_empty_frames(oracle, hip):
    """>= 128 frames of OS1-128: ordinary scans mixed with frames of > 49 152 and > 100 000 surviving points (key lists
    beyond what a workgroup keeps in registers / LDS staging), thousands of extras per voxel, an empty frame"""
    F = 132
    ref, dev = make_pair(oracle, hip, "os1-128", 0.25, max_batch=F)
    scene = synth.bench_scene()
    for d in (ref, dev):
        synth.seed_ground(d)
    frames = synth.bench_frames(scene, "os1-128", 8)
    pose = synth.make_pose(3)
    special = {5: _dense_scan("os1-128", 1, 12.0), 40: _dense_scan("os1-128", 2, 3.0), 77: _dense_scan("os1-128", 3, 1.0, empty=True),
               100: _dense_scan("os1-128", 4, 9.0, zlo=-1.0, zhi=0.5), 131: _dense_scan("os1-128", 5, 6.0)}
    scans, tfs = [], []
    for f in range(F):
        if f in special:
            scans.append(special[f])
            tfs.append(pose)
        else:
            scans.append(frames[f % 8].scan)
            tfs.append(frames[f % 8].tf)
    tfs = np.stack(tfs).astype(np.float32)
    gb, _, _ = _compare_batches(ref, dev, scans, tfs, clusters_cap=65536)
    assert gb[77]["n_input_after_crop"] == 0 and len(gb[77]["weighted"]) == 0
    assert gb[5]["n_input_after_crop"] > 100_000
    assert gb[40]["n_input_after_crop"] - len(gb[40]["weighted"]) > 50_000
    assert 49_152 < gb[131]["n_input_after_crop"]


def test_large_batch_falls_back_per_batch_when_a_frame_overflows_lds(oracle, hip):
    """a frame with more occupied bricks than the LDS clustering takes: that batch runs on the global-memory kernels,
    the next (ordinary) batch is back on the LDS path and both equal the oracle"""
    F = 128
    ref, dev = make_pair(oracle, hip, "os1-128", 0.25, max_batch=F)
    scene = synth.bench_scene()
    for d in (ref, dev):
        synth.seed_ground(d)
    frames = synth.bench_frames(scene, "os1-128", 4)
    pose = synth.make_pose(3)
    wide = _dense_scan("os1-128", 4, 25.0)  # ~10 k occupied bricks
    scans = [frames[f % 4].scan for f in range(F)]
    tfs = np.stack([frames[f % 4].tf for f in range(F)]).astype(np.float32)
    scans_w, tfs_w = list(scans), tfs.copy()
    scans_w[17], tfs_w[17] = wide, pose
    lib = dev.lib
    for batch, (sc, tf) in enumerate(((scans_w, tfs_w), (scans, tfs))):
        lib.profile_enable(dev.h, 1)
        _compare_batches(ref, dev, sc, tf, clusters_cap=65536)
        names = _profiled_kernels(lib, dev)
        lib.profile_enable(dev.h, 0)
        lds_off = os.environ.get("VOFOD_CCL") == "voxel" or os.environ.get("VOFOD_BRICK_LDS") == "0"  # (no LDS clustering: no frame kernel either)
        if batch == 1 and not lds_off:  # (tools/run_fallback_matrix.sh switches the LDS kernels off altogether)
            assert any(n.startswith("k_frame_lds") for n in names), names  # no permanent latch


def _profiled_kernels(lib, det):
    import ctypes as C

    names = (C.c_char * (64 * 96))()
    ms = (C.c_double * 96)()
    calls = (C.c_uint64 * 96)()
    n = lib.profile_read(det.h, names, ms, calls, 96)
    return [names[64 * i : 64 * i + 64].split(b"\0", 1)[0].decode() for i in range(n)]


def test_config3_apriori_map_1m_voxels_at_025(oracle, hip):
    """configs[2]: OS1-128 @ 0.25 m with a 1 M-voxel apriori (+inf) background, latches set: kNN-against-background
    classification, flood fill and detections, sequential scans with map update + a read-only batch"""
    ref, dev = make_pair(oracle, hip, "os1-128", 0.25, max_batch=8)
    scene = synth.bench_scene()
    ap = synth.apriori_points(scene, 0.25, n_voxels=1_000_000, solid_ground_to=-1.2)  # terrain volume + building shells
    assert ap.shape[0] == 1_000_000
    for d in (ref, dev):
        d.load_apriori(ap)
    np.testing.assert_array_equal(dev.read_map(), ref.read_map())
    assert np.isinf(ref.read_map()).sum() > 900_000  # ~1 M distinct voxels
    n_det = 0
    for s in synth.scan_sequence(scene, "os1-128", 3, seed0=1000):
        dr, gr = ref.process_scan(s.scan, s.tf, debug=True)
        dh, gh = dev.process_scan(s.scan, s.tf, debug=True)
        assert_scan_debug_equal(gr, gh)
        assert_detections_equal(dr, dh)
        n_det += len(dr)
        np.testing.assert_array_equal(dev.read_map(capi.MAP_VOXELS), ref.read_map(capi.MAP_VOXELS))
        np.testing.assert_array_equal(dev.read_map(capi.MAP_FLAGS), ref.read_map(capi.MAP_FLAGS))
    frames = synth.bench_frames(scene, "os1-128", 8)
    gb, nb, _ = _compare_batches(ref, dev, [s.scan for s in frames], np.stack([s.tf for s in frames]))
    # classification ran (far clusters exist and were classified); detections themselves are rare on this scene
    assert sum(int((g["clusters"]["is_close"] == 0).sum()) for g in gb) > 0
    assert gb[0]["background_pts_sufficient"] and gb[0]["sure_background_sufficient"]


def test_config5_os2_128x2048_at_01(oracle, hip):
    """configs[4]: one dense OS2-128 x 2048 scan at 0.1 m voxels (M = 301.7 M): weighted grid + clustering + map update,
    then raycast + update sweep, against the oracle (3 x 1.2 GB of host maps on either side)"""
    sensor = "os2-128x2048"
    ref, dev = make_pair(oracle, hip, sensor, 0.1)
    assert dev.n_voxels == 301_752_451
    scene = synth.bench_scene()
    s0, s1 = synth.scan_sequence(scene, sensor, 2, seed0=1000)
    dr, gr = ref.process_scan(s0.scan, s0.tf, debug=True)
    dh, gh = dev.process_scan(s0.scan, s0.tf, debug=True)
    assert_scan_debug_equal(gr, gh)
    assert_detections_equal(dr, dh)
    assert len(gr["weighted"]) > 40_000
    ma = ref.read_map(capi.MAP_VOXELS)
    np.testing.assert_array_equal(dev.read_map(capi.MAP_VOXELS), ma)
    del ma
    np.testing.assert_array_equal(dev.read_map(capi.MAP_FLAGS), ref.read_map(capi.MAP_FLAGS))
    assert ref.raycast_begin(s0.scan, s0.tf) == dev.raycast_begin(s0.scan, s0.tf) == capi.OK
    ra, rb = ref.read_map(capi.MAP_RAYCAST), dev.read_map(capi.MAP_RAYCAST)
    assert np.count_nonzero(ra) > 1_000_000
    # tolerance: float-atomic accumulation order.  2048 columns at 0.1 m put ~10^4 ray segments into the voxels
    # next to the sensor: 2e-4 relative here (2e-5 holds for the 1024-column sensors of test_gpu_parity.py)
    np.testing.assert_allclose(rb, ra, rtol=2e-4, atol=2e-6)
    del ra, rb
    dr, gr = ref.process_scan(s1.scan, s1.tf, debug=True)
    dh, gh = dev.process_scan(s1.scan, s1.tf, debug=True)
    assert_scan_debug_equal(gr, gh)
    assert ref.raycast_finish() == dev.raycast_finish() == capi.OK
    ma, mb = ref.read_map(capi.MAP_VOXELS), dev.read_map(capi.MAP_VOXELS)
    fin = np.isfinite(ma)
    np.testing.assert_array_equal(np.isfinite(mb), fin)
    np.testing.assert_allclose(mb[fin], ma[fin], rtol=1e-4, atol=1e-3)
    del ma, mb, fin
    assert not dev.read_map(capi.MAP_FLAGS).any()


def test_config1_chain_text_file_counted_grid_cluster(oracle, hip, tmp_path):
    """configs[0]: OS1-16 scan written as text -> load_cloud -> points promoted to XYZI (intensity 1) -> VoxelGridCounted
    (leaf 0.5, threshold 0.5) -> cluster (tol 1.5); HIP chain against the oracle chain"""
    scene = synth.make_scene(0)
    s = synth.make_scan(scene, synth.make_pose(0), "os1-16", seed=0)
    f = tmp_path / "scan.xyz"
    np.savetxt(f, np.stack([s.x, s.y, s.z], axis=1), fmt="%.6f")
    _, dev = make_pair(oracle, hip)
    out = []
    for lib, h in ((oracle, None), (hip, dev.h)):
        p = load_cloud(lib, str(f))
        assert p.shape == (16 * 1024, 3)
        pts, keys, grid, _ = voxel_grid_counted(lib, p[:, 0], p[:, 1], p[:, 2], np.ones(len(p), np.float32), 0.5, 0.5, handle=h)
        labels, nc = cluster(lib, pts, keys, grid, 1.5, handle=h)
        out.append((p, pts, keys, labels, nc))
    a, b = out
    np.testing.assert_array_equal(b[0].view(np.uint32), a[0].view(np.uint32))
    np.testing.assert_array_equal(b[1].view(np.uint32), a[1].view(np.uint32))
    np.testing.assert_array_equal(b[2], a[2])
    np.testing.assert_array_equal(b[3], a[3])
    assert a[4] == b[4] and a[4] >= 1 and len(a[1]) > 500


def test_non_finite_and_on_face_points(oracle, hip):
    """NaN / +-Inf coordinates are dropped in the first crop; points exactly on the inclusive faces of the exclude box
    (sensor frame) are removed, points exactly on the faces of the operation area (world frame) are kept
    (vofod_nodelet.cpp:625-655, pcl::CropBox inclusive bounds)"""
    ref, dev = make_pair(oracle, hip, "os1-16", 0.5, max_batch=4)
    for d in (ref, dev):
        synth.seed_ground(d)
    scene = synth.make_scene(7, n_targets=1)
    s = synth.make_scan(scene, synth.make_pose(1), "os1-16", seed=3)
    sp = dev.sp
    x, y, z = s.x.copy(), s.y.copy(), s.z.copy()
    n = x.size
    rng = np.random.default_rng(0)
    bad = rng.choice(n, 600, replace=False)
    x[bad[:100]] = np.nan
    y[bad[100:200]] = np.inf
    z[bad[200:300]] = -np.inf
    x[bad[300:350]], y[bad[300:350]], z[bad[300:350]] = np.nan, np.nan, np.nan
    f32 = np.float32
    tf_id = f32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 5.0]])  # pure translation: the world-frame faces stay exact
    # operation area, world frame (vofod_nodelet.cpp:645-648; z offset is the bottom, :212)
    oc = [f32(sp.oparea_offset[0]), f32(sp.oparea_offset[1]), f32(sp.oparea_offset[2]) + f32(sp.oparea_size[2]) / f32(2)]
    op_lo = [f32(oc[a] - f32(sp.oparea_size[a]) / f32(2)) for a in range(3)]
    op_hi = [f32(oc[a] + f32(sp.oparea_size[a]) / f32(2)) for a in range(3)]
    faces = [(op_lo[0], 0.0, 1.0), (op_hi[0], 0.0, 1.0), (3.0, op_lo[1], 1.0), (3.0, op_hi[1], 1.0), (3.0, 4.0, op_lo[2]), (3.0, 4.0, op_hi[2])]
    # exclude box, sensor frame (:626-629; z offset is the bottom, :204): a point on each max face, inclusive -> removed
    ec = [f32(sp.exclude_offset[0]), f32(sp.exclude_offset[1]), f32(sp.exclude_offset[2]) + f32(sp.exclude_size[2]) / f32(2)]
    for a in range(3):
        i = bad[360 + a]
        q = [ec[0], ec[1], ec[2]]
        q[a] = f32(ec[a] + f32(sp.exclude_size[a]) / f32(2))
        x[i], y[i], z[i] = q
        j = bad[370 + a]  # one ulp outside the exclude box: kept (if inside the operation area)
        q[a] = np.nextafter(q[a], f32(1e9))
        x[j], y[j], z[j] = q
    for k, (fx, fy, fz) in enumerate(faces):
        i = bad[400 + k]
        x[i], y[i], z[i] = np.float32(fx), np.float32(fy), np.float32(fz - 5.0)
        j = bad[420 + k]  # one ulp outside
        x[j], y[j], z[j] = x[i], y[i], z[i]
        if k < 2:
            x[j] = np.nextafter(x[i], np.float32(-1e9 if k == 0 else 1e9))
        elif k < 4:
            y[j] = np.nextafter(y[i], np.float32(-1e9 if k == 2 else 1e9))
        else:
            z[j] = np.nextafter(z[i], np.float32(-1e9 if k == 4 else 1e9))
    scan = ScanData(x=x, y=y, z=z, width=s.scan.width, height=s.scan.height, intensity=s.intensity, range=s.range, stride_bytes=4)
    dr, gr = ref.process_scan(scan, tf_id, debug=True)
    dh, gh = dev.process_scan(scan, tf_id, debug=True)
    assert_scan_debug_equal(gr, gh)
    assert_detections_equal(dr, dh)
    np.testing.assert_array_equal(dev.read_map(), ref.read_map())
    # the same frame through the batched (read-only) path
    scans = [scan, s.scan, scan, scan]
    tfs = np.stack([tf_id, s.tf, tf_id, s.tf]).astype(np.float32)
    da, pa, ga = ref.process_batch(scans, tfs, debug=True)
    db, pb, gb = dev.process_batch(scans, tfs, debug=True)
    np.testing.assert_array_equal(pb, pa)
    for u, v in zip(ga, gb):
        assert_scan_debug_equal(u, v)
    assert gr["n_input_after_crop"] > 1000


def test_calls_that_would_race_a_batch_in_flight_are_refused(oracle, hip):
    """between vofod_batch_submit and vofod_batch_collect the synchronous workspace (ticket 0) and the voxel
    map are in use: process_scan / write_map / ... return VOFOD_ERR_BUSY instead of racing the batch; the batch itself
    collects unharmed and the refused calls succeed afterwards"""
    from vofod_amd.detector import VofodError

    ref, dev = make_pair(oracle, hip, "os1-16", 0.25, max_batch=8)
    scene = synth.make_scene(2, n_targets=2)
    for d in (ref, dev):
        synth.seed_ground(d)
    frames = synth.scan_sequence(scene, "os1-16", 8, seed0=20)
    scans, tfs = [s.scan for s in frames], np.stack([s.tf for s in frames])
    want = ref.process_batch(scans, tfs)
    t0 = dev.batch_submit(scans, tfs)  # ticket 0 = the synchronous workspace
    for call in (lambda: dev.process_scan(scans[0], tfs[0]), lambda: dev.process_scan(scans[0], tfs[0], flags=capi.SCAN_NO_MAP_UPDATE),
                 lambda: dev.process_batch(scans, tfs), lambda: dev.write_map(capi.MAP_VOXELS, np.zeros(dev.n_voxels, np.float32)), lambda: dev.reset()):
        with pytest.raises(VofodError) as e:
            call()
        assert e.value.status == capi.ERR_BUSY
    t1 = dev.batch_submit(scans, tfs)  # a second ticket is fine
    g0, g1 = dev.batch_collect(t0), dev.batch_collect(t1)
    for got in (g0, g1):
        np.testing.assert_array_equal(got[1], want[1])
        assert len(got[0]) == len(want[0])
    dr = ref.process_scan(scans[0], tfs[0])
    dh = dev.process_scan(scans[0], tfs[0])
    assert_detections_equal(dr, dh)
    np.testing.assert_array_equal(dev.read_map(), ref.read_map())


def test_cabi_collective_single_rank(hip):
    """vofod_comm_* / vofod_allgather_detections (RCCL behind the C-ABI): a one-rank communicator on the GPU returns the
    rank's own slots - records in frame order, truncated at d_max, true counts kept.  (More ranks need more GPUs: the
    multi-rank layout is covered on the CPU by tests/test_dist_gloo.py through the same packing.)"""
    from vofod_amd import dist as vdist

    comm = vdist.CabiComm(hip, rank=0, world=1, device=0)
    rng = np.random.default_rng(3)
    per = np.uint32([0, 2, 0, 5, 1, 0, 20, 0])
    dets = np.zeros(int(per.sum()), dtype=capi.DETECTION)
    dets["id"] = np.arange(len(dets))
    dets["n_points"] = rng.integers(1, 50, len(dets))
    dets["confidence"] = rng.random(len(dets))
    dets["position"] = rng.normal(size=(len(dets), 3))
    frame_of = np.repeat(np.arange(len(per)), per)
    dets["frame"] = frame_of
    out, cnt = comm.allgather(dets, per, d_max=16)
    assert out.shape == (1, len(per), 16)
    np.testing.assert_array_equal(cnt[0], per)
    start = np.cumsum(per) - per
    for f in range(len(per)):
        m = min(int(per[f]), 16)
        np.testing.assert_array_equal(out[0, f, :m], dets[start[f] : start[f] + m])
        assert not out[0, f, m:]["n_points"].any()
    out2, cnt2 = comm.allgather(np.zeros(0, dtype=capi.DETECTION), np.zeros(4, np.uint32))  # nothing detected anywhere
    assert not cnt2.any() and out2.shape == (1, 4, 16)
    comm.close()


def test_voxels_as_pc_debug_clouds(oracle, hip):
    """row N4: VoxelMap::voxelsAsPC (voxel_map.cpp:157-183) - background cloud (map > new_obstacles) and sure-air cloud
    (!(map > frontiers)) of a warmed map, in the reference's x-outer / z-inner order, bit for bit"""
    ref, dev = make_pair(oracle, hip, "os1-16", 0.5)
    scene = synth.make_scene(6, n_targets=1)
    for d in (ref, dev):
        synth.seed_ground(d)
    for s in synth.scan_sequence(scene, "os1-16", 3, seed0=5):
        ref.process_scan(s.scan, s.tf)
        dev.process_scan(s.scan, s.tf)
    np.testing.assert_array_equal(dev.read_map(), ref.read_map())
    for thr, gt in ((float(dev.dp.voxel_map__thresholds__new_obstacles), True), (float(dev.dp.voxel_map__thresholds__frontiers), False), (1e9, True)):
        a, b = ref.voxels_as_pc(thr, gt), dev.voxels_as_pc(thr, gt)
        assert a.shape == b.shape
        np.testing.assert_array_equal(b.view(np.uint32), a.view(np.uint32))
    bg = dev.voxels_as_pc(float(dev.dp.voxel_map__thresholds__new_obstacles), True)
    assert len(bg) > 100 and (bg[:, 3] > dev.dp.voxel_map__thresholds__new_obstacles).all()
    # x outer, y, z inner: x never decreases
    assert (np.diff(bg[:, 0]) >= 0).all()


@pytest.mark.parametrize("voxel_size", [0.25, 0.1])
def test_single_pass_input_points_on_cell_boundaries(oracle, hip, voxel_size):
    """The frame kernel reads the input once and encodes the survivors in a reference lattice; points within a rounding band of a
    cell boundary are re-encoded with the frame's own offset.  Frames full of points a few ulps around cell boundaries
    (of the reference lattice and of the frame's own), with different bounding boxes per frame, against the oracle."""
    sensor = "os1-16"
    ref, dev = make_pair(oracle, hip, sensor, voxel_size, max_batch=6, ground_points_max_distance=6 * voxel_size)
    for d in (ref, dev):
        synth.seed_ground(d)
    h, w, _, _ = synth.SENSORS[sensor]
    n = h * w
    rng = np.random.default_rng(17)
    scans, tfs = [], []
    vs = np.float32(voxel_size)
    for f in range(6):
        # cell corners of a lattice shifted by a frame-specific whole number of cells, then nudged by -3..3 ulps / tiny offsets
        base = np.float32([-15.0 + 3.1 * f, -20.0 + 1.7 * f, -1.0])
        k = np.stack([rng.integers(0, int(30 / voxel_size), n), rng.integers(0, int(30 / voxel_size), n), rng.integers(0, int(6 / voxel_size), n)], axis=1)
        p = (base + k.astype(np.float32) * vs).astype(np.float32)
        nudge = rng.integers(-3, 4, (n, 3))
        for _ in range(3):
            up = np.nextafter(p, np.float32(1e9))
            dn = np.nextafter(p, np.float32(-1e9))
            p = np.where(nudge > 0, up, np.where(nudge < 0, dn, p))
            nudge = nudge - np.sign(nudge)
        tiny = rng.choice(np.float32([0, 0, 1e-6, -1e-6, 1e-5, -1e-5, 3e-4, -3e-4, 0.01]), (n, 3))
        p = (p + tiny).astype(np.float32)
        free = rng.random(n) < 0.2  # a fifth of the points anywhere
        p[free] = rng.uniform([-18, -25, -1.2], [40, 30, 8], (int(free.sum()), 3)).astype(np.float32)
        t = np.float32([[1, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 0.0]])  # world = sensor: the boundaries stay where they were put
        scans.append(ScanData(x=np.ascontiguousarray(p[:, 0]), y=np.ascontiguousarray(p[:, 1]), z=np.ascontiguousarray(p[:, 2]), width=w, height=h, stride_bytes=4))
        tfs.append(t)
    tfs = np.stack(tfs)
    da, pa, ga = ref.process_batch(scans, tfs, debug=True)
    db, pb, gb = dev.process_batch(scans, tfs, debug=True)
    np.testing.assert_array_equal(pb, pa)
    for x, y in zip(ga, gb):
        assert_scan_debug_equal(x, y)
    assert min(len(x["weighted"]) for x in ga) > 2000
