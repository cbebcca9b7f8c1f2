"""Property tests of the oracle: conservation, permutation invariance, clustering vs brute force."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from vofod_amd import capi, synth
from vofod_amd.detector import cluster, voxel_grid_counted, voxel_grid_weighted


def _cloud(seed, n, span):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-span, span, size=(n, 3))).astype(np.float32)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 3000), leaf=st.sampled_from([0.1, 0.25, 0.5, 1.0, 3.0]), aligned=st.booleans())
def test_weighted_grid_conserves_points_and_is_permutation_invariant(oracle, seed, n, leaf, aligned):
    q = _cloud(seed, n, 20.0)
    align = (-19.75, -29.75, -1.0) if aligned else None
    out, keys, grid, _ = voxel_grid_weighted(oracle, q[:, 0], q[:, 1], q[:, 2], leaf, align)
    assert int(out["range"].sum()) == n
    assert np.all(np.diff(keys.astype(np.int64)) > 0)  # strictly ascending lattice keys
    perm = np.random.default_rng(seed + 1).permutation(n)
    out2, keys2, _, _ = voxel_grid_weighted(oracle, q[perm, 0], q[perm, 1], q[perm, 2], leaf, align)
    np.testing.assert_array_equal(keys2, keys)
    np.testing.assert_array_equal(out2.view(np.uint32), out.view(np.uint32))
    # every input point lies inside its voxel: |p - centre| <= leaf/2 (+ rounding)
    dx = np.array(list(grid.div_b))
    assert int(keys.max()) < int(dx[0]) * int(dx[1]) * int(dx[2])


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 2000), thr=st.sampled_from([-100.0, -0.1, 0.5]))
def test_counted_grid_total_is_the_true_total(oracle, seed, n, thr):
    q = np.floor(_cloud(seed, n, 30.0)).astype(np.float32)
    inten = np.random.default_rng(seed).uniform(-200, 10, size=n).astype(np.float32)
    out, keys, _, _ = voxel_grid_counted(oracle, q[:, 0], q[:, 1], q[:, 2], inten, 3.0, thr)
    assert int(out["range"].sum()) == int((inten > thr).sum())  # the positional ranges still tile [0, n)


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(2, 600), tol=st.sampled_from([0.7, 1.5, 2.0]))
def test_cluster_labels_match_bruteforce_union_find(oracle, seed, n, tol):
    q = _cloud(seed, n, 6.0)
    pts, keys, grid, _ = voxel_grid_weighted(oracle, q[:, 0], q[:, 1], q[:, 2], 0.5)
    labels, nc = cluster(oracle, pts, keys, grid, tol)
    bf = oracle.extra("vofod_oracle_cluster_bruteforce", C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p])
    ref = np.zeros(len(pts), dtype=np.uint32)
    bf(capi.ptr(pts), len(pts), tol, capi.ptr(ref))
    np.testing.assert_array_equal(labels, ref)
    assert nc == len(np.unique(ref))
    # labels are canonical: the smallest member of each component
    for l in np.unique(labels):
        assert np.flatnonzero(labels == l)[0] == l


def test_sensor_helpers_of_the_product_match_the_oracle(oracle):
    """vofod_ouster_lut / vofod_mask_layout are host code on both sides (no GPU involved): identical outputs"""
    import vofod_amd
    from vofod_amd.detector import mask_layout, ouster_lut

    try:
        hip = vofod_amd.library()
    except (ImportError, OSError) as e:  # libamdhip64 missing on a CPU-only box
        pytest.skip(f"product library not loadable here: {e}")
    rng = np.random.default_rng(11)
    w, h = 64, 16
    az = rng.uniform(-3, 3, h)
    alt = np.linspace(16.6, -16.6, h)
    tf = np.eye(4)
    tf[:3, :3] = [[-1, 0, 0], [0, -1, 0], [0, 0, 1]]
    tf[:3, 3] = [0.0, 0.0, 36.18]
    for kw in ({}, {"tf": tf, "origin_mm": 15.806}):
        a = ouster_lut(oracle, w, h, az, alt, **kw)
        b = ouster_lut(hip, w, h, az, alt, **kw)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    sh = rng.integers(0, 9, h).astype(np.int32)
    np.testing.assert_array_equal(mask_layout(oracle, img, w, h, sh), mask_layout(hip, img, w, h, sh))
    # check_sensor_params: independent implementations (oracle: the reference's statements; product: its own), same verdicts
    from vofod_amd.detector import ScanData, check_sensor_params, sim_lut

    lut = sim_lut(oracle, w, h, 0.58).reshape(h * w, 3)
    for trial in range(40):
        r = rng.integers(0, 60000, h * w).astype(np.uint32)
        r[rng.random(h * w) < 0.3] = 0
        scale = r[:, None].astype(np.float32) * np.float32(0.001)
        p = (lut * scale + rng.normal(0, [0.0, 2e-4, 6e-4, 2e-3][trial % 4], (h * w, 3))).astype(np.float32)
        m = (rng.random(h * w) < 0.8).astype(np.uint8)
        sc = ScanData(x=np.ascontiguousarray(p[:, 0]), y=np.ascontiguousarray(p[:, 1]), z=np.ascontiguousarray(p[:, 2]), width=w, height=h,
                      intensity=np.zeros(h * w, np.float32), range=r, stride_bytes=4)
        assert check_sensor_params(oracle, sc, lut, mask=m) == check_sensor_params(hip, sc, lut, mask=m)


def test_message_serialisation_byte_layout():
    """row N3: vofod/Detections, vofod/Status, vofod/ProfilingInfo in the ROS 1 wire format - expected bytes built by hand
    from msgs/*.msg (little endian, uint32 length prefixes; Header = seq, stamp.sec, stamp.nsec, frame_id)"""
    import ctypes as C
    import struct

    import vofod_amd
    from vofod_amd import capi

    try:
        hip = vofod_amd.library()
    except (ImportError, OSError) as e:
        pytest.skip(f"product library not loadable here: {e}")
    hdr = capi.MsgHeader(7, 1700000000, 250000000, b"uav1/world_origin")
    hdr_bytes = struct.pack("<III", 7, 1700000000, 250000000) + struct.pack("<I", 17) + b"uav1/world_origin"
    dets = np.zeros(2, dtype=capi.DETECTION)
    dets["id"] = [41, 42]
    dets["n_points"] = [9, 3]
    dets["confidence"] = [0.75, 0.5]
    dets["detection_probability"] = [0.9, 0.1]
    dets["position"] = [[1.0, 2.0, 3.0], [-4.0, 5.5, 6.25]]
    dets["covariance"][:, 0] = dets["covariance"][:, 4] = dets["covariance"][:, 8] = [0.3, 0.6]
    want = hdr_bytes + struct.pack("<I", 2)
    for d in dets:
        want += struct.pack("<I", int(d["id"])) + struct.pack("<d", float(d["confidence"])) + struct.pack("<Q", int(d["n_points"]))
        want += struct.pack("<3d", *d["position"]) + struct.pack("<9d", *d["covariance"]) + struct.pack("<d", float(d["detection_probability"]))
    n = C.c_size_t(0)
    assert hip.serialize_detections(C.byref(hdr), capi.ptr(dets), 2, None, 0, C.byref(n)) == capi.ERR_CAPACITY and n.value == len(want)
    buf = np.zeros(n.value, dtype=np.uint8)
    assert hip.serialize_detections(C.byref(hdr), capi.ptr(dets), 2, capi.ptr(buf), buf.size, C.byref(n)) == capi.OK
    assert buf.tobytes() == want
    # Detection.msg: 4 + 8 + 8 + 24 + 72 + 8 bytes per record
    assert len(want) == len(hdr_bytes) + 4 + 2 * 124
    buf = np.zeros(64, dtype=np.uint8)
    assert hip.serialize_status(C.byref(hdr), 1, 0, capi.ptr(buf), buf.size, C.byref(n)) == capi.OK
    assert buf[: n.value].tobytes() == hdr_bytes + bytes([1, 0])
    assert hip.serialize_profiling_info(12, 34, 5, 6, 2, capi.ptr(buf), buf.size, C.byref(n)) == capi.OK
    assert buf[: n.value].tobytes() == struct.pack("<IIIQB", 12, 34, 5, 6, 2) and n.value == 21


@pytest.mark.parametrize("sensor,vs,seed", [("os1-16", 0.5, 3), ("os1-16", 0.25, 4), ("os1-128", 0.5, 5)])
def test_far_clusters_are_the_untainted_components_of_the_far_voxels(oracle, sensor, vs, seed):
    """The claim the close-first kernels rest on (kernels_frame.h, kernels_far.h), checked on the ORACLE's own
    output with scipy: the far clusters of findCloseFarClusters (vofod_nodelet.cpp:727-748: a cluster is close as soon as ANY
    member has a background voxel within hasCloseTo's stencil) are exactly the connected components of the FAR voxels (own
    hasCloseTo false) that have no edge d^2 < tol^2 to a close voxel - same member sets, hence same sizes, smallest members and
    order.  Nothing of the HIP library takes part."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree

    from vofod_amd import synth
    from vofod_amd.detector import VoFOD, default_params

    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(oracle, sp, dp)
    scene = synth.make_scene(seed, n_targets=6)
    warm = synth.make_scene(seed, n_targets=0)  # same buildings, no targets: the targets stay far from the background
    synth.warm_map(det, warm, sensor, 6, seed0=700)
    s = synth.scan_sequence(scene, sensor, 1, seed0=800)[0]
    _, dbg = det.process_scan(s.scan, s.tf, flags=capi.SCAN_NO_MAP_UPDATE, debug=True)
    pts = np.stack([dbg["weighted"]["x"], dbg["weighted"]["y"], dbg["weighted"]["z"]], axis=1).astype(np.float32)
    V = len(pts)
    labels, cl = dbg["labels"], dbg["clusters"]
    assert V > 200 and (cl["is_close"] == 0).any() and (cl["is_close"] == 1).any()
    tol = np.float32(dp.ground_points_max_distance)
    thr = np.float32(dp.voxel_map__thresholds__new_obstacles)
    f = oracle.extra("vofod_oracle_map_has_close_to", C.c_int, [C.c_void_p] + [C.c_float] * 5)
    close = np.array([bool(f(det.h, float(p[0]), float(p[1]), float(p[2]), float(tol), float(thr))) for p in pts])
    # edges: FLANN's L2_Simple in float32 (diff * diff accumulated in x, y, z order), strict < tol^2
    pairs = cKDTree(pts.astype(np.float64)).query_pairs(float(tol) * 1.001, output_type="ndarray")
    d = pts[pairs[:, 0]] - pts[pairs[:, 1]]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]
    pairs = pairs[d2 < tol * tol]
    a, b = pairs[:, 0], pairs[:, 1]
    # the oracle's clustering is the component structure of this graph (clusterCloud :932)
    n_all, comp_all = connected_components(coo_matrix((np.ones(len(a)), (a, b)), shape=(V, V)), directed=False)
    first = np.full(n_all, V, dtype=np.int64)
    np.minimum.at(first, comp_all, np.arange(V))
    np.testing.assert_array_equal(first[comp_all], labels)
    # close first: taint from edges between a far and a close voxel, components of the far-far edges
    far = ~close
    taint = np.zeros(V, dtype=bool)
    mixed = far[a] != far[b]
    taint[np.where(far[a], a, b)[mixed]] = True
    ff = far[a] & far[b]
    _, comp = connected_components(coo_matrix((np.ones(int(ff.sum())), (a[ff], b[ff])), shape=(V, V)), directed=False)
    comp_tainted = np.zeros(comp.max() + 1, dtype=bool)
    np.logical_or.at(comp_tainted, comp[far], taint[far])
    survivors = far & ~comp_tainted[comp]
    far_labels = cl["first_member"][cl["is_close"] == 0]
    np.testing.assert_array_equal(survivors, np.isin(labels, far_labels))
    # ... and component by component: the same member sets (hence sizes, smallest members, order)
    for lab in far_labels:
        members = np.flatnonzero(labels == lab)
        assert len(set(comp[members])) == 1 and (comp == comp[members[0]]).sum() == len(members)
    det.close()


@settings(max_examples=80, deadline=None)
@given(seed=st.integers(0, 100_000), p_unknown=st.sampled_from([0.15, 0.25, 0.3, 0.4, 0.6]), p_ground=st.sampled_from([0.0, 0.01, 0.05]), R=st.integers(2, 14))
def test_explore_to_ground_against_scipy_labelling(oracle, seed, p_unknown, p_ground, R):
    """exploreToGround (voxel_map.cpp:402-488) on random maps against scipy.ndimage: the walk spreads through UNKNOWN voxels
    (unknown_thr < v <= ground_thr, 6-neighbourhood, Manhattan distance <= R from the start); it is "connected" iff a voxel it
    pops lies above the ground threshold or an unknown voxel at distance exactly R - 1 is popped; otherwise it returns the
    unknown voxels it popped (some twice: SURVEY Q7) - as a set, the start's component.  The depth-first order does not matter
    for any of this, which is what the wave-parallel fill of kernels_classify.h relies on."""
    from scipy import ndimage

    from vofod_amd.detector import VoFOD, default_params

    sp, dp = default_params(oracle)
    sp.voxel_size = 0.5
    sp.oparea_offset[:] = (5.0, 5.0, 0.0)
    sp.oparea_size[:] = (10.0, 10.0, 10.0)
    sp.sensor_hrays, sp.sensor_vrays = 8, 2
    det = VoFOD(oracle, sp, dp)
    sx, sy, sz = det.map_size
    rng = np.random.default_rng(seed)
    u = rng.random((sz, sy, sx))
    m = np.full((sz, sy, sx), -1000.0, dtype=np.float32)  # air
    m[u < p_unknown] = -740.0                              # unknown
    m[u > 1.0 - p_ground] = 0.0                            # ground / obstacles
    o = tuple(int(v) for v in rng.integers(3, [sx - 3, sy - 3, sz - 3]))
    if rng.random() < 0.8:
        m[o[2], o[1], o[0]] = -740.0
    det.write_map(capi.MAP_VOXELS, m)
    f = oracle.extra("vofod_oracle_map_explore_to_ground", C.c_int, [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)])
    buf = np.zeros((8192, 3), dtype=np.int32)
    n = C.c_size_t(0)
    off = det.map_offset
    c = [off[a] + (o[a] + 0.5) * 0.5 for a in range(3)]
    conn = bool(f(det.h, c[0], c[1], c[2], -750.0, -300.0, float(R), capi.ptr(buf), 8192, C.byref(n)))
    got = {tuple(r) for r in buf[: n.value].tolist()}

    zz, yy, xx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    manh = np.abs(xx - o[0]) + np.abs(yy - o[1]) + np.abs(zz - o[2])
    ball = manh <= R
    unknown = (m > -750.0) & (m <= -300.0)
    ground = m > -300.0
    start_unknown = bool(unknown[o[2], o[1], o[0]])
    if ground[o[2], o[1], o[0]]:
        want_conn, want = True, set()
    elif not start_unknown:
        want_conn, want = False, set()  # popped, neither ground nor unknown: nothing spreads
    else:
        lab, _ = ndimage.label(unknown & ball)  # 6-neighbourhood by default
        comp = lab == lab[o[2], o[1], o[0]]
        popped = ndimage.binary_dilation(comp) & ball  # the component and everything it pushes
        want_conn = bool((popped & ground).any() or (comp & (manh == R - 1)).any())
        want = set() if want_conn else {(int(x), int(y), int(z)) for z, y, x in zip(*np.nonzero(comp))}
    assert conn == want_conn
    assert got == want
    det.close()


@settings(max_examples=150, deadline=None)
@given(seed=st.integers(0, 1_000_000), length=st.floats(0.05, 9.0))
def test_ray_walk_is_the_geometric_intersection_of_the_segment_with_the_lattice(oracle, seed, length):
    """forEachRay (voxel_map.cpp:229-263, Amanatides-Woo) for random rays inside the default map, against plain geometry: the
    pieces tile the segment (their lengths are >= 0 and sum to the ray's length), consecutive voxels differ by one step along one
    axis in the ray's direction, and the middle of every piece of positive length lies inside the voxel it was charged to."""
    from vofod_amd.detector import VoFOD

    big = VoFOD(oracle)  # 241 x 201 x 51 voxels of 0.5 m, offset (-20, -30, -1.25)
    f = oracle.extra("vofod_oracle_map_ray", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)])
    rng = np.random.default_rng(seed)
    off = np.float64(big.map_offset)
    start = (off + np.float64([40.0, 40.0, 12.0]) + rng.uniform(-3, 3, 3)).astype(np.float32)  # >= 9 m from every face
    d = rng.normal(size=3)
    if rng.random() < 0.2:
        d[rng.integers(0, 3)] = 0.0  # axis-parallel planes: an infinite tmax on that axis
    if not np.any(d):
        d[0] = 1.0
    d = (d / np.linalg.norm(d)).astype(np.float32)
    vox = np.zeros((256, 3), dtype=np.int32)
    dd = np.zeros(256, dtype=np.float32)
    n = C.c_size_t(0)
    f(big.h, capi.ptr(start), capi.ptr(d), np.float32(length), capi.ptr(vox), capi.ptr(dd), 256, C.byref(n))
    k = n.value
    assert 0 < k < 256
    v, w = vox[:k].astype(np.int64), dd[:k].astype(np.float64)
    assert (w >= -1e-6).all()
    assert abs(w.sum() - np.float32(length)) < 1e-4
    first = np.floor((start.astype(np.float64) - off) / 0.5).astype(np.int64)
    np.testing.assert_array_equal(v[0], first)
    steps = np.diff(v, axis=0)
    assert (np.abs(steps).sum(axis=1) == 1).all()            # one axis at a time
    assert (steps * np.sign(d.astype(np.float64)) >= 0).all()  # never against the ray
    t1 = np.cumsum(w)
    mid = start.astype(np.float64) + d.astype(np.float64) * (t1 - 0.5 * w)[:, None]
    cell = (mid - off) / 0.5
    inside = (cell >= v - 1e-4) & (cell <= v + 1 + 1e-4)
    assert inside[w > 1e-4].all()
    big.close()


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 100_000), n=st.integers(1, 5000), leaf=st.sampled_from([0.1, 0.25, 0.5, 1.0]), aligned=st.booleans())
def test_weighted_grid_against_numpy_unique(oracle, seed, n, leaf, aligned):
    """VoxelGridWeighted's sort + run-length pass against numpy: with the lattice the call reports (offset, div_b), the cell of a
    point is floor((p - offset) * inv_leaf) in float32 (voxel_grid_weighted.cpp:131-133), the key i + j*dx + k*dx*dy (:136);
    np.unique of those keys gives the occupied voxels in the output's order, their multiplicities the weights, and the output
    positions are the voxel centres (i + 0.5) * leaf + offset (:175-177), every operation rounded to float32."""
    q = _cloud(seed, n, 25.0)
    if seed % 3 == 0:
        q = np.round(q * 4) / 4  # points on cell boundaries
        q = q.astype(np.float32)
    align = (-19.75, -29.75, -1.0) if aligned else None
    out, keys, grid, _ = voxel_grid_weighted(oracle, q[:, 0], q[:, 1], q[:, 2], leaf, align)
    off = np.float32(list(grid.offset))
    div = np.int64(list(grid.div_b))
    inv = np.float32(1.0) / np.float32(leaf)
    cell = np.floor(((q - off).astype(np.float32) * inv).astype(np.float32)).astype(np.int64)
    assert (cell >= 0).all() and (cell < div).all()
    k = cell[:, 0] + cell[:, 1] * div[0] + cell[:, 2] * div[0] * div[1]
    uk, first, counts = np.unique(k, return_index=True, return_counts=True)
    np.testing.assert_array_equal(keys.astype(np.int64), uk)
    np.testing.assert_array_equal(out["range"].astype(np.int64), counts)
    c = cell[first].astype(np.float32)
    lf = np.float32(leaf)
    centre = (((c + np.float32(0.5)).astype(np.float32) * lf).astype(np.float32) + off).astype(np.float32)
    got = np.stack([out["x"], out["y"], out["z"]], axis=1)
    np.testing.assert_array_equal(got.view(np.uint32), centre.view(np.uint32))


@settings(max_examples=12, deadline=None)
@given(seed=st.integers(0, 100_000), p_bg=st.sampled_from([0.002, 0.01, 0.05]), max_dist=st.sampled_from([0.5, 1.0, 1.5, 1.7]))
def test_has_close_to_is_a_dilation_of_the_occupancy_image(oracle, seed, p_bg, max_dist):
    """hasCloseTo (voxel_map.cpp:376-400) at EVERY cell of a small map against scipy's binary dilation of the thresholded map by
    the stencil {o in [-d, d)^3 : floor(sqrt(|o|^2)) <= max_dist / vs} (half-open cube, truncated integer norm: SURVEY Q3 / Q4),
    cut at the map's faces - i.e. the answer depends on the cell only and is one bit of a dilated image, which is what k_dilate
    builds once per map state and the frame kernel reads (`mapclose`)."""
    from scipy import ndimage

    from vofod_amd.detector import VoFOD, default_params

    sp, dp = default_params(oracle)
    sp.voxel_size = 0.5
    sp.oparea_offset[:] = (3.0, 3.0, 0.0)
    sp.oparea_size[:] = (6.0, 6.0, 6.0)
    sp.sensor_hrays, sp.sensor_vrays = 8, 2
    det = VoFOD(oracle, sp, dp)
    sx, sy, sz = det.map_size
    rng = np.random.default_rng(seed)
    m = np.full((sz, sy, sx), -1000.0, dtype=np.float32)
    m[rng.random((sz, sy, sx)) < p_bg] = 0.0
    det.write_map(capi.MAP_VOXELS, m)
    thr = -300.0
    f = oracle.extra("vofod_oracle_map_has_close_to", C.c_int, [C.c_void_p] + [C.c_float] * 5)
    off = det.map_offset
    got = np.zeros((sz, sy, sx), dtype=bool)
    for k in range(sz):
        for j in range(sy):
            for i in range(sx):
                got[k, j, i] = bool(f(det.h, off[0] + (i + 0.5) * 0.5, off[1] + (j + 0.5) * 0.5, off[2] + (k + 0.5) * 0.5, max_dist, thr))
    mdi = np.float32(max_dist) * np.float32(2.0)  # max_dist * vs_inv
    d = int(np.ceil(mdi))
    # structuring element over offsets -d .. d (scipy wants odd sizes): the +d plane stays empty (half-open cube)
    o = np.arange(-d, d + 1)
    oz, oy, ox = np.meshgrid(o, o, o, indexing="ij")
    S = (np.floor(np.sqrt((ox * ox + oy * oy + oz * oz).astype(np.float64))).astype(np.float32) <= mdi) & (ox < d) & (oy < d) & (oz < d)
    # close(c) = OR over o in S of occ(c + o): a dilation by the REFLECTED stencil (scipy's convention is occ(c - o))
    want = ndimage.binary_dilation(m > thr, structure=S[::-1, ::-1, ::-1])
    np.testing.assert_array_equal(got, want)
    det.close()


def _geometry_raycast(mg_off, vs, sizes, origin, R, dirs, offs, mask, intensity, rng_mm, max_dist, min_intensity):
    """raycast_cloud (vofod_nodelet.cpp:1441-1492) as plain geometry in float64, no DDA: for every cast ray the parameters at
    which the segment [start, start + dir * dist] crosses the voxel planes of each axis, sorted; the piece between two
    consecutive crossings lies in ONE voxel (the voxel of its middle) and adds its length to it."""
    sx, sy, sz = sizes
    acc = np.zeros(sx * sy * sz, dtype=np.float64)
    cast = ~((intensity < min_intensity) | ((mask == 0) & (rng_mm == 0)))  # :1449
    ray_dist = rng_mm.astype(np.float64) * 0.001
    dist = np.where(rng_mm == 0, max_dist, np.minimum(ray_dist - vs, max_dist))  # :1455-1457
    d = dirs.astype(np.float64) @ R.T
    st = offs.astype(np.float64) @ R.T + origin
    inlim = np.all((np.floor((st - mg_off) / vs) >= 0) & (np.floor((st - mg_off) / vs) < np.array(sizes)), axis=1)  # :1482
    upper = mg_off + vs * np.array(sizes)
    for i in np.nonzero(cast & inlim & (dist > 0))[0]:
        s, v, L = st[i], d[i], dist[i]
        # the walk stops where the ray would leave the map (voxel_map.cpp:246-256: cur == last): clip the segment to the map's box
        with np.errstate(divide="ignore", invalid="ignore"):
            t_out = np.where(v > 0, (upper - s) / v, np.where(v < 0, (mg_off - s) / v, np.inf))
        L = min(L, float(t_out.min()))
        ts = [np.array([0.0, L])]
        for a in range(3):
            if v[a] == 0.0:
                continue
            lo, hi = sorted((s[a], s[a] + v[a] * L))
            k = np.arange(np.ceil((lo - mg_off[a]) / vs), np.floor((hi - mg_off[a]) / vs) + 1)
            t = (mg_off[a] + k * vs - s[a]) / v[a]
            ts.append(t[(t > 0) & (t < L)])
        t = np.unique(np.concatenate(ts))
        mid = s[None, :] + v[None, :] * (0.5 * (t[1:] + t[:-1]))[:, None]
        c = np.floor((mid - mg_off) / vs).astype(np.int64)
        ok = np.all((c >= 0) & (c < np.array(sizes)), axis=1)
        np.add.at(acc, ((c[ok, 2] * sy + c[ok, 1]) * sx + c[ok, 0]), np.diff(t)[ok])
    return acc, int((cast & inlim).sum())


def test_whole_scan_raycast_map_is_segment_voxel_geometry(oracle):
    """Row a10 end to end on the oracle, against geometry written without looking at the DDA: the LUT's directions AND beam offsets
    rotated by the pose (`start = R lut.off + t`, :1477), the intensity gate and the `!mask && range == 0` rule (:1446-1449), the
    range clamp `min(range * 0.001 - vs, max_dist)` and the no-return rays cast to max_dist (:1455-1457), the in-limits test of the
    start point (:1482), the clipping at the map's border, and the accumulation over a whole OS1-16 scan.  These are the inputs
    whose host code the product and the oracle share a skeleton for: this pins the oracle's side with a third,
    independent statement.  Tolerance: the oracle accumulates ~1e2 float pieces per voxel (relative 2e-5); the reference's DDA
    keeps `tmax` as a running float sum (voxel_map.cpp:258), so after ~100 steps the boundary between two pieces sits up to ~1e-3 m
    from the geometric plane and that much length moves to the neighbour voxel (absolute 1e-3 m; measured: 4 of 2.47 M voxels
    differ by more than 1e-4 m, the largest by 6.7e-4 m) - the total path length is conserved far more tightly."""
    from vofod_amd.detector import VoFOD, default_params

    sensor, vs = "os1-16", 0.5
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    rng = np.random.default_rng(77)
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    dp.raycast__min_intensity = 250.0
    # a LUT of our own: unit directions of a tilted fan and beam offsets of a few centimetres
    az = np.linspace(0, 2 * np.pi, w, endpoint=False)[None, :] + rng.uniform(-0.05, 0.05, (h, 1))
    alt = np.deg2rad(np.linspace(vfov_deg / 2, -vfov_deg / 2, h))[:, None] + np.zeros((1, w))
    dirs = np.stack([np.cos(alt) * np.cos(az), np.cos(alt) * np.sin(az), np.sin(alt)], axis=-1).reshape(-1, 3).astype(np.float32)
    offs = (0.03 * np.stack([np.cos(az), np.sin(az), 0 * az], axis=-1) + np.array([0.0, 0.0, 0.036]) + 0 * alt[..., None]).reshape(-1, 3).astype(np.float32)
    mask = (rng.random(h * w) < 0.8).astype(np.uint8)
    det = VoFOD(oracle, sp, dp, lut_directions=dirs, lut_offsets=offs, mask=mask)
    scene = synth.make_scene(21, n_targets=2)
    s = synth.scan_sequence(scene, sensor, 1, seed0=300)[0]
    assert det.raycast_begin(s.scan, s.tf) == capi.OK
    got = det.read_map(capi.MAP_RAYCAST).astype(np.float64).reshape(-1)
    tf = np.asarray(s.tf, dtype=np.float64).reshape(3, 4)
    want, n_cast = _geometry_raycast(np.array(det.map_offset, dtype=np.float64), float(vs), tuple(int(x) for x in det.map_size), tf[:, 3], tf[:, :3], dirs, offs, mask,
                                     np.asarray(s.intensity, dtype=np.float64), np.asarray(s.range, dtype=np.int64), float(dp.raycast__max_distance), 250.0)
    assert n_cast > 0.4 * h * w and np.count_nonzero(want) > 20_000
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-3)
    assert int((np.abs(got - want) > 1e-4 + 2e-5 * np.abs(want)).sum()) < 50  # (the drift of tmax shows in a handful of far voxels only)
    np.testing.assert_allclose(got.sum(), want.sum(), rtol=1e-6)
    # the gates were really exercised: dropping either changes the map
    assert int(((np.asarray(s.intensity) < 250.0)).sum()) > 1000 and int(((mask == 0) & (np.asarray(s.range) == 0)).sum()) > 100
    det.close()


@pytest.mark.parametrize("new_rule", [1, 0])
def test_raycast_update_sweep_is_the_voxelwise_formula(oracle, new_rule):
    """The other half of raycast_cloud (vofod_nodelet.cpp:1540-1604) on a whole map, against the formula written out in numpy: a
    voxel changes iff its flag is unmarked and a ray passed; new rule w1 = 2^(-its_diff * coef / (sqrt(3) vs) * r) with the power
    taken in double and rounded to float, old rule w1 = clamp((1 - coef * sqrt(r / max))^its_diff, 0, 1); map <- w1 map + (1 - w1)
    ray_score in float; flags cleared.  Two detection iterations pass between begin and finish (its_diff = 2), the second scan marks
    voxels of its own.  Elementwise, so the tolerance is one float rounding of the product (1e-6 relative): the compiler may or
    may not keep `w1*mapval + w2*score` unfused - the oracle is built with -ffp-contract=off, as the product is."""
    from vofod_amd.detector import VoFOD, default_params

    sensor, vs = "os1-16", 0.5
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    dp.raycast__new_update_rule = new_rule
    det = VoFOD(oracle, sp, dp)
    synth.seed_ground(det)
    scene = synth.make_scene(21, n_targets=2)
    s0, s1, s2 = synth.scan_sequence(scene, sensor, 3, seed0=300)
    det.process_scan(s0.scan, s0.tf)
    its0 = det.status().detection_its
    assert det.raycast_begin(s0.scan, s0.tf) == capi.OK
    det.process_scan(s1.scan, s1.tf)
    det.process_scan(s2.scan, s2.tf)
    its_diff = np.float32(det.status().detection_its - its0)
    assert its_diff == 2
    m, fl, r = (det.read_map(k).astype(np.float32).reshape(-1) for k in (capi.MAP_VOXELS, capi.MAP_FLAGS, capi.MAP_RAYCAST))
    assert det.raycast_finish() == capi.OK
    got = det.read_map(capi.MAP_VOXELS).reshape(-1)
    upd = (fl == 0) & (r > 0)
    # both branches of the condition occur (rays end one voxel in front of their hit, :1457: only a few marked voxels see a ray)
    assert upd.sum() > 10_000 and (fl != 0).sum() > 100 and ((fl != 0) & (r > 0)).sum() >= 1
    score, coef = np.float32(dp.voxel_map__scores__ray), np.float32(dp.raycast__weight_coefficient)
    if new_rule:
        wf = np.float32(coef / np.float32(np.float32(np.sqrt(3.0)) * np.float32(vs)))  # std::sqrt(3) is a double, times a float: rounded once
        n_int = wf * r[upd]
        w1 = np.exp2(-(its_diff.astype(np.float64) * n_int.astype(np.float64))).astype(np.float32)
    else:
        mx = r.max()
        ws = coef * np.sqrt(r[upd] / mx, dtype=np.float32)
        w1 = np.clip(np.power((np.float32(1.0) - ws).astype(np.float64), np.float64(its_diff)).astype(np.float32), np.float32(0), np.float32(1))
    want = m.copy()
    with np.errstate(invalid="ignore"):
        want[upd] = w1 * m[upd] + (np.float32(1.0) - w1) * score
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), fin)
    np.testing.assert_array_equal(got[~upd], m[~upd])  # untouched elsewhere, bit for bit
    np.testing.assert_allclose(got[fin], want[fin], rtol=2e-6, atol=1e-6)
    assert not det.read_map(capi.MAP_FLAGS).any()  # m_voxel_flags.clear() :1601
    det.close()


@pytest.mark.parametrize("sensor,vs", [("os1-16", 0.5), ("os1-128", 0.25)])
def test_map_update_of_a_scan_is_the_voxelwise_formula(oracle, sensor, vs):
    """updateVMaps / updateVoxel (vofod_nodelet.cpp:777-815, called at :946-948) for a whole scan, from the scan's own debug output:
    every voxel of the weighted cloud lands in map cell floor((p - offset) / vs) (float), a voxel of a close cluster pulls the cell
    towards scores/point and flags it 2, a voxel of a far cluster towards scores/unknown and flags it 3, with w = 2^-min(weight, 63)
    (`1lu << clamp(range, 0, 63)`, :789); nothing else of the map changes.  No latch is set on this map, so classification writes no
    frontiers (:1694) and the update is all that happens.  Bit-exact: one product and one sum per cell, no contraction."""
    from vofod_amd.detector import VoFOD, default_params

    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(oracle, sp, dp)
    synth.seed_ground(det)
    scene = synth.make_scene(21, n_targets=2)
    s = synth.scan_sequence(scene, sensor, 1, seed0=300)[0]
    m0, f0 = det.read_map(capi.MAP_VOXELS).reshape(-1).copy(), det.read_map(capi.MAP_FLAGS).reshape(-1).copy()
    dets, g = det.process_scan(s.scan, s.tf, debug=True)
    st = det.status()
    assert not (st.background_pts_sufficient and st.sure_background_sufficient) and len(dets) == 0
    pts, lab, cl = g["weighted"], g["labels"], g["clusters"]
    close_of_root = dict(zip(cl["first_member"].tolist(), cl["is_close"].tolist()))
    is_close = np.array([close_of_root[int(r)] for r in lab], dtype=bool)
    assert is_close.any() and (~is_close).any()
    off = np.array(det.map_offset, dtype=np.float32)
    sx, sy, sz = (int(x) for x in det.map_size)
    inv = np.float32(1.0) / np.float32(vs)
    c = np.stack([np.floor((pts[k] - off[a]) * inv) for a, k in enumerate("xyz")], axis=1).astype(np.int64)
    assert ((c >= 0) & (c < np.array([sx, sy, sz]))).all()
    li = (c[:, 2] * sy + c[:, 1]) * sx + c[:, 0]
    assert len(np.unique(li)) == len(li)  # the grid is aligned to the map: one voxel per cell
    wgt = (np.float32(1.0) / np.exp2(np.minimum(pts["range"], 63).astype(np.float64))).astype(np.float32)
    score = np.where(is_close, np.float32(dp.voxel_map__scores__point), np.float32(dp.voxel_map__scores__unknown)).astype(np.float32)
    want_m, want_f = m0.copy(), f0.copy()
    want_m[li] = wgt * m0[li] + (np.float32(1.0) - wgt) * score
    want_f[li] = np.where(is_close, np.float32(2.0), np.float32(3.0))  # m_vflags_point / m_vflags_unknown, :2335-2337
    np.testing.assert_array_equal(det.read_map(capi.MAP_VOXELS).reshape(-1).view(np.uint32), want_m.view(np.uint32))
    np.testing.assert_array_equal(det.read_map(capi.MAP_FLAGS).reshape(-1), want_f)
    det.close()


@pytest.mark.parametrize("sensor,vs,seed", [("os1-16", 0.5, 300), ("os1-128", 0.25, 301)])
def test_filter_and_transform_of_a_scan_against_numpy_crops(oracle, sensor, vs, seed):
    """filterAndTransform (vofod_nodelet.cpp:621-668) inside process_scan, stage by stage in numpy float32: CropBox 1 removes the
    points INSIDE the exclude box (sensor frame, closed box: offset +- size / 2, the z offset being the box's bottom, :204), the
    rigid transform in PCL's association c0 x + (c1 y + (c2 z + c3)), CropBox 2 keeps the points inside the closed operation area
    (world frame, :212).  The survivors, handed to the stand-alone weighted grid (itself pinned against np.unique by
    test_weighted_grid_against_numpy_unique) with the map's alignment (:664), must give the scan's weighted cloud bit for bit, and
    their number is `n_input_after_crop`."""
    from vofod_amd.detector import VoFOD, default_params

    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(oracle, sp, dp)
    synth.seed_ground(det)
    scene = synth.make_scene(21, n_targets=2)
    s = synth.scan_sequence(scene, sensor, 1, seed0=seed)[0]
    _, g = det.process_scan(s.scan, s.tf, flags=capi.SCAN_NO_MAP_UPDATE, debug=True)
    f32 = np.float32
    x, y, z = (np.asarray(a, dtype=f32) for a in (s.x, s.y, s.z))

    def box(off, size):
        o, sz = np.array(list(off), dtype=f32), np.array(list(size), dtype=f32)
        c = o.copy()
        c[2] = o[2] + sz[2] / f32(2)  # the yaml's z offset is the bottom of the box (:204, :212)
        return c - sz / f32(2), c + sz / f32(2)

    lo, hi = box(sp.exclude_offset, sp.exclude_size)
    inside_ex = (x >= lo[0]) & (x <= hi[0]) & (y >= lo[1]) & (y <= hi[1]) & (z >= lo[2]) & (z <= hi[2])
    keep1 = ~inside_ex & np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    tf = np.asarray(s.tf, dtype=f32).reshape(3, 4)
    q = [tf[r, 0] * x + (tf[r, 1] * y + (tf[r, 2] * z + tf[r, 3])) for r in range(3)]  # every operation rounded to float
    lo, hi = box(sp.oparea_offset, sp.oparea_size)
    keep2 = keep1 & (q[0] >= lo[0]) & (q[0] <= hi[0]) & (q[1] >= lo[1]) & (q[1] <= hi[1]) & (q[2] >= lo[2]) & (q[2] <= hi[2])
    assert 0.1 * x.size < keep2.sum() < 0.9 * x.size and inside_ex.sum() > 0
    assert int(keep2.sum()) == g["n_input_after_crop"]
    align = np.array(det.map_offset, dtype=f32) + f32(vs) / f32(2)  # idxToCoord(0, 0, 0): the centre of map voxel 0 (:664)
    pts, keys, grid, _ = voxel_grid_weighted(oracle, q[0][keep2], q[1][keep2], q[2][keep2], vs, align)
    np.testing.assert_array_equal(pts.view(np.uint32), g["weighted"].view(np.uint32))
    det.close()


def test_detection_records_against_the_formulas_in_numpy(oracle):
    """extractDetections (vofod_nodelet.cpp:834-879) from a scan's own debug output and the map the call leaves behind: one record
    per cluster classified MAV, in cluster order; n_points; position = the OBB centre; covariance = sqrt(dist) sigma I; the
    uncertainty sum over the map box of the cluster's AABB inflated by two voxels (getSubmapCopy, voxel_map.cpp:550-559), the
    cluster's own cells counted as free air, divided by the number of points, confidence = 1 / exp(u) as a float; the detection
    probability from the sensor's ray resolutions.  Double arithmetic on both sides: 1e-9 relative (the order of the double sum)."""
    from vofod_amd.detector import VoFOD, default_params

    sensor, vs = "os1-128", 0.5
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(oracle, sp, dp)
    scene = synth.make_scene(21, n_targets=3)
    det.load_apriori(synth.apriori_points(scene, vs))
    off = np.array(det.map_offset, dtype=np.float32)
    sx, sy, sz = (int(x) for x in det.map_size)
    ray_score = float(dp.voxel_map__scores__ray)
    n_checked = 0
    for s in synth.scan_sequence(scene, sensor, 6, seed0=300):
        dets, g = det.process_scan(s.scan, s.tf, flags=capi.SCAN_AUTO_RAYCAST, debug=True)
        m = det.read_map(capi.MAP_VOXELS).reshape(sz, sy, sx).astype(np.float64)  # x fastest (voxel_map.cpp:81)
        cl, lab, pts = g["clusters"], g["labels"], g["weighted"]
        mav = cl[cl["cclass"] == capi.CLASS_MAV]
        assert len(dets) == len(mav)
        tpos = np.asarray(s.tf, dtype=np.float32).reshape(3, 4)[:, 3]
        for d, c in zip(dets, mav):  # cluster order
            mem = pts[lab == c["first_member"]]
            assert d["n_points"] == len(mem) == c["n_points"]
            np.testing.assert_array_equal(d["position"].astype(np.float32), c["obb_center"])
            dv = (tpos - c["obb_center"]).astype(np.float32)
            dist = float(np.sqrt(np.float32(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2])))  # Eigen's float norm(), widened
            np.testing.assert_allclose(np.asarray(d["covariance"]).reshape(3, 3), np.sqrt(dist) * float(dp.output__position_sigma) * np.eye(3), rtol=1e-6)
            cell = lambda p: np.floor((np.asarray(p, dtype=np.float32) - off) * np.float32(1.0 / vs)).astype(np.int64)
            mn = np.array([mem[k].min() for k in "xyz"], dtype=np.float32)
            mx = np.array([mem[k].max() for k in "xyz"], dtype=np.float32)
            lo = np.clip(cell(mn) - 2, 0, [sx - 1, sy - 1, sz - 1])
            hi = np.clip(cell(mx) + 2, 0, [sx - 1, sy - 1, sz - 1])
            sub = m[lo[2] : hi[2] + 1, lo[1] : hi[1] + 1, lo[0] : hi[0] + 1].copy()
            mc = np.stack([cell(np.array([p["x"], p["y"], p["z"]])) for p in mem]) - lo
            sub[mc[:, 2], mc[:, 1], mc[:, 0]] = np.float64(np.float32(ray_score))  # the cluster's own voxels count as free air (:853-857)
            u = float((1.0 - sub / ray_score).sum()) / len(mem)
            np.testing.assert_allclose(d["confidence"], np.float32(1.0 / np.exp(u)), rtol=2e-6)
            vres, hres = float(sp.sensor_vfov) / h, 2 * np.pi / w
            pdet = min(np.arctan(1.0 / dist) / (vres * int(dp.classification__min_points)), 1.0) * min(np.arctan(1.0 / dist) / hres, 1.0)
            np.testing.assert_allclose(d["detection_probability"], pdet, rtol=1e-9)
            n_checked += 1
    assert n_checked >= 2
    det.close()


def test_classification_of_a_scan_against_scipy_fills_in_cluster_order(oracle):
    """classify_cluster over all far clusters of a scan (vofod_nodelet.cpp:1648-1730, called in cluster order at :961): the three
    gates on the scan's own boxes, then - both latches set - exploreToGround from every member in turn on the map AS THE EARLIER
    FILLS LEFT IT: a connected member makes the cluster UNKNOWN and ends its walk, an unconnected one turns the unknown voxels it
    explored into frontiers (:1712-1715).  The fill is scipy's component of the unknown voxels inside the Manhattan ball (the
    statement test_explore_to_ground_against_scipy_labelling pins on random maps), the map before classification is the pre-scan map
    under the update formula of test_map_update_of_a_scan_is_the_voxelwise_formula.  Classes of all far clusters and the map the
    call leaves behind - frontier voxels included - must be the oracle's, bit for bit."""
    from scipy import ndimage

    from vofod_amd.detector import VoFOD, default_params

    sensor, vs = "os1-128", 0.5
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(oracle, sp, dp)
    scene = synth.make_scene(21, n_targets=3)
    det.load_apriori(synth.apriori_points(scene, vs))
    scans = synth.scan_sequence(scene, sensor, 8, seed0=300)
    for s in scans[:6]:  # carve the free space around the targets (the raycast role), unchecked
        det.process_scan(s.scan, s.tf, flags=capi.SCAN_AUTO_RAYCAST)
    off = np.array(det.map_offset, dtype=np.float32)
    sx, sy, sz = (int(x) for x in det.map_size)
    thr_f, thr_g = np.float32(dp.voxel_map__thresholds__frontiers), np.float32(dp.voxel_map__thresholds__new_obstacles)
    seen = {capi.CLASS_MAV: 0, capi.CLASS_UNKNOWN: 0, capi.CLASS_INVALID: 0}
    n_frontier = 0
    for s in scans[6:]:
        m = det.read_map(capi.MAP_VOXELS).reshape(sz, sy, sx).copy()
        _, g = det.process_scan(s.scan, s.tf, debug=True)
        st = det.status()
        assert st.background_pts_sufficient and st.sure_background_sufficient
        pts, lab, cl = g["weighted"], g["labels"], g["clusters"]
        cell = np.stack([np.floor((pts[k] - off[a]) * np.float32(1.0 / vs)) for a, k in enumerate("xyz")], axis=1).astype(np.int64)
        close_of_root = dict(zip(cl["first_member"].tolist(), cl["is_close"].tolist()))
        is_close = np.array([close_of_root[int(r)] for r in lab], dtype=bool)
        wgt = (np.float32(1.0) / np.exp2(np.minimum(pts["range"], 63).astype(np.float64))).astype(np.float32)
        score = np.where(is_close, np.float32(dp.voxel_map__scores__point), np.float32(dp.voxel_map__scores__unknown)).astype(np.float32)
        iz, iy, ix = cell[:, 2], cell[:, 1], cell[:, 0]
        m[iz, iy, ix] = wgt * m[iz, iy, ix] + (np.float32(1.0) - wgt) * score  # updateVMaps :946-948
        tpos = np.asarray(s.tf, dtype=np.float32).reshape(3, 4)[:, 3]
        for c in cl[cl["is_close"] == 0]:  # far clusters, in cluster order
            want = capi.CLASS_INVALID
            members = np.flatnonzero(lab == c["first_member"])  # ascending voxel index = the order of cluster_indices
            dv = (tpos - c["obb_center"]).astype(np.float32)
            dist = float(np.sqrt(np.float32(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2])))
            if len(members) >= int(dp.classification__min_points) and not dist > float(dp.classification__max_distance) and not float(c["obb_size"]) > float(dp.classification__max_size):
                R = int((float(c["obb_size"]) + float(dp.classification__max_explore_distance)) / float(np.float32(vs)))
                floating = True
                for v in members:
                    o = cell[v]
                    if (o <= 0).any() or (o >= np.array([sx - 1, sy - 1, sz - 1])).any():  # voxel_map.cpp:408-411
                        floating = False
                        break
                    lo = np.maximum(o - (R + 1), 0)
                    hi = np.minimum(o + (R + 1), [sx - 1, sy - 1, sz - 1])
                    box = m[lo[2] : hi[2] + 1, lo[1] : hi[1] + 1, lo[0] : hi[0] + 1]
                    zz, yy, xx = np.meshgrid(*(np.arange(lo[a], hi[a] + 1) for a in (2, 1, 0)), indexing="ij")
                    manh = np.abs(xx - o[0]) + np.abs(yy - o[1]) + np.abs(zz - o[2])
                    ball, ground = manh <= R, box > thr_g
                    unknown = (box > thr_f) & ~ground
                    so = (o[2] - lo[2], o[1] - lo[1], o[0] - lo[0])
                    if ground[so]:
                        floating = False
                        break
                    if not unknown[so]:
                        continue  # popped, neither ground nor unknown: nothing explored
                    labels, _ = ndimage.label(unknown & ball)
                    comp = labels == labels[so]
                    popped = ndimage.binary_dilation(comp) & ball
                    if (popped & ground).any() or (comp & (manh == R - 1)).any():
                        floating = False
                        break
                    box[comp] = thr_f  # (a view: the frontiers stay for the members and clusters that follow)
                    n_frontier += int(comp.sum())
                want = capi.CLASS_MAV if floating else capi.CLASS_UNKNOWN
            assert int(c["cclass"]) == want, (int(c["first_member"]), int(c["n_points"]), int(c["cclass"]), want)
            seen[want] += 1
        np.testing.assert_array_equal(det.read_map(capi.MAP_VOXELS).reshape(sz, sy, sx).view(np.uint32), m.view(np.uint32))
    assert seen[capi.CLASS_MAV] >= 2 and seen[capi.CLASS_UNKNOWN] >= 1 and n_frontier > 0, (seen, n_frontier)
    det.close()


@pytest.mark.parametrize("sensor,vs", [("os1-16", 0.5), ("os1-128", 0.25)])
def test_close_far_split_and_clusters_of_a_scan_against_scipy(oracle, sensor, vs):
    """clusterCloud + findCloseFarClusters of a whole scan (vofod_nodelet.cpp:689-750) from its weighted cloud alone: the clusters
    are the connected components of the graph `d^2 < tol^2` over the voxel centres (FLANN's float expression; scipy's sparse
    components here), a cluster is close iff ANY member's cell lies in the occupancy image `m > new_obstacles` of the map BEFORE the
    update, dilated by hasCloseTo's stencil (half-open cube, truncated integer norm: the structuring element of
    test_has_close_to_is_a_dilation_of_the_occupancy_image); `n_bg_voxels` = nVoxelsOver of that map.  Labels (smallest member),
    sizes, close flags: bit for bit."""
    from scipy import ndimage, sparse
    from scipy.sparse import csgraph
    from scipy.spatial import cKDTree

    from vofod_amd.detector import VoFOD, default_params

    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(oracle)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(oracle, sp, dp)
    scene = synth.make_scene(21, n_targets=3)
    synth.warm_map(det, scene, sensor, 4)
    s = synth.scan_sequence(scene, sensor, 1, seed0=555)[0]
    sx, sy, sz = (int(x) for x in det.map_size)
    m = det.read_map(capi.MAP_VOXELS).reshape(sz, sy, sx).copy()
    _, g = det.process_scan(s.scan, s.tf, flags=capi.SCAN_NO_MAP_UPDATE, debug=True)
    pts, lab, cl = g["weighted"], g["labels"], g["clusters"]
    P = np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)
    tol = np.float32(dp.ground_points_max_distance)
    # candidate pairs from a k-d tree with a margin, the decision by the float expression itself (strict <)
    pairs = cKDTree(P.astype(np.float64)).query_pairs(float(tol) * 1.001, output_type="ndarray")
    d = P[pairs[:, 0]] - P[pairs[:, 1]]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]  # L2_Simple accumulates in float, axis by axis
    keep = d2 < tol * tol
    n = len(P)
    _, comp = csgraph.connected_components(sparse.coo_matrix((np.ones(int(keep.sum())), (pairs[keep, 0], pairs[keep, 1])), shape=(n, n)), directed=False)
    first = np.full(comp.max() + 1, n, dtype=np.int64)
    np.minimum.at(first, comp, np.arange(n))
    np.testing.assert_array_equal(lab, first[comp].astype(lab.dtype))  # label = smallest member
    # close / far
    thr = np.float32(dp.voxel_map__thresholds__new_obstacles)
    occ = m > thr
    assert g["n_bg_voxels"] == int(occ.sum())
    mdi = np.float32(dp.ground_points_max_distance) * np.float32(np.float32(1.0) / np.float32(vs))
    dd = int(np.ceil(mdi))
    o = np.arange(-dd, dd + 1)
    oz, oy, ox = np.meshgrid(o, o, o, indexing="ij")
    S = (np.floor(np.sqrt((ox * ox + oy * oy + oz * oz).astype(np.float64))).astype(np.float32) <= mdi) & (ox < dd) & (oy < dd) & (oz < dd)
    close_img = ndimage.binary_dilation(occ, structure=S[::-1, ::-1, ::-1])
    off = np.array(det.map_offset, dtype=np.float32)
    c = np.stack([np.floor((pts[k] - off[a]) * np.float32(1.0 / vs)) for a, k in enumerate("xyz")], axis=1).astype(np.int64)
    vox_close = close_img[c[:, 2], c[:, 1], c[:, 0]]
    want_close = np.zeros(comp.max() + 1, dtype=bool)
    np.logical_or.at(want_close, comp, vox_close)
    roots = cl["first_member"].astype(np.int64)
    np.testing.assert_array_equal(cl["is_close"].astype(bool), want_close[comp[roots]])
    np.testing.assert_array_equal(cl["n_points"], np.bincount(comp, minlength=comp.max() + 1)[comp[roots]])
    assert len(cl) == comp.max() + 1 and want_close.any() and (~want_close).any()
    det.close()


@pytest.mark.parametrize("seed,max_bg", [(1, 0.8), (2, 0.8), (3, 1.6), (4, 2.2), (5, 0.8)])
def test_sepclusters_role_against_numpy(oracle, seed, max_bg):
    """updateSeparatedBGClusters (vofod_nodelet.cpp:1126-1277) on random maps, restated in numpy from the cited lines:
    voxelsAsVoxelPC in x-outer / z-inner order with voxel INDICES as coordinates (voxel_map.cpp:186-211); VoxelGridCounted at leaf
    max(ceil(d / vs) - 1, 0) (voxel_grid_counted.cpp:58-213, no alignment) whose `range` counts the sure voxels among the INPUT
    positions [first, last) of the SORTED run (Q1: positions of one order applied to the other); Euclidean clusters at tolerance
    ceil(d / vs), strict; a cluster is sure when its ranges sum to min_sure_points; without any sure cluster the latch goes off and
    nothing changes; otherwise every voxel within the truncated-norm stencil of every down-sampled point (cast<int>) of every unsure
    cluster is pulled half-way to scores/ray, once per (point, offset) pair.  Latch and map bit for bit."""
    from scipy import sparse
    from scipy.sparse import csgraph
    from scipy.spatial import cKDTree

    from vofod_amd.detector import VoFOD, default_params

    f32 = np.float32
    sp, dp = default_params(oracle)
    vs = 0.5
    sp.voxel_size = vs
    sp.oparea_offset[:] = (10.0, 10.0, 0.0)
    sp.oparea_size[:] = (20.0, 20.0, 10.0)
    sp.sensor_hrays, sp.sensor_vrays = 8, 2
    dp.sepclusters__max_bg_distance = max_bg
    det = VoFOD(oracle, sp, dp)
    sx, sy, sz = (int(x) for x in det.map_size)
    rng = np.random.default_rng(seed)
    thr_new, thr_sure = f32(dp.voxel_map__thresholds__new_obstacles), f32(dp.voxel_map__thresholds__sure_obstacles)
    air = f32(-1000.0)
    m = np.full((sz, sy, sx), air, dtype=f32)
    # a ground sheet of sure voxels with a few unsure holes, islands of unsure voxels in the air, one small island of sure ones
    if seed != 5:  # (seed 5: no sheet - no cluster collects min_sure_points sure voxels, the latch goes off, the map stays)
        m[1, 3 : sy - 3, 3 : sx - 3] = f32(5.0)
        holes = rng.random((sy - 6, sx - 6)) < 0.1
        m[1, 3 : sy - 3, 3 : sx - 3][holes] = f32(-150.0)
    for _ in range(int(rng.integers(4, 9))):
        c = rng.integers([6, 6, 8], [sx - 6, sy - 6, sz - 3])
        for _ in range(int(rng.integers(1, 7))):
            q = c + rng.integers(-1, 2, 3)
            m[q[2], q[1], q[0]] = f32(rng.choice([-150.0, -200.0, 5.0], p=[0.5, 0.3, 0.2]))
    assert thr_new < f32(-200.0) and f32(-150.0) <= thr_sure < f32(5.0)
    det.write_map(capi.MAP_VOXELS, m)
    st, sure = det.sepclusters_begin(allow=(capi.ERR_EMPTY,))
    assert st == capi.OK
    if sure:
        assert det.sepclusters_finish() == capi.OK
    got = det.read_map(capi.MAP_VOXELS).reshape(sz, sy, sx)

    # ---- the same in numpy
    mdi = f32(np.float64(max_bg) / np.float64(f32(vs)))
    mvd = int(np.ceil(mdi))
    mx = m.transpose(2, 1, 0)  # [x, y, z]
    occ = mx > thr_new
    P = np.argwhere(occ)  # x outer, z inner
    inten = mx[occ]
    lsz = f32(max(mvd - 1, 0))
    assert lsz > 0
    inv = f32(1.0) / lsz
    Pf = P.astype(f32)
    min_b = np.floor(Pf.min(0) * inv).astype(np.int64)
    max_b = np.floor(Pf.max(0) * inv).astype(np.int64)
    offset = min_b.astype(f32) * lsz
    div = max_b - min_b + 1
    ijk = np.floor((Pf - offset) * inv).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    uniq, first, counts = np.unique(idx[order], return_index=True, return_counts=True)
    ds = ((ijk[order[first]].astype(f32) + f32(0.5)) * lsz + offset).astype(f32)
    ranges = np.array([int((inten[a : a + n] > thr_sure).sum()) for a, n in zip(first, counts)])  # INPUT positions (Q1)
    tol = f32(mvd)
    pairs = cKDTree(ds.astype(np.float64)).query_pairs(float(tol) * 1.001, output_type="ndarray")
    d = ds[pairs[:, 0]] - ds[pairs[:, 1]]
    keep = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2] < tol * tol
    n = len(ds)
    _, comp = csgraph.connected_components(sparse.coo_matrix((np.ones(int(keep.sum())), (pairs[keep, 0], pairs[keep, 1])), shape=(n, n)), directed=False)
    n_sure = np.bincount(comp, weights=ranges, minlength=comp.max() + 1)
    sure_cluster = n_sure >= int(dp.sepclusters__min_sure_points)
    assert sure == bool(sure_cluster.any()) == (seed != 5)
    want = m.copy()
    if sure_cluster.any():
        o = np.arange(-mvd, mvd + 1)
        ox, oy, oz = (a.ravel() for a in np.meshgrid(o, o, o, indexing="ij"))
        sel = np.floor(np.sqrt((ox * ox + oy * oy + oz * oz).astype(np.float64))).astype(f32) <= mdi  # Vector3i::norm() is an int
        offs = np.stack([ox[sel], oy[sel], oz[sel]], axis=1)
        hits = np.zeros((sx, sy, sz), dtype=np.int64)
        for p in ds[~sure_cluster[comp]].astype(np.int64):  # cast<int>: truncation (coordinates are >= 0)
            t = p + offs
            ok = ((t >= 0) & (t < [sx, sy, sz])).all(axis=1)
            np.add.at(hits, (t[ok, 0], t[ok, 1], t[ok, 2]), 1)
        w1 = f32(0.5)  # clamp(pow(1 - 0.5, max(its_diff, 1)), 0, 1): no detection iteration between begin and finish
        u = f32(dp.voxel_map__scores__ray)
        wx = want.transpose(2, 1, 0)
        for k in range(1, int(hits.max()) + 1):
            again = hits >= k
            wx[again] = w1 * wx[again] + (f32(1.0) - w1) * u
        assert (~sure_cluster).any() and hits.max() >= 1
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    det.close()
