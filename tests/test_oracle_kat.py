"""Hand-derived known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors, so every expected value below is derived
by hand from the reference lines cited next to it (paths relative to the reference tree).
"""
import ctypes as C
import math

import numpy as np
import pytest

from vofod_amd import capi
from vofod_amd.detector import ScanData, VoFOD, default_params, load_cloud, mask_layout, ouster_lut, sim_lut, voxel_grid_counted, voxel_grid_weighted, cluster


def _pts(*xyz):
    a = np.asarray(xyz, dtype=np.float32)
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()


EIGHT = [(0.1, 0.1, 0.1), (0.2, 0.3, 0.4), (0.45, 0.05, 0.2), (0.6, 0.1, 0.1), (0.9, 0.4, 0.4), (0.1, 0.1, 0.7), (0.3, 0.2, 0.9), (0.49, 0.49, 0.51)]


def test_weighted_grid_8_points_3_voxels(oracle):
    # voxel_grid_weighted.cpp:72-80: min_p=(0.1,0.05,0.1) -> min_b=0 -> offset=0; max_b=(1,0,1) -> div=(2,1,2)
    # :136 key = i + j*2 + k*2;  :178-181 centre=(ijk+0.5)*0.5, weight=#points
    x, y, z = _pts(*EIGHT)
    out, keys, grid, st = voxel_grid_weighted(oracle, x, y, z, 0.5)
    assert st == capi.OK
    assert list(grid.div_b) == [2, 1, 2] and list(grid.min_b) == [0, 0, 0]
    assert keys.tolist() == [0, 1, 2]
    assert out["range"].tolist() == [3, 2, 3]
    np.testing.assert_array_equal(out["x"], np.float32([0.25, 0.75, 0.25]))
    np.testing.assert_array_equal(out["y"], np.float32([0.25, 0.25, 0.25]))
    np.testing.assert_array_equal(out["z"], np.float32([0.25, 0.25, 0.75]))


def test_weighted_grid_aligned_to_sim_map_lattice(oracle):
    # SURVEY Q2 / voxel_grid_weighted.cpp:81-106 with align_center = idxToCoord(0,0,0) of the sim.yaml map at
    # 0.5 m: map offset = (40-60, 20-50, (-1.25+12.5)-12.5) = (-20,-30,-1.25) -> centre (-19.75,-29.75,-1.0).
    # aco = fmod(c - 0.25, 0.5) = (0, 0, -0.25 -> +0.5 = 0.25): z faces move to -0.25, 0.25, 0.75, 1.25.
    x, y, z = _pts(*EIGHT)
    out, keys, grid, st = voxel_grid_weighted(oracle, x, y, z, 0.5, align_center=(-19.75, -29.75, -1.0))
    assert list(grid.min_b) == [0, 0, -1] and list(grid.div_b) == [2, 1, 3]
    np.testing.assert_array_equal(np.float32(list(grid.offset)), np.float32([0.0, 0.0, -0.25]))
    # cells (i,j,k): A,C->(0,0,0) D->(1,0,0) B,F,H->(0,0,1) E->(1,0,1) G->(0,0,2); key = i + 2k
    assert keys.tolist() == [0, 1, 2, 3, 4]
    assert out["range"].tolist() == [2, 1, 3, 1, 1]
    np.testing.assert_array_equal(out["z"], np.float32([0.0, 0.0, 0.5, 0.5, 1.0]))
    np.testing.assert_array_equal(out["x"], np.float32([0.25, 0.75, 0.25, 0.75, 0.25]))


def test_weighted_grid_empty_and_overflow(oracle):
    e = np.zeros(0, dtype=np.float32)
    out, keys, grid, st = voxel_grid_weighted(oracle, e, e, e, 0.5)
    assert st == capi.OK and out.size == 0
    # voxel_grid_weighted.cpp:61-69: (1e6/0.01+2)^3 > INT32_MAX -> warn + empty output
    x, y, z = _pts((0, 0, 0), (1e6, 1e6, 1e6))
    out, keys, grid, st = voxel_grid_weighted(oracle, x, y, z, 0.01, allow=(capi.ERR_INDEX_OVERFLOW,))
    assert st == capi.ERR_INDEX_OVERFLOW and out.size == 0


def test_counted_grid_positional_range_quirk(oracle):
    # SURVEY Q1 / voxel_grid_counted.cpp:185-187.  2x2x2 index cloud in voxelsAsVoxelPC order
    # (x outer, z inner: voxel_map.cpp:191-195), leaf 1 -> every point its own voxel, key = x + 2y + 4z.
    # Sorted run k is [k,k+1), so voxel key k receives the flag of *input position* k.
    pts = [(xx, yy, zz) for xx in (0, 1) for yy in (0, 1) for zz in (0, 1)]
    x, y, z = _pts(*pts)
    inten = np.float32([0, 5, 0, 0, 0, 0, 7, 0])  # input positions 1 and 6 are "sure" (> 1.0)
    out, keys, grid, st = voxel_grid_counted(oracle, x, y, z, inten, 1.0, 1.0)
    assert keys.tolist() == list(range(8))
    assert out["range"].tolist() == [0, 1, 0, 0, 0, 0, 1, 0]
    # a true per-voxel count would instead mark key 4 (point (0,0,1)) and key 3 (point (1,1,0))
    assert int(out["range"].sum()) == 2
    np.testing.assert_array_equal(out["x"], np.float32([0.5, 1.5] * 4))


def test_counted_grid_runs_cover_positions(oracle):
    # two points per voxel: leaf 2 over the same 2x2x2 cloud -> one voxel, run [0,8) -> all flags counted
    pts = [(xx, yy, zz) for xx in (0, 1) for yy in (0, 1) for zz in (0, 1)]
    x, y, z = _pts(*pts)
    inten = np.float32([0, 5, 0, 0, 0, 0, 7, 0])
    out, keys, grid, st = voxel_grid_counted(oracle, x, y, z, inten, 2.0, 1.0)
    assert keys.tolist() == [0] and out["range"].tolist() == [2]


def test_cluster_strict_tolerance(oracle):
    # [3P] FLANN radius search keeps d^2 < r^2 (strict).  Lattice leaf 0.5, tol 1.5: centres 3 voxels apart
    # (1.75-0.25 = 1.5 exactly, 2.25 < 2.25 false) do not connect; 2 voxels apart do.
    pts = np.zeros(4, dtype=capi.POINT_XYZR)
    pts["x"] = [0.25, 1.75, 10.25, 11.25]
    pts["y"] = 0.25
    pts["z"] = 0.25
    labels, nc = cluster(oracle, pts, None, None, 1.5)
    assert labels.tolist() == [0, 1, 2, 2] and nc == 3


def _toy(oracle, **dyn):
    # 5x5x5 map with offset 0: sizes = ceil(2/0.5)+1 (voxel_map.cpp:16), offset = centre - dims/2 (:15)
    sp, dp = default_params(oracle)
    sp.voxel_size = 0.5
    sp.oparea_offset[:] = (1.0, 1.0, 0.0)
    sp.oparea_size[:] = (2.0, 2.0, 2.0)
    sp.sensor_hrays, sp.sensor_vrays = 8, 2
    for k, v in dyn.items():
        setattr(dp, k, v)
    return VoFOD(oracle, sp, dp)


def test_map_geometry(oracle):
    d = _toy(oracle)
    assert d.map_size == (5, 5, 5) and d.map_offset == (0.0, 0.0, 0.0)
    big = VoFOD(oracle)
    assert big.map_size == (241, 201, 51)  # default sim.yaml map at 0.5 m
    assert big.map_offset == (-20.0, -30.0, -1.25)
    f = oracle.extra("vofod_oracle_map_coord_to_idx", C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p])
    out = np.zeros(3, dtype=np.int32)
    assert f(big.h, 100.0, 70.0, 23.75, capi.ptr(out)) == 1  # the max corner is still inside (inclusive crop + the "+1" size)
    assert out.tolist() == [240, 200, 50]
    assert f(big.h, -20.01, 0.0, 0.0, capi.ptr(out)) == 0 and out[0] == -1


def test_has_close_to_quirks(oracle):
    # voxel_map.cpp:376-400 with vs 0.5, max_dist 1.5 -> max_dist_idx 3, d = 3.
    big = VoFOD(oracle)
    f = oracle.extra("vofod_oracle_map_has_close_to", C.c_int, [C.c_void_p] + [C.c_float] * 5)
    sx, sy, sz = big.map_size
    o = (100, 100, 25)
    cx, cy, cz = (-20 + (o[0] + 0.5) * 0.5, -30 + (o[1] + 0.5) * 0.5, -1.25 + (o[2] + 0.5) * 0.5)

    def probe(delta):
        m = np.full((sz, sy, sx), -740.0, dtype=np.float32)
        m[o[2] + delta[2], o[1] + delta[1], o[0] + delta[0]] = 0.0
        big.write_map(capi.MAP_VOXELS, m)
        return bool(f(big.h, cx, cy, cz, 1.5, -300.0))

    assert probe((-3, 0, 0)) is True     # the -d plane is scanned
    assert probe((3, 0, 0)) is False     # half-open cube, the +d plane never is
    assert probe((2, 2, 1)) is True      # 9 <= 9
    assert probe((-3, -2, -1)) is True   # floor(sqrt(14)) = 3 <= 3 although 3.74 > 3
    assert probe((-3, -2, -2)) is False  # floor(sqrt(17)) = 4
    assert probe((-3, -3, -3)) is False  # floor(sqrt(27)) = 5
    assert probe((0, 0, 0)) is True


def test_explore_to_ground_exits(oracle):
    # voxel_map.cpp:402-488 on a 5^3 map
    d = _toy(oracle)
    f = oracle.extra("vofod_oracle_map_explore_to_ground", C.c_int, [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)])
    buf = np.zeros((256, 3), dtype=np.int32)
    n = C.c_size_t(0)

    def run(m, idx, max_voxel_dist):
        d.write_map(capi.MAP_VOXELS, m)
        c = [(i + 0.5) * 0.5 for i in idx]
        conn = f(d.h, c[0], c[1], c[2], -750.0, -300.0, float(max_voxel_dist), capi.ptr(buf), 256, C.byref(n))
        return bool(conn), [tuple(r) for r in buf[: n.value].tolist()]

    unknown = np.full((5, 5, 5), -740.0, dtype=np.float32)
    air = np.full((5, 5, 5), -1000.0, dtype=np.float32)
    # exit 1 (:408-411): the start voxel touches the map border
    assert run(unknown, (0, 2, 2), 3) == (True, [])
    assert run(unknown, (2, 4, 2), 3) == (True, [])
    # exit 2 (:424): a voxel above the ground threshold is reached through unknown voxels
    m = air.copy()
    m[2, 2, 2] = -740.0
    m[2, 2, 3] = 0.0  # [z,y,x] -> voxel (3,2,2)
    assert run(m, (2, 2, 2), 3) == (True, [])
    # exit 3 (:430): an unknown voxel at Manhattan distance max-1 is popped
    assert run(unknown, (2, 2, 2), 2) == (True, [])
    # not connected: the unknown start voxel is enclosed by air; it is the only explored unknown voxel
    m = air.copy()
    m[2, 2, 2] = -740.0
    assert run(m, (2, 2, 2), 3) == (False, [(2, 2, 2)])
    # an enclosed 2-voxel unknown pocket: DFS pops the most recently pushed neighbour first; the start voxel is
    # pushed again from the neighbour because `explored` is only filled after expansion (:484) -> duplicate entry
    m = air.copy()
    m[2, 2, 2] = -740.0
    m[2, 2, 1] = -740.0  # voxel (1,2,2)
    conn, expl = run(m, (2, 2, 2), 3)
    assert conn is False and expl[0] == (2, 2, 2) and set(expl) == {(2, 2, 2), (1, 2, 2)}


def test_ray_traversal_axis_and_diagonal(oracle):
    # voxel_map.cpp:229-263
    big = VoFOD(oracle)
    f = oracle.extra("vofod_oracle_map_ray", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)])
    vox = np.zeros((64, 3), dtype=np.int32)
    dd = np.zeros(64, dtype=np.float32)
    n = C.c_size_t(0)
    o = (40, 60, 10)
    start = np.float32([-20 + (o[0] + 0.5) * 0.5, -30 + (o[1] + 0.5) * 0.5, -1.25 + (o[2] + 0.5) * 0.5])
    # along +x from a voxel centre, length 2: 0.25 + 0.5*3 + 0.25
    f(big.h, capi.ptr(start), capi.ptr(np.float32([1, 0, 0])), 2.0, capi.ptr(vox), capi.ptr(dd), 64, C.byref(n))
    assert n.value == 5
    assert vox[:5].tolist() == [[40 + k, 60, 10] for k in range(5)]
    np.testing.assert_array_equal(dd[:5], np.float32([0.25, 0.5, 0.5, 0.5, 0.25]))
    # along -z
    f(big.h, capi.ptr(start), capi.ptr(np.float32([0, 0, -1])), 0.6, capi.ptr(vox), capi.ptr(dd), 64, C.byref(n))
    assert vox[: n.value].tolist() == [[40, 60, 10], [40, 60, 9]]
    np.testing.assert_allclose(dd[: n.value], [0.25, 0.35], rtol=0, atol=1e-7)
    # exact diagonal in xy: tmax ties are broken towards the lower axis (Eigen minCoeff keeps the first minimum),
    # so the walk alternates x then y and every other visit has zero length
    s = np.float32(1.0 / math.sqrt(2.0))
    f(big.h, capi.ptr(start), capi.ptr(np.float32([s, s, 0])), 1.5, capi.ptr(vox), capi.ptr(dd), 64, C.byref(n))
    v = vox[: n.value].tolist()
    assert v[0] == [40, 60, 10] and v[1] == [41, 60, 10] and v[2] == [41, 61, 10] and v[3] == [42, 61, 10]
    assert abs(float(dd[: n.value].sum()) - 1.5) < 1e-6
    assert dd[1] == 0.0 and dd[3] == 0.0
    # negative length (point closer than one voxel: vofod_nodelet.cpp:1457) visits nothing
    f(big.h, capi.ptr(start), capi.ptr(np.float32([1, 0, 0])), -0.2, capi.ptr(vox), capi.ptr(dd), 64, C.byref(n))
    assert n.value == 0
    # the walk stops at the map face (:257)
    f(big.h, capi.ptr(start), capi.ptr(np.float32([0, 0, 1])), 1000.0, capi.ptr(vox), capi.ptr(dd), 64, C.byref(n))
    assert vox[n.value - 1].tolist() == [40, 60, 50] and n.value == 41


def _single_return_scan(oracle, d, col, row, range_mm):
    w, h = d.sp.sensor_hrays, d.sp.sensor_vrays
    lut = sim_lut(oracle, w, h, d.sp.sensor_vfov)
    n = w * h
    rng = np.zeros(n, dtype=np.uint32)
    rng[row * w + col] = range_mm
    r = rng.astype(np.float32) * np.float32(0.001)
    x, y, z = (lut[:, 0] * r).astype(np.float32), (lut[:, 1] * r).astype(np.float32), (lut[:, 2] * r).astype(np.float32)
    return ScanData(x=x, y=y, z=z, width=w, height=h, intensity=np.full(n, 100, dtype=np.float32), range=rng)


def test_single_voxel_detection_confidence(oracle):
    # a 1-voxel cluster in an all-init map.  Sub-map = AABB inflated by 2 -> 5^3 = 125 cells
    # (voxel_map.cpp:547-584); cluster voxel := ray score -> 1-1 = 0; the other 124 hold -740 ->
    # 1-0.74 = 0.26 each; u = 124*0.26/1 = 32.24; confidence = exp(-32.24)  (vofod_nodelet.cpp:851-867)
    sp, dp = default_params(oracle)
    dp.classification__min_points = 1
    dp.voxel_map__thresholds__frontiers = -700.0  # init voxels count as air -> the voxel floats
    d = VoFOD(oracle, sp, dp)
    d.load_apriori(np.zeros((0, 3), dtype=np.float32))  # sets both background latches (:343-344)
    tf = np.float32([[1, 0, 0, 40], [0, 1, 0, 20], [0, 0, 1, 5]])
    scan = _single_return_scan(oracle, d, col=0, row=64, range_mm=10000)
    dets, dbg = d.process_scan(scan, tf, debug=True)
    assert dbg["n_input_after_crop"] == 1 and len(dbg["weighted"]) == 1 and len(dbg["clusters"]) == 1
    assert dbg["clusters"][0]["is_close"] == 0 and dbg["clusters"][0]["cclass"] == capi.CLASS_MAV
    assert len(dets) == 1
    det = dets[0]
    assert det["id"] == 0 and det["n_points"] == 1
    assert det["confidence"] == pytest.approx(math.exp(-32.24), rel=1e-5)
    # position = centre of the voxel holding the point
    wpt = dbg["weighted"][0]
    assert (det["position"] == [wpt["x"], wpt["y"], wpt["z"]]).all()
    dist = math.dist(det["position"], (40, 20, 5))
    vres = float(np.float32(d.sp.sensor_vfov)) / 128
    hres = 2 * math.pi / 1024
    exp_p = min(math.atan(1 / dist) / (vres * 1), 1.0) * min(math.atan(1 / dist) / hres, 1.0)
    assert det["detection_probability"] == pytest.approx(exp_p, rel=1e-6)
    assert det["covariance"][0] == pytest.approx(math.sqrt(dist) * 0.1, rel=1e-6) and det["covariance"][1] == 0
    # the far voxel was updated with the unknown score and flagged 3 (vofod_nodelet.cpp:948, 2337)
    flags = d.read_map(capi.MAP_FLAGS)
    assert flags.sum() == 3.0 and np.count_nonzero(flags) == 1
    assert d.status().detection_its == 1 and d.status().last_detection_id == 1


def test_update_voxel_weight(oracle):
    # vofod_nodelet.cpp:791-794: w = 2^-clamp(count,0,63); m = w*m + (1-w)*score.  Seed a background voxel
    # next to the point so its cluster is "close" and is updated with scores/point = 0.
    d = VoFOD(oracle)
    tf = np.float32([[1, 0, 0, 40], [0, 1, 0, 20], [0, 0, 1, 5]])
    scan = _single_return_scan(oracle, d, col=0, row=64, range_mm=10000)
    _, dbg = d.process_scan(scan, tf, debug=True)
    w = dbg["weighted"][0]
    ix, iy, iz = int((w["x"] + 20) * 2), int((w["y"] + 30) * 2), int((w["z"] + 1.25) * 2)
    d.reset()
    m = d.read_map()
    m[iz, iy, ix + 1] = 0.0
    d.write_map(capi.MAP_VOXELS, m)
    _, dbg = d.process_scan(scan, tf, debug=True)
    assert dbg["clusters"][0]["is_close"] == 1 and dbg["n_bg_voxels"] == 1
    m2 = d.read_map()
    assert m2[iz, iy, ix] == np.float32(-370.0)  # count 1 -> w = 0.5
    assert d.read_map(capi.MAP_FLAGS)[iz, iy, ix] == 2.0
    # an apriori (+inf) voxel stays +inf under the update
    m2[iz, iy, ix] = np.inf
    d.write_map(capi.MAP_VOXELS, m2)
    d.process_scan(scan, tf)
    assert np.isinf(d.read_map()[iz, iy, ix])


def test_load_cloud_text_formats(oracle, tmp_path):
    # pc_loader.cpp:17-90
    p = tmp_path / "a.xyz"
    p.write_text("1 2 3\n\n  4.5\t5.5   6.5 7 8\r\nbad line\n-1e1 0 .5")
    np.testing.assert_array_equal(load_cloud(oracle, str(p)), np.float32([[1, 2, 3], [4.5, 5.5, 6.5], [-10, 0, 0.5]]))
    q = tmp_path / "b.pts"
    q.write_text("2\n1 1 1\n2 2 2\n")
    np.testing.assert_array_equal(load_cloud(oracle, str(q)), np.float32([[1, 1, 1], [2, 2, 2]]))
    with pytest.raises(Exception):
        load_cloud(oracle, str(tmp_path / "missing.xyz"))


def test_ingest_apriori_transform_and_centroids(oracle, tmp_path):
    # vofod_nodelet.cpp:214-226 + :306-345.  yaw 90 deg, t=(1,0,0), correction (0,0,0.5): p' = R*(p + (1,0,0.5)),
    # R = rot_z(90deg): (x,y,z) -> (-y, x, z) up to cos(pi/2 as float) = -4.37e-8.
    sp, dp = default_params(oracle)
    sp.voxel_size = 0.5
    sp.oparea_offset[:] = (0.0, 0.0, -1.0)  # map spans x,y in [-4,4], z in [-1,3]  (:208-212)
    sp.oparea_size[:] = (8.0, 8.0, 4.0)
    d = VoFOD(oracle, sp, dp)
    f = tmp_path / "map.xyz"
    # two points sharing a 0.5 m cell after the transform (centroid = their mean), one alone, one outside the map
    f.write_text("0.1 0.1 0.1\n0.3 0.2 0.2\n2.1 -1.1 1.0\n50 0 0\n")
    nl, nv = d.ingest_apriori(str(f), (1.0, 0.0, 0.0), 90.0, (0.0, 0.0, 0.5))
    assert (nl, nv) == (4, 3)
    m = d.read_map()
    inf = np.argwhere(np.isinf(m))
    # centroid 1: (-0.15, 1.2, 0.65) -> idx ((x+4)/0.5, (y+4)/0.5, (z+1)/0.5) = (7, 10, 3)
    # centroid 2: (1.1, 3.1, 1.5) -> (10, 14, 5);  the third (0, 51, 0.5) is outside the map
    assert sorted(map(tuple, inf.tolist())) == [(3, 10, 7), (5, 14, 10)]
    assert np.all(m[~np.isinf(m)] == m[0, 0, 0])


def test_rangefinder_ground_update(oracle):
    # vofod_nodelet.cpp:581-613: p = tf * (range, 0, 0); map(p) = (map(p) + scores/point) / 2.0, outside the area: untouched
    sp, dp = default_params(oracle)
    sp.voxel_size = 0.5
    sp.oparea_offset[:] = (0.0, 0.0, -1.0)  # map spans x,y in [-4,4], z in [-1,3]
    sp.oparea_size[:] = (8.0, 8.0, 4.0)
    d = VoFOD(oracle, sp, dp)
    init = d.read_map()[0, 0, 0]
    down = np.float32([[0, 0, 1, 1.1], [0, 1, 0, -0.3], [-1, 0, 0, 2.0]])  # the sensor's x axis points down
    assert d.update_ground(2.6, down) == capi.OK  # hits (1.1, -0.3, -0.6) -> idx (10, 7, 0)
    m = d.read_map()
    want = np.float32((np.float64(init) + dp.voxel_map__scores__point) / 2.0)
    assert m[0, 7, 10] == want and np.count_nonzero(m != init) == 1
    d.update_ground(2.6, down)
    assert d.read_map()[0, 7, 10] == np.float32((np.float64(want) + dp.voxel_map__scores__point) / 2.0)
    before = d.read_map()
    assert d.update_ground(9.0, down, allow=(capi.ERR_MAP_RANGE,)) == capi.ERR_MAP_RANGE  # z = -7: below the map
    np.testing.assert_array_equal(d.read_map(), before)
    # the reference's validity test is `range <= min && range >= max`: a range of 0 with min 0.1 still updates
    assert d.update_ground(0.0, down, min_range=0.1, max_range=10.0) == capi.OK
    assert np.count_nonzero(d.read_map() != before) == 1


def test_ouster_lut_and_mask_layout(oracle):
    # vofod_nodelet.cpp:358-372 + [3P] make_xyz_lut: pixel (u, v): encoder = 2 pi - v 2 pi / w, azimuth = -az[u], altitude = alt[u]
    w, h = 4, 2
    d, o = ouster_lut(oracle, w, h, azimuth_deg=[0.0, 90.0], altitude_deg=[0.0, 30.0], origin_mm=15.0)
    d, o = d.reshape(h, w, 3), o.reshape(h, w, 3)
    # row 0 (no beam offsets): unit vectors turning clockwise with the column, offsets (cos e, sin e, 0) - dir = 0
    np.testing.assert_allclose(d[0], [[1, 0, 0], [0, -1, 0], [-1, 0, 0], [0, 1, 0]], atol=1e-6)
    np.testing.assert_allclose(o[0], 0, atol=1e-9)
    # row 1: azimuth -90 deg, altitude 30 deg: column 0 looks along -y, tilted up; normalised in float (:369)
    c30, s30 = math.cos(math.radians(30)), math.sin(math.radians(30))
    np.testing.assert_allclose(d[1, 0], [0, -c30, s30], atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(d.reshape(-1, 3), axis=1), 1.0, atol=1e-6)
    # offset = ((cos e, sin e, 0) - dir) * 15 mm * 0.001 (range_unit): encoder 2 pi -> (1, 0, 0) - (0, -c30, s30)
    np.testing.assert_allclose(o[1, 0], np.array([1.0, c30, -s30]) * 0.015, atol=1e-7)
    # lidar_to_sensor transform: rotation by 180 deg about z + 36.18 mm up (the usual Ouster metadata): directions flip in x,y
    tf = np.diag([-1.0, -1.0, 1.0, 1.0])
    tf[2, 3] = 36.18
    d2, o2 = ouster_lut(oracle, w, h, [0.0, 90.0], [0.0, 30.0], origin_mm=15.0, tf=tf)
    np.testing.assert_allclose(d2.reshape(h, w, 3)[0, 0], [-1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(o2.reshape(h, w, 3)[0, 0], [0, 0, 0.03618], atol=1e-7)
    # load_mask :527-541: mask[((v + shift[u]) % w) * h + u] = image[u * w + v]; no image: all ones (:558)
    img = np.arange(1, 9, dtype=np.uint8).reshape(h, w)  # rows [1 2 3 4], [5 6 7 8]
    m = mask_layout(oracle, img, w, h, pixel_shift_by_row=[0, 1], mangle=True)
    #   column-major pairs (u=0, u=1) per shifted column vv: vv=0: (1, 8)  vv=1: (2, 5)  vv=2: (3, 6)  vv=3: (4, 7)
    assert m.tolist() == [1, 8, 2, 5, 3, 6, 4, 7]
    assert mask_layout(oracle, img, w, h, mangle=False).tolist() == list(range(1, 9))
    assert mask_layout(oracle, None, w, h).tolist() == [1] * 8


def test_sim_lut_formula(oracle):
    # vofod_nodelet.cpp:374-420: yaw = col*2pi/(w-1), pitch = row*vfov/(h-1) - vfov/2
    lut = sim_lut(oracle, 1024, 128, math.radians(45.0)).reshape(128, 1024, 3)
    vf = float(np.float32(math.radians(45.0)))
    np.testing.assert_allclose(lut[0, 0], [math.cos(-vf / 2), 0, math.sin(-vf / 2)], atol=1e-7)
    np.testing.assert_allclose(lut[127, 1023], [math.cos(vf / 2), 0, math.sin(vf / 2)], atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(lut, axis=-1), 1.0, atol=1e-6)
    yaw = 256 * 2 * math.pi / 1023
    np.testing.assert_allclose(lut[64, 256, :2] / np.linalg.norm(lut[64, 256, :2]), [math.cos(yaw), math.sin(yaw)], atol=1e-6)


def test_moie_boxes(oracle):
    f = oracle.extra("vofod_oracle_moie", C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5)
    # three collinear lattice points along (1,1,0): OBB centre = midpoint, diagonal = end-to-end length
    pts = np.zeros(3, dtype=capi.POINT_XYZR)
    pts["x"], pts["y"], pts["z"] = [0.25, 0.75, 1.25], [0.25, 0.75, 1.25], [0.25, 0.25, 0.25]
    mn, mx, ctr, eig = (np.zeros(3, dtype=np.float32) for _ in range(4))
    size = np.zeros(1, dtype=np.float32)
    f(capi.ptr(pts), 3, capi.ptr(mn), capi.ptr(mx), capi.ptr(ctr), capi.ptr(size), capi.ptr(eig))
    np.testing.assert_array_equal(mn, np.float32([0.25, 0.25, 0.25]))
    np.testing.assert_array_equal(mx, np.float32([1.25, 1.25, 0.25]))
    np.testing.assert_allclose(ctr, [0.75, 0.75, 0.25], atol=1e-6)
    assert size[0] == pytest.approx(math.sqrt(2.0), rel=1e-6)
    # [3P] covariance is normalised by n^2: sum((p-mean)^2) along the axis = 2*0.5 = 1.0 -> 1/9
    assert eig[0] == pytest.approx(1.0 / 9.0, rel=1e-5) and abs(eig[1]) < 1e-7
    # an L-shaped triple: asymmetric -> OBB centre differs from the centroid
    pts["x"], pts["y"], pts["z"] = [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 0.0]
    f(capi.ptr(pts), 3, capi.ptr(mn), capi.ptr(mx), capi.ptr(ctr), capi.ptr(size), capi.ptr(eig))
    # principal axis (1,-1)/sqrt2 spans [-1/sqrt2, 1/sqrt2]; the other (1,1)/sqrt2 spans proj {-.471,.236}
    np.testing.assert_allclose(ctr, [0.25, 0.25, 0.0], atol=1e-6)
    assert size[0] == pytest.approx(math.hypot(math.sqrt(2.0), math.sqrt(0.5)), rel=1e-5)

def _moie_call(oracle, pts_xyz, solver):
    f = oracle.extra("vofod_oracle_moie", C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5)
    setsolver = oracle.extra("vofod_oracle_set_obb_solver", C.c_int, [C.c_int])
    setsolver(solver)
    try:
        pts = np.zeros(len(pts_xyz), dtype=capi.POINT_XYZR)
        pts["x"], pts["y"], pts["z"] = pts_xyz[:, 0], pts_xyz[:, 1], pts_xyz[:, 2]
        mn, mx, ctr, eig = (np.zeros(3, dtype=np.float32) for _ in range(4))
        size = np.zeros(1, dtype=np.float32)
        f(capi.ptr(pts), len(pts), capi.ptr(mn), capi.ptr(mx), capi.ptr(ctr), capi.ptr(size), capi.ptr(eig))
        return ctr.copy(), float(size[0]), eig.copy()
    finally:
        setsolver(0)


def _lattice(cells, vs=0.25, off=(12.375, -7.125, 3.625)):
    return ((np.array(cells, dtype=np.float32) + np.float32(0.5)) * np.float32(vs) + np.array(off, dtype=np.float32)).astype(np.float32)


def test_obb_solver_restatement_is_a_correct_eigen_solver(oracle):
    """The restated Eigen::EigenSolver<Matrix3f> path (Hessenberg + shifted QR + back substitution, float) against numpy on
    clusters with a simple spectrum: eigenvalues as numpy's eigh of the same n^2-normalised covariance (float tolerance)"""
    rng = np.random.default_rng(3)
    for _ in range(300):
        p = rng.uniform(-3, 3, (int(rng.integers(4, 30)), 3)).astype(np.float32) * np.float32([1.0, 0.6, 0.3])
        _, _, eig = _moie_call(oracle, p, 0)
        c = (p - p.mean(0)).astype(np.float64)
        want = np.sort(np.linalg.eigvalsh(c.T @ c / len(p) ** 2))[::-1]
        np.testing.assert_allclose(eig, want, rtol=2e-4, atol=1e-6)


def test_obb_solvers_agree_where_the_spectrum_is_simple(oracle):
    """general small clusters (no repeated eigenvalue): the general solver PCL calls and an independent symmetric Jacobi solver
    in double give the same OBB (centre, diagonal) within 1e-3 m"""
    rng = np.random.default_rng(4)
    n = 0
    for _ in range(400):
        p = rng.uniform(-1, 1, (int(rng.integers(4, 12)), 3)).astype(np.float32) * np.float32([1.0, 0.55, 0.25]) + np.float32(rng.uniform(-40, 40, 3))
        (c0, s0, e0), (c1, s1, _) = _moie_call(oracle, p, 0), _moie_call(oracle, p, 1)
        if min(e0[0] - e0[1], e0[1] - e0[2]) < 0.05 * e0[0]:
            continue  # nearly repeated eigenvalue: the basis, hence the box, is the solver's choice
        n += 1
        np.testing.assert_allclose(c0, c1, atol=1e-3)
        assert abs(s0 - s1) < 1e-3
    assert n > 200


def test_obb_gates_on_degenerate_lattice_clusters(oracle):
    """The shapes MAV-sized clusters take (min_points 2): repeated eigenvalues everywhere.  Axis-aligned shapes have an exactly
    diagonal covariance: both solvers return the coordinate axes and identical boxes.  A repeated eigenvalue whose
    eigen-space is NOT axis aligned (the lattice tetrahedron) is where a symmetric solver and the general one pick different
    bases: the centre moves by centimetres - the reason product and oracle follow the EigenSolver path."""
    vs = 0.25
    max_size, max_explore = 3.0, 3.0  # config/detection_params.yaml:52,56
    import itertools

    aligned = {
        "2 points along x": [(0, 0, 0), (1, 0, 0)],
        "3 collinear": [(0, 0, 0), (1, 0, 0), (2, 0, 0)],
        "2x2 square": [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)],
        "2x2 square xz": [(0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 0, 1)],
        "2x2x2 cube": list(itertools.product((0, 1), repeat=3)),
        "3x3 square": [(i, j, 0) for i in range(3) for j in range(3)],
        "plus": [(1, 0, 0), (0, 1, 0), (1, 1, 0), (2, 1, 0), (1, 2, 0)],
    }
    want_size = {"2 points along x": vs, "3 collinear": 2 * vs, "2x2 square": math.sqrt(2) * vs, "2x2 square xz": math.sqrt(2) * vs,
                 "2x2x2 cube": math.sqrt(3) * vs, "3x3 square": 2 * math.sqrt(2) * vs, "plus": 2 * math.sqrt(2) * vs}
    for name, cells in aligned.items():
        p = _lattice(cells, vs)
        (c0, s0, _), (c1, s1, _) = _moie_call(oracle, p, 0), _moie_call(oracle, p, 1)
        np.testing.assert_allclose(c0, p.mean(0), atol=1e-5, err_msg=name)  # symmetric shapes: centre = centroid
        np.testing.assert_allclose(c0, c1, atol=1e-6, err_msg=name)
        assert s0 == pytest.approx(want_size[name], rel=1e-5), name
        assert s1 == pytest.approx(s0, rel=1e-6), name
        assert (s0 > max_size) == (s1 > max_size)
        assert int((s0 + max_explore) / vs) == int((s1 + max_explore) / vs)
    # diagonal pairs / triples: one non-zero eigenvalue, a two-fold zero one; the extent along the cluster's line decides
    for cells, length in (([(0, 0, 0), (1, 1, 0)], math.sqrt(2) * vs), ([(0, 0, 0), (1, 1, 1)], math.sqrt(3) * vs), ([(0, 0, 0), (1, 1, 0), (2, 2, 0)], 2 * math.sqrt(2) * vs)):
        p = _lattice(cells, vs)
        for solver in (0, 1):
            c, s, _ = _moie_call(oracle, p, solver)
            np.testing.assert_allclose(c, p.mean(0), atol=1e-4)
            assert s == pytest.approx(length, rel=1e-4)
    # the lattice tetrahedron: eigenvalues (a, a, b) with a skew two-dimensional eigen-space
    p = _lattice([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], vs)
    (c0, s0, e0), (c1, s1, e1) = _moie_call(oracle, p, 0), _moie_call(oracle, p, 1)
    assert e0[0] == pytest.approx(e0[1], rel=1e-5) and e0[1] > 3 * e0[2]
    assert np.abs(c0 - c1).max() > 0.02  # the solvers' boxes differ by centimetres ...
    assert abs(s0 - s1) < 0.01           # ... while the gated diagonal hardly moves
    assert (s0 > max_size) == (s1 > max_size) and int((s0 + max_explore) / vs) == int((s1 + max_explore) / vs)
    # whatever the basis, the box of the general solver contains the points: centre within the AABB, diagonal between the
    # largest pairwise distance and the AABB's diagonal + one voxel
    rng = np.random.default_rng(9)
    for _ in range(500):
        cells = np.unique(rng.integers(0, 3, size=(int(rng.integers(2, 10)), 3)), axis=0)
        if len(cells) < 2:
            continue
        p = _lattice(cells, vs, off=rng.uniform(-50, 50, 3))
        c, s, _ = _moie_call(oracle, p, 0)
        assert np.all(c >= p.min(0) - 1e-3) and np.all(c <= p.max(0) + 1e-3)
        far = max(np.linalg.norm(a - b) for a in p for b in p)
        assert far - 1e-3 <= s <= np.linalg.norm(p.max(0) - p.min(0)) * math.sqrt(3) + 1e-3


def test_check_sensor_params(oracle):
    """vofod_nodelet.cpp:1869-1917: the first pixel with mask != 0 and range != 0 is compared with the LUT (direction,
    distance = range * 0.001, unit LUT vector), each within 1e-3; a scan without such a pixel leaves the check open"""
    from vofod_amd.detector import check_sensor_params

    w, h = 4, 2
    lut = sim_lut(oracle, w, h, 0.5).reshape(h * w, 3)
    rng_mm = np.uint32([0, 0, 12345, 2000, 3000, 4000, 5000, 6000])
    xyz = (lut * (rng_mm[:, None].astype(np.float32) * np.float32(0.001))).astype(np.float32)

    def scan_of(p, r):
        return ScanData(x=np.ascontiguousarray(p[:, 0]), y=np.ascontiguousarray(p[:, 1]), z=np.ascontiguousarray(p[:, 2]), width=w, height=h,
                        intensity=np.zeros(h * w, np.float32), range=np.ascontiguousarray(r), stride_bytes=4)

    assert check_sensor_params(oracle, scan_of(xyz, rng_mm), lut) == (True, True)
    # the first two pixels have no return: the third decides.  1.5 mm off in range: |pt_dist - lut_dist| > 1e-3 -> mismatch
    bad = xyz.copy()
    bad[2] = lut[2] * np.float32(12.3465)
    assert check_sensor_params(oracle, scan_of(bad, rng_mm), lut) == (False, True)
    # 0.5 mm off: within the tolerance
    near = xyz.copy()
    near[2] = lut[2] * np.float32(12.3455)
    assert check_sensor_params(oracle, scan_of(near, rng_mm), lut) == (True, True)
    # direction off by 2e-3 rad: |pt_dir - lut_dir| = 2e-3 > 1e-3
    rot = xyz.copy()
    c, s = np.cos(2e-3), np.sin(2e-3)
    rot[2] = np.float32([c * xyz[2, 0] - s * xyz[2, 1], s * xyz[2, 0] + c * xyz[2, 1], xyz[2, 2]])
    assert check_sensor_params(oracle, scan_of(rot, rng_mm), lut) == (False, True)
    # masked out: pixel 2 is skipped, pixel 3 (consistent) decides although pixel 2 is wrong
    mask = np.uint8([1, 1, 0, 1, 1, 1, 1, 1])
    assert check_sensor_params(oracle, scan_of(bad, rng_mm), lut, mask=mask) == (True, True)
    # a LUT vector that is not normalised: 1 - |lut_dir| > 1e-3
    lut2 = lut.copy()
    lut2[2] *= np.float32(0.99)
    assert check_sensor_params(oracle, scan_of(xyz, rng_mm), lut2) == (False, True)
    # beam offsets are subtracted before the comparison
    offs = np.tile(np.float32([0.01, -0.02, 0.03]), (h * w, 1))
    assert check_sensor_params(oracle, scan_of(xyz + offs, rng_mm), lut, lut_offsets=offs) == (True, True)
    # no valid pixel at all: nothing checked, parameters not rejected
    assert check_sensor_params(oracle, scan_of(xyz, np.zeros(h * w, np.uint32)), lut) == (True, False)


@pytest.mark.parametrize("n_its", [0, 1, 3])
def test_sepclusters_two_islands_by_hand(oracle, n_its):
    """updateSeparatedBGClusters (vofod_nodelet.cpp:1126-1277) on a 9^3 toy map: leaf, positional counts (Q1), sums per
    cluster, the latch, the truncated-norm erase stencil around cast<int> centres, the erase weight after 0 (-> 1), 1 and 3
    detection iterations - tests/kat_cases.py derives every number from the cited lines"""
    import kat_cases

    kat_cases.sepclusters_case(oracle, n_its)


@pytest.mark.parametrize("new_rule", [True, False])
@pytest.mark.parametrize("n_its", [1, 3])
def test_raycast_update_three_rays_by_hand(oracle, new_rule, n_its):
    """raycast_cloud's accumulation and both update rules (vofod_nodelet.cpp:1455-1457, 1550-1604) on three axis rays with
    hand-summed path lengths; flagged voxels untouched, flags cleared"""
    import kat_cases

    kat_cases.raycast_case(oracle, new_rule, n_its)
