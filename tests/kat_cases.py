"""Hand-derived known-answer cases of the two stateful map roles, written against the C-ABI so that the same case pins the
CPU oracle (tests/test_oracle_kat.py) and the HIP library (tests/test_gpu_kat.py).  The reference ships no fixtures: every
expected value is derived by hand from the cited lines of the reference (paths relative to its tree); where a whole map is
compared, the expectation is rebuilt here from the reference's rule in a few explicit lines, independent of both
implementations.

Toy world: voxel size 1 m, operation area 8 x 8 x 8 m with its corner at the origin -> a 9 x 9 x 9 map, offset (0,0,0)
(voxel_map.cpp:15-16: sizes = ceil(dim / vs) + 1), voxel (i,j,k) covers [i,i+1) x [j,j+1) x [k,k+1)."""
import numpy as np

from vofod_amd import capi
from vofod_amd.detector import ScanData, VoFOD, default_params

W, H = 8, 2  # a sensor of 16 rays
RAY = np.float32(-1000.0)  # voxel_map/scores/ray (detection_params.yaml)


def toy(lib, lut=None, mask=None, **dyn):
    sp, dp = default_params(lib)
    sp.voxel_size = 1.0
    sp.oparea_offset[:] = (4.0, 4.0, 0.0)  # centre x, centre y, bottom z (vofod_nodelet.cpp:197-204)
    sp.oparea_size[:] = (8.0, 8.0, 8.0)
    sp.sensor_hrays, sp.sensor_vrays = W, H
    for k, v in dyn.items():
        setattr(dp, k, v)
    d = VoFOD(lib, sp, dp, lut_directions=lut, mask=mask)
    assert d.map_size == (9, 9, 9) and d.map_offset == (0.0, 0.0, 0.0)
    return d


def empty_scan():
    """no-return pixels only ((0,0,0), range 0: inside the exclude box, dropped by the first crop, vofod_nodelet.cpp:625-636):
    a detection iteration that touches no voxel (++m_detection_its :949)"""
    n = W * H
    z = np.zeros(n, dtype=np.float32)
    return ScanData(x=z, y=z.copy(), z=z.copy(), width=W, height=H, intensity=np.zeros(n, dtype=np.float32), range=np.zeros(n, dtype=np.uint32))


IDENT_AT = lambda t: np.float32([[1, 0, 0, t[0]], [0, 1, 0, t[1]], [0, 0, 1, t[2]]])  # noqa: E731


def sepclusters_case(lib, n_its):
    """updateSeparatedBGClusters, vofod_nodelet.cpp:1126-1277, with max_bg_distance 2 m -> max_dist_idx 2.0, max_voxel_dist 2
    (:1142-1143), leaf of the counted grid lsz = max(2 - 1, 0) = 1 (:1162): every background voxel is its own downsampled voxel.

    Two islands: A = {(6,1,1), (6,1,2)} holds SURE values (+5 > sure_obstacles -0.1), B = {(1,6,6), (1,6,7)} unsure ones
    (-100: above new_obstacles -300, below -0.1).  voxelsAsVoxelPC emits x outer / y / z inner (voxel_map.cpp:187-212):
    input order B0 B1 A0 A1.  VoxelGridCounted sorts by idx = i + j*dx + k*dx*dy relative to the minimum (1,1,1), div = (6,6,7)
    (voxel_grid_counted.cpp:109-113, 136-143): A0 -> 5, A1 -> 41, B0 -> 210, B1 -> 246: sorted order A0 A1 B0 B1.  The count of an
    output voxel is taken over INPUT positions [first, last) of its run in the SORTED vector (:185-187): A0 <- input[0] =
    B0 (unsure: 0), A1 <- input[1] = B1 (0), B0 <- input[2] = A0 (sure: 1), B1 <- input[3] = A1 (1).
    Clusters under tolerance 2 (:1171): {A0, A1} with sum 0, {B0, B1} with sum 2 (:1177-1183).  min_sure_points = 2: one sure cluster ->
    the latch is set (:1188-1206) and the UNSURE cluster - the island whose voxels are in fact the sure ones - is erased.

    Erase (:1209-1272): w1 = clamp(0.5^max(its diff, 1), 0, 1), every downsampled voxel of the cluster at centre.cast<int>()
    = (6,1,1) / (6,1,2) applies m = w1 m + (1 - w1) ray to every in-limits voxel at an offset of [-2,2]^3 whose TRUNCATED integer
    norm is <= 2.0 (:1223-1237; Eigen's integer norm(): squared norms up to 8 pass, 9 and 12 do not: 125 - 24 - 8 = 93 offsets).
    A voxel inside both stencils is updated twice."""
    d = toy(lib, sepclusters__max_bg_distance=2.0, sepclusters__min_sure_points=2)
    m = d.read_map(capi.MAP_VOXELS)  # [z, y, x]
    init = np.float32(m[0, 0, 0])
    assert (m == init).all() and init < -300.0
    for (x, y, z), v in {(6, 1, 1): 5.0, (6, 1, 2): 5.0, (1, 6, 6): -100.0, (1, 6, 7): -100.0}.items():
        m[z, y, x] = v
    d.write_map(capi.MAP_VOXELS, m)
    st, sure = d.sepclusters_begin()
    assert st == capi.OK and sure  # the latch: the cluster at B carries the two sure counts
    for _ in range(n_its):
        dets = d.process_scan(empty_scan(), IDENT_AT((4.5, 4.5, 4.5)))
        assert len(dets) == 0
    assert d.status().detection_its == n_its
    d.sepclusters_finish()
    got = d.read_map(capi.MAP_VOXELS)
    # the rule, restated: offsets, clipping, float32 arithmetic in the reference's order
    w1 = np.float32(0.5) ** max(n_its, 1)
    w1 = np.float32(min(max(w1, 0.0), 1.0))
    w2 = np.float32(1.0) - w1
    offs = [(a, b, c) for a in range(-2, 3) for b in range(-2, 3) for c in range(-2, 3) if int(np.sqrt(float(a * a + b * b + c * c))) <= 2.0]
    assert len(offs) == 93
    want = m.copy()
    hits = np.zeros_like(want, dtype=np.int32)
    for cx, cy, cz in ((6, 1, 1), (6, 1, 2)):
        for a, b, c in offs:
            x, y, z = cx + a, cy + b, cz + c
            if 0 <= x < 9 and 0 <= y < 9 and 0 <= z < 9:
                want[z, y, x] = np.float32(w1 * want[z, y, x]) + np.float32(w2 * RAY)
                hits[z, y, x] += 1
    np.testing.assert_array_equal(got, want)
    # spot values by hand (its diff 1: w1 = 0.5; 3: w1 = 0.125)
    once = {1: -870.0, 3: -967.5}[max(n_its, 1)] if init == -740.0 else None
    if once is not None:
        assert got[4, 1, 6] == np.float32(once)  # (6,1,4): offset (0,0,2) of (6,1,2) only ((0,0,3) of (6,1,1) is outside the cube)
        assert got[2, 3, 8] == np.float32(once)  # (8,3,2): (2,2,0) of (6,1,2), |.|^2 = 8 -> 2; (2,2,1) of (6,1,1), |.|^2 = 9 -> 3: not
        assert got[3, 3, 8] == init              # (8,3,3): (2,2,1) and (2,2,2): neither
    twice = {1: -748.75, 3: -984.296875}[max(n_its, 1)]  # 5 -> 0.5*5 - 500 = -497.5 -> -748.75; 0.125*5 - 875 = -874.375 -> 0.125*(-874.375) - 875
    assert got[1, 1, 6] == np.float32(twice) and hits[1, 1, 6] == 2
    assert got[6, 6, 1] == np.float32(-100.0) and got[7, 6, 1] == np.float32(-100.0)  # the cluster that counted as sure stays
    assert int((hits > 0).sum()) == int((got != m).sum())
    return d


def raycast_case(lib, new_rule, n_its):
    """raycast_cloud, vofod_nodelet.cpp:1397-1605, three rays from the centre of voxel (2,2,2) along +x, +y, +z.

    Ray length = min(range - voxel size, max_distance) (:1455-1457): ranges 4.5 / 3.0 / 2.5 m -> 3.5 / 2.0 / 1.5 m.  forEachRay
    (voxel_map.cpp:229-263) from a voxel centre: half a voxel, then whole voxels: +x: (2,2,2) 0.5, (3,2,2) 1, (4,2,2) 1, (5,2,2) 1;
    +y: (2,2,2) 0.5, (2,3,2) 1, (2,4,2) 0.5; +z: (2,2,2) 0.5, (2,2,3) 1.  Accumulator: (2,2,2) = 1.5, the rest as listed.
    Pixels with range 0 are skipped when the mask does not mark them (:1449).

    Update sweep after `n_its` detection iterations (:1539): voxels with flag 0 and accumulator r > 0 only (:1559/:1582), flags
    cleared afterwards (:1602).  New rule (:1550-1573): weighting_factor = coefficient / (sqrt(3) * vs); the coefficient is set to
    float(sqrt(3)) so the factor is exactly 1: w1 = 2^(-its * r) (std::pow(int, float): evaluated in double, stored as float),
    m = w1 m + (1 - w1) ray.  Old rule (:1574-1601): max = 1.5, w = coefficient * sqrt(r / max), w1 = clamp((1 - w)^its, 0, 1);
    coefficient 0.5: the sensor's voxel gets w = 0.5.  Voxel (4,2,2) carries flag 2 (a point of the scan, :946): untouched."""
    n = W * H
    lut = np.zeros((n, 3), dtype=np.float32)
    lut[:, 0] = 1.0
    lut[0], lut[1], lut[2] = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    coef = float(np.float32(np.sqrt(3.0))) if new_rule else 0.5
    d = toy(lib, lut=lut, mask=np.zeros(n, dtype=np.uint8), raycast__new_update_rule=1 if new_rule else 0, raycast__weight_coefficient=coef, raycast__max_distance=20.0,
            raycast__min_intensity=0.0)
    m0 = d.read_map(capi.MAP_VOXELS)
    init = np.float32(m0[0, 0, 0])
    flags = d.read_map(capi.MAP_FLAGS)
    flags[2, 2, 4] = 2.0
    d.write_map(capi.MAP_FLAGS, flags)
    rng = np.zeros(n, dtype=np.uint32)
    rng[0], rng[1], rng[2] = 4500, 3000, 2500
    z = np.zeros(n, dtype=np.float32)
    scan = ScanData(x=z, y=z.copy(), z=z.copy(), width=W, height=H, intensity=np.full(n, 10.0, dtype=np.float32), range=rng)
    tf = IDENT_AT((2.5, 2.5, 2.5))
    d.raycast_begin(scan, tf)
    acc = d.read_map(capi.MAP_RAYCAST)
    want_acc = {(2, 2, 2): 1.5, (3, 2, 2): 1.0, (4, 2, 2): 1.0, (5, 2, 2): 1.0, (2, 3, 2): 1.0, (2, 4, 2): 0.5, (2, 2, 3): 1.0}
    exp = np.zeros_like(acc)
    for (x, y, zz), v in want_acc.items():
        exp[zz, y, x] = v
    np.testing.assert_allclose(acc, exp, rtol=0, atol=1e-6)  # (float sums of at most three segments)
    for _ in range(n_its):
        d.process_scan(empty_scan(), IDENT_AT((4.5, 4.5, 4.5)))
    d.raycast_finish()
    got = d.read_map(capi.MAP_VOXELS)
    want = m0.copy()
    its = np.float32(n_its)
    for (x, y, zz), r in want_acc.items():
        if (x, y, zz) == (4, 2, 2):
            continue  # flagged by the scan: no ray update
        r = np.float32(r)
        if new_rule:
            n_int = np.float32(1.0) * r
            w1 = np.float32(np.power(2.0, float(-its * n_int)))  # std::pow(2, float): double arithmetic, float result
        else:
            w = np.float32(coef) * np.sqrt(np.float32(r / np.float32(1.5)))
            w1 = np.float32(min(max(float(np.power(np.float32(1.0) - w, its)), 0.0), 1.0))
        w2 = np.float32(1.0) - w1
        want[zz, y, x] = np.float32(w1 * want[zz, y, x]) + np.float32(w2 * RAY)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)
    assert got[2, 2, 4] == init  # the flagged voxel
    assert int((got != m0).sum()) == 6
    if new_rule and n_its == 1 and init == -740.0:
        assert got[2, 2, 3] == np.float32(-870.0)  # r = 1: w1 = 0.5
        np.testing.assert_allclose(got[2, 2, 2], 0.35355339 * -740.0 + (1 - 0.35355339) * -1000.0, rtol=1e-6)  # r = 1.5: 2^-1.5
    if (not new_rule) and n_its == 1 and init == -740.0:
        assert got[2, 2, 2] == np.float32(-870.0)  # r = max: w = 0.5
    assert not d.read_map(capi.MAP_FLAGS).any()  # :1602
    return d
