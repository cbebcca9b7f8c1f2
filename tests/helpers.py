"""Shared helpers of the parity tests: drive two implementations of include/vofod.h side by side."""
import numpy as np

from vofod_amd import capi, synth
from vofod_amd.detector import VoFOD, default_params


def make_pair(oracle, hip, sensor="os1-16", voxel_size=0.5, max_batch=1, lut=None, mask=None, **dyn):
    """An oracle detector and a HIP detector with identical parameters.  `lut` = (directions, offsets) and `mask` replace
    the simulated sensor model (vofod_nodelet.cpp:358-372, 506-560) on both sides."""
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    dets = []
    for lib in (oracle, hip):
        sp, dp = default_params(lib)
        sp.voxel_size = voxel_size
        sp.sensor_hrays, sp.sensor_vrays = w, h
        sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
        sp.max_batch_frames = max_batch
        for k, v in dyn.items():
            setattr(dp, k, v)
        dets.append(VoFOD(lib, sp, dp, lut_directions=None if lut is None else lut[0], lut_offsets=None if lut is None else lut[1], mask=mask))
    return dets


def sync_maps(src: VoFOD, dst: VoFOD):
    for which in (capi.MAP_VOXELS, capi.MAP_FLAGS):
        dst.write_map(which, src.read_map(which))


def assert_scan_debug_equal(d_ref, d_hip):
    """bit-exact voxel indices / cluster membership; tolerance only on the OBB-derived floats."""
    assert d_hip["n_input_after_crop"] == d_ref["n_input_after_crop"]
    assert d_hip["n_bg_voxels"] == d_ref["n_bg_voxels"]
    assert d_hip["background_pts_sufficient"] == d_ref["background_pts_sufficient"]
    np.testing.assert_array_equal(d_hip["weighted"].view(np.uint32), d_ref["weighted"].view(np.uint32))
    np.testing.assert_array_equal(d_hip["labels"], d_ref["labels"])
    cr, ch = d_ref["clusters"], d_hip["clusters"]
    assert len(cr) == len(ch)
    for k in ("first_member", "n_points", "is_close", "cclass"):
        np.testing.assert_array_equal(ch[k], cr[k], err_msg=k)
    np.testing.assert_array_equal(ch["aabb_min"], cr["aabb_min"])
    np.testing.assert_array_equal(ch["aabb_max"], cr["aabb_max"])
    ev = ~np.isnan(ch["obb_size"]) & ~np.isnan(cr["obb_size"])
    # tolerance: OBB centre/diagonal within 1e-3 m (eigen-solver rounding)
    np.testing.assert_allclose(ch["obb_size"][ev], cr["obb_size"][ev], atol=1e-3)
    evc = ~np.isnan(ch["obb_center"][:, 0])
    np.testing.assert_allclose(ch["obb_center"][evc], cr["obb_center"][evc], atol=1e-3)


def assert_detections_equal(a, b):
    assert len(a) == len(b)
    for k in ("id", "frame", "n_points"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # tolerance (north_star: "within a stated float tolerance on centroids/confidences"):
    # positions 1e-3 m, confidence / probability / covariance 1e-5 relative
    np.testing.assert_allclose(a["position"], b["position"], atol=1e-3)
    np.testing.assert_allclose(a["confidence"], b["confidence"], rtol=1e-5, atol=1e-300)
    np.testing.assert_allclose(a["detection_probability"], b["detection_probability"], rtol=1e-5)
    np.testing.assert_allclose(a["covariance"], b["covariance"], rtol=1e-5)


def far_view(d):
    """The far-only view of a full debug dict: what the production path of a read-only batch computes (include/vofod.h,
    vofod_scan_debug::far_only) - only the far clusters are ever used (vofod_nodelet.cpp:727-748, :946-963).  The cluster
    table keeps its (canonical) order, labels outside the far clusters become LABEL_NONE."""
    cl = d["clusters"]
    far = cl[cl["is_close"] == 0]
    lab = d["labels"].copy()
    lab[~np.isin(lab, far["first_member"])] = capi.LABEL_NONE
    out = dict(d)
    out["clusters"] = far
    out["labels"] = lab
    return out
