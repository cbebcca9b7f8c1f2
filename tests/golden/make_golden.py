#!/usr/bin/env python3
"""Generates tests/golden/os1_16_sequence.npz with the CPU oracle.

The reference itself cannot be built or run in this image, so these vectors are produced by the build's own
oracle; they pin the oracle against regressions and let the GPU tests run without it.  Inputs are stored
explicitly (two seeded OS1-16 scans + poses); expected outputs: weighted cloud, labels, cluster table and the
sparse set of map / flag voxels each scan changed.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from vofod_amd import capi, synth  # noqa: E402
from helpers import make_pair  # noqa: E402


def main():
    oracle = capi.Library(ROOT / "oracle" / "libvofod_oracle.so", "vofod_oracle_")
    det, _ = make_pair(oracle, oracle, "os1-16", 0.5)
    scene = synth.make_scene(7, n_targets=2)
    scans = synth.scan_sequence(scene, "os1-16", 2, seed0=70)
    out = {"voxel_size": np.float32(0.5), "seed_value": np.float32(-185.0)}
    # a hand-placed background patch so that close and far clusters both occur (written through write_map)
    m0 = det.read_map()
    m0[2:4, 40:160, 20:140] = -185.0
    det.write_map(capi.MAP_VOXELS, m0)
    out["seed_box"] = np.array([2, 4, 40, 160, 20, 140], dtype=np.int32)
    for k, s in enumerate(scans):
        before = det.read_map().copy()
        dets, dbg = det.process_scan(s.scan, s.tf, debug=True)
        after, flags = det.read_map(), det.read_map(capi.MAP_FLAGS)
        changed = np.flatnonzero((after != before).ravel() | (flags.ravel() != 0)).astype(np.uint32)
        out[f"s{k}_x"], out[f"s{k}_y"], out[f"s{k}_z"] = s.x, s.y, s.z
        out[f"s{k}_tf"] = s.tf
        out[f"s{k}_weighted"] = dbg["weighted"]
        out[f"s{k}_labels"] = dbg["labels"]
        out[f"s{k}_clusters"] = dbg["clusters"]
        out[f"s{k}_n_in"] = np.uint64(dbg["n_input_after_crop"])
        out[f"s{k}_n_bg"] = np.uint64(dbg["n_bg_voxels"])
        out[f"s{k}_changed_idx"] = changed
        out[f"s{k}_changed_map"] = after.ravel()[changed]
        out[f"s{k}_changed_flags"] = flags.ravel()[changed]
        det.write_map(capi.MAP_FLAGS, np.zeros_like(flags))
    np.savez_compressed(ROOT / "tests" / "golden" / "os1_16_sequence.npz", **out)
    print("written", (ROOT / "tests" / "golden" / "os1_16_sequence.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
