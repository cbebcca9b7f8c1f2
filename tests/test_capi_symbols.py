"""The C-ABI library loads and exports every symbol include/vofod.h declares (no compute calls: no GPU here)."""
import subprocess
from pathlib import Path

import pytest

from vofod_amd import capi

ROOT = Path(__file__).resolve().parent.parent


def test_header_declares_the_expected_surface():
    names = capi.declared_entry_points()
    for n in ("create", "destroy", "reset", "process_scan", "process_batch", "raycast_begin", "raycast_finish", "sepclusters_begin",
              "sepclusters_finish", "voxel_grid_weighted", "voxel_grid_counted", "cluster", "load_apriori", "read_map", "write_map",
              "load_cloud", "sim_lut", "default_params", "set_dynamic_params", "get_status", "last_error_string"):
        assert n in names
    assert set(names) == set(capi._SIGS)


def test_hip_library_exports_every_declared_symbol():
    so = ROOT / "vofod_amd" / "csrc" / "libvofod_hip.so"
    if not so.exists():  # hipcc cross-compiles gfx950 without a GPU
        subprocess.run(["make", "-C", str(so.parent)], check=True, capture_output=True)
    lib = capi.Library(so, "vofod_")  # raises ImportError listing any missing symbol
    for n in capi.declared_entry_points():
        assert hasattr(lib, n)
    # plain C types only: parameter defaults are readable without touching a device
    sp, dp = capi.StaticParams(), capi.DynParams()
    lib.default_params(sp, dp)
    assert sp.voxel_size == pytest.approx(0.5) and dp.sepclusters__min_sure_points == 24
    assert sp.sensor_hrays == 1024 and sp.sensor_vrays == 128


def test_oracle_exports_the_same_surface(oracle):
    sp, dp = capi.StaticParams(), capi.DynParams()
    oracle.default_params(sp, dp)
    assert dp.ground_points_max_distance == 1.5 and dp.raycast__weight_coefficient == 0.003


def test_product_does_not_reference_the_oracle():
    """the product path must fail loudly rather than fall back: nothing under vofod_amd/ may mention oracle/"""
    for p in (ROOT / "vofod_amd").rglob("*"):
        if p.suffix in (".py", ".h", ".hip", ".cpp") and p.name != "capi.py":
            text = p.read_text()
            assert "libvofod_oracle" not in text and "oracle/" not in text.replace("oracle/.", ""), p


def test_library_loader_fails_loudly_when_missing(monkeypatch, tmp_path):
    import vofod_amd

    monkeypatch.setattr(vofod_amd, "_lib", None)
    monkeypatch.setattr(vofod_amd, "LIB_PATH", tmp_path / "libvofod_hip.so")
    with pytest.raises(ImportError):
        vofod_amd.library()


def test_bench_byte_model_names_the_kernels_of_the_batched_path():
    """bench.py prices the path kernels by name: a kernel of the batched fast path that is launched by the driver but missing
    from the algorithmic-bytes table would silently drop out of the roofline (k_key1, the streaming kernel of rounds 2-4, once did)"""
    import re

    root = ROOT
    bench = (root / "bench.py").read_text()
    table = bench[bench.index("alg = {"):bench.index("for nm, k in kernels.items():")]
    priced = set(re.findall(r'"(k_[a-z0-9_]+)', table))
    driver = (root / "vofod_amd" / "csrc" / "vofod_hip.hip").read_text()
    launched = set(re.findall(r"KLAUNCH(?:_LDS)?\(h, (?:vk::)?(k_[a-z0-9_]+)", driver))
    launched |= set(re.findall(r'VOFOD_FRAME_LAUNCH\("(k_[a-z0-9_]+)"', driver))  # (the frame kernel's instantiations are profiled under their algorithmic names)
    names = ("k_bbox", "k_frame_lds_full", "k_frame_lds_far")
    streaming = {k for k in launched if k in names}
    assert streaming == set(names)
    assert streaming <= priced, streaming - priced
    prefixes = bench[bench.index("path_prefixes = ("):]
    prefixes = prefixes[: prefixes.index(")")]
    for k in streaming:
        assert any(k.startswith(p) for p in re.findall(r'"(k_[a-z0-9_]+)"', prefixes)), k


def test_kernel_source_hash_ignores_comments_and_layout(tmp_path, monkeypatch):
    """bench.py reports PMC `traffic` only for the kernel sources it was measured on; a corrected comment or a re-wrapped line is
    not a new kernel, a changed token is."""
    import importlib
    import sys

    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    src = tmp_path / "vofod_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "kernels_a.h").write_text("// a comment\nint f(int x) { return x + 1; }  /* another */\n")
    (src / "driver.hip").write_text("int host_side() { return 1; }\n")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    h0 = bench.kernel_source_sha()
    (src / "kernels_a.h").write_text("// a corrected comment\nint f(int x)\n{\n  return x + 1;\n}\n")
    assert bench.kernel_source_sha() == h0
    (src / "driver.hip").write_text("int host_side() { return 2; }\n")  # the host driver holds no kernel of the path
    assert bench.kernel_source_sha() == h0
    (src / "kernels_a.h").write_text("int f(int x) { return x + 2; }\n")
    assert bench.kernel_source_sha() != h0
