import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _oracle_path() -> Path:
    so = ROOT / "oracle" / "libvofod_oracle.so"
    srcs = [ROOT / "oracle" / n for n in ("oracle.cpp", "algorithms.hpp", "voxel_map.hpp")] + [ROOT / "include" / "vofod.h"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs if s.exists()):
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return so


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure) bound through the same ctypes mirror as the product."""
    from vofod_amd import capi

    return capi.Library(_oracle_path(), "vofod_oracle_")


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests call through its C-ABI."""
    import os

    if os.environ.get("VOFOD_TEST_HARNESS_SELFCHECK"):
        # harness self-check on a machine without a GPU: run the parity tests oracle-vs-oracle
        from vofod_amd import capi

        return capi.Library(_oracle_path(), "vofod_oracle_")
    import vofod_amd

    return vofod_amd.library()
