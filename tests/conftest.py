import os
import sys

# before anything imports torch: the HIP runtime reads this when it starts (mygauhuman_amd/__init__.py, graph.py)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _install_native_backtrace():
    """tests/native_bt.c: a fatal signal prints the native stack and the name of the thread that raised it (faulthandler only
    knows Python frames).  Best effort: the suite runs without it when gcc is missing."""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, lib = os.path.join(here, "native_bt.c"), os.path.join(here, "_native_bt.so")
    try:
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", lib, src])
        return ctypes.CDLL(lib).native_bt_install() == 0
    except Exception as ex:  # noqa: BLE001
        print(f"[conftest] native backtrace handler not installed: {ex}")
        return False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _install_native_backtrace()


@pytest.fixture(scope="session", autouse=True)
def _stress_tuning():
    """GSR_TEST_TUNING="key=value,key=value": process-default tuning knobs for the whole run, e.g. "blend_segments=4,blend_tail_cut=8"
    to push every long enough list of the parity tests through the segmented backward (tests that pin a knob themselves still do)."""
    spec = os.environ.get("GSR_TEST_TUNING", "")
    if spec:
        from mygauhuman_amd import _lib
        for kv in spec.split(","):
            k, v = kv.split("=")
            _lib.set_tuning(k.strip(), int(v))
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    orc.set_threads(min(8, os.cpu_count() or 1))
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
