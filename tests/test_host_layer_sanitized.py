"""The HOST side of libgsr under sanitizers: every .hip file of the library is compiled host-only
(`hipcc --offload-host-only`) with -fsanitize=address,undefined and with -fsanitize=thread, linked against a stand-in for the
twenty HIP runtime entry points the library imports (tests/host_san/hip_stub.cpp: "device memory" = malloc, kernels are not
executed, launch geometry is validated) and driven through the C ABI by tests/host_san/driver.cpp: arena carving vs the
reported byte counts, validation / error strings, every binning mode / wave count / reduction variant / deterministic mode
over ragged, one-pixel and full-size shapes, the asynchronous entry points, the introspection selectors, and four threads on
four streams hammering the option map, the stage-profile store and the error strings while the main thread switches both.
The GPU pool has no sanitizers (gpurun refuses them); this is what can be sanitized of the product itself."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SAN = os.path.join(HERE, "host_san")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc for the host-only build")


def _build(tag, flags):
    out = os.path.join(SAN, "build", tag)
    shutil.rmtree(out, ignore_errors=True)
    r = subprocess.run([os.path.join(SAN, "build.sh"), out] + flags, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return os.path.join(out, "driver")


def _run(exe, args, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=env)
    tail = (r.stdout + r.stderr)[-6000:]
    assert r.returncode == 0, tail
    assert "0 failures" in r.stdout, tail
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, tail


def test_host_layer_under_asan_ubsan():
    exe = _build("asan", ["-fsanitize=address,undefined"])
    _run(exe, [], {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})


def test_host_layer_under_tsan():
    exe = _build("tsan", ["-fsanitize=thread"])
    probe = subprocess.run([exe, "threads"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if "unexpected memory mapping" in probe.stderr:  # ThreadSanitizer cannot start under this kernel's ASLR settings
        pytest.skip("ThreadSanitizer does not run in this environment")
    tail = (probe.stdout + probe.stderr)[-6000:]
    assert probe.returncode == 0 and "0 failures" in probe.stdout and "ThreadSanitizer" not in probe.stderr, tail
