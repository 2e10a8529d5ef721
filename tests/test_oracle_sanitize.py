"""The oracle (CPU restatement) under AddressSanitizer + UBSan: sanitizers run on
the CPU build only (the GPU pool offers none).  The reference's own
convolveBlockFastFIR fails this (heap-buffer-overflow); the
restatement must pass with exact-size buffers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_selftest_under_asan_ubsan(tmp_path):
    exe = tmp_path / "oracle_selftest"
    src = [os.path.join(ROOT, "oracle", f) for f in ("selftest.c", "fm_oracle.c")]
    r = subprocess.run(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "oracle"),
                        "-o", str(exe)] + src + ["-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest ok" in r.stdout
