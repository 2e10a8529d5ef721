"""libfmrx.so and PyTorch in one process, in both load orders.

Both bring a HIP runtime: torch bundles libamdhip64.so / libhsa-runtime64.so in torch/lib, libfmrx uses the
ROCm installation's.  libfmrx records its runtime dependencies by their unversioned names (csrc/Makefile), so
ld.so resolves the second request to the library that is already mapped: ONE HIP and ONE HSA runtime in the
process whichever comes first.  (Round 1 recorded the sonames; `libfmrx first, torch second` then mapped two
runtimes and `import torch` hung on the GPU box.)  The CPU test checks the mapping; the GPU test runs both
orders for real, each in a child process under a timeout."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "software-defined-radio_amd", "lib", "libfmrx.so")

CHILD = r"""
import ctypes, os, re, sys
order, lib, use_gpu = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
if order == "fmrx_first":
    L = ctypes.CDLL(lib)
    import torch
else:
    import torch
    L = ctypes.CDLL(lib)
L.fmrx_device_count.restype = ctypes.c_int
n = L.fmrx_device_count()
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if re.search(r"libamdhip64|libhsa-runtime64", l)})
hip = [m for m in maps if "libamdhip64" in m]
hsa = [m for m in maps if "libhsa-runtime64" in m]
assert len(hip) == 1 and len(hsa) == 1, maps
if use_gpu:
    assert n >= 1 and torch.cuda.is_available()
    x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    assert float(x.sum().item()) == float((1 << 20) * ((1 << 20) - 1) // 2)
    import numpy as np
    out = np.zeros(4, np.float32)
    a = np.array([0.5, 1000.0, -3.0, 123456.0], np.float32)
    L.fmrx_diag_libm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    assert L.fmrx_diag_libm(0, a.ctypes.data, None, 4, out.ctypes.data) == 0
    assert abs(out[0] - np.sin(0.5)) < 1e-6
print("OK", order, n, hip, hsa)
"""


def _run(order, gpu):
    env = dict(os.environ)
    return subprocess.run([sys.executable, "-c", CHILD, order, LIB, "1" if gpu else "0"], capture_output=True, text=True,
                          timeout=300, env=env)


@pytest.mark.parametrize("order", ["fmrx_first", "torch_first"])
def test_one_hip_runtime_mapped(order):
    pytest.importorskip("torch")
    if not os.path.exists(LIB):
        pytest.skip("libfmrx.so not built")
    r = _run(order, gpu=False)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["fmrx_first", "torch_first"])
def test_both_load_orders_work_on_the_gpu(order):
    pytest.importorskip("torch")
    r = _run(order, gpu=True)          # a hang shows as subprocess.TimeoutExpired after 300 s
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
