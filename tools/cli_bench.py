#!/usr/bin/env python3
"""Steady-state rate of the drop-in CLI (u8 stdin -> s16 stdout) on a ~1 GB input in /dev/shm."""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
blk = synth.synth_fm_u8(3 * 1_024_000)
path = "/dev/shm/fmrx_cli_in.u8"
reps = 160
with open(path, "wb") as f:
    for _ in range(reps):
        f.write(blk.tobytes())
n = reps * 3 * 1_024_000
exe = os.path.join(os.path.dirname(fmrx.LIB_PATH), "fmrx_project")
for args in (["0", "1"], ["0", "1", "--blocks-per-call", "20"], ["0", "2", "--blocks-per-call", "20"], ["2", "1", "--blocks-per-call", "18"]):
    t0 = time.perf_counter()
    with open(path, "rb") as f:
        r = subprocess.run([exe] + args, stdin=f, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    print(f"fmrx_project {' '.join(args)}: {n / dt / 1e6:.0f} MS/s ({dt:.2f} s, rc {r.returncode}) = {n / dt / 2.4e6:.0f} x real time", flush=True)
os.remove(path)
