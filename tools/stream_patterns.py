#!/usr/bin/env python3
"""Pure streaming-read rate of the LDS-DMA ring by how the input is dealt to the 2 048 waves: one contiguous run per wave
(method 1) vs chunks of C steps of 3 KiB dealt round-robin (method C + 1).  The spacing between concurrently read addresses
decides: 5.1 TB/s ... 6.8 TB/s on one box."""
import importlib, sys, torch, time
sys.path.insert(0, '.')
fmrx = importlib.import_module("software-defined-radio_amd")
n = 512 * 1024 * 1024
d = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
chunks = [int(a) for a in sys.argv[1:]] or [0, 1, 8, 16, 24, 32, 40, 43, 48, 56, 64, 72, 80, 83, 84, 85, 86]
for c in chunks:
    m = 1 if c == 0 else c + 1
    for _ in range(30): fmrx.diagStreamRead(d.data_ptr(), n, m, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): fmrx.diagStreamRead(d.data_ptr(), n, m, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print(f"chunk {c:3d} steps = {c * 3072:7d} B between neighbouring waves: {n / dt / 1e9:.0f} GB/s", flush=True)
