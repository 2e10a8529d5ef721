#!/usr/bin/env python3
"""gpurun_out/profr3 (tools/gpu_profile_r3.sh) -> the artifacts to commit: gpurun_out/profr3/{00_bench_command.txt, fe_traffic.json,
03_bench_default_run.json, 00_bench_command_kernel_stats.csv}; copy them to profiles/round3/ and profiles/fe_traffic.json."""
import collections, csv, glob, json, shutil
out = 'gpurun_out/profr3'
L = []
P = L.append
P("rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-side-legs --steps 30 --warmup 5   (MI355X, 1 GPU, round 3)")
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, out + '/00_bench_command_kernel_stats.csv')
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 6: P("  {Name:.100s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**r))
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "mono_fused" in r["Kernel_Name"]]
    if d:
        P(f"== kernel trace, mono_fused_kernel: launches={len(d)} avg_ns(all, incl. the 400 ms settle phase)={sum(d)/len(d):.0f}  "
          f"avg_ns(last 30 = the timed region)={sum(d[-30:])/30:.0f}")
bj = None
for name in ("bench_under_rocprof", "bench_default"):
    try:
        line = [l for l in open(f"{out}/{name}.log") if l.startswith('{')][-1]
    except Exception:
        continue
    j = json.loads(line); r = j["roofline"]
    if name == "bench_default":
        bj = j
        open(out + '/03_bench_default_run.json', 'w').write(line)
    P(f"== {name}: value {j['value']} MS/s, ms_per_step {j['ms_per_step']}, HIP-event avg launch {r['avg_launch_ms']} ms over "
      f"{r['launches_timed']} launches, achieved {r['achieved']} GB/s, frac {r['frac']}, blocks {j['config']['blocks_per_step']}, "
      f"library {j['config']['library']['version']}")
tr = {}
for c, d_ in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{d_}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "mono_fused" in r["Kernel_Name"]: agg["fe"].append(float(r["Counter_Value"]))
    if agg["fe"]:
        tr[c] = sum(agg["fe"]) / len(agg["fe"])
        P(f"== {c} per dispatch of mono_fused_kernel: {tr[c]:.1f} KB (n={len(agg['fe'])})")
if len(tr) == 2 and bj:
    n = bj["config"]["samples_per_step_per_gpu"]
    hbm = int(tr["FETCH_SIZE"] * 1024 * 2 + tr["WRITE_SIZE"] * 1024)
    alg = int(n * 2.04)
    t = {"blocks": bj["config"]["blocks_per_step"], "output": "s16", "round": "round 3", "kernel": "mono_fused_kernel<101,10,101,5>",
         "lib_src": bj["config"]["library"]["version"].split("src:")[-1], "fetch_size_kb": tr["FETCH_SIZE"], "write_size_kb": tr["WRITE_SIZE"],
         "fetch_correction": 2.0, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch of `python3 bench.py --no-cpu-baseline --no-side-legs "
                 "--steps 5 --warmup 2 --settle-ms 0`; gfx950 FETCH_SIZE counts wide streaming reads at half (MI355X_MICROARCH.md, HBM) -> x2; "
                 "WRITE_SIZE exact; lib_src = the build (hash of the library's sources) the counters were collected on"}
    json.dump(t, open(out + '/fe_traffic.json', 'w'), indent=1)
    P(f"== HBM traffic per launch: FETCH_SIZE {tr['FETCH_SIZE']:.1f} KB x 2 (gfx950 correction) + WRITE_SIZE {tr['WRITE_SIZE']:.1f} KB = {hbm} B; "
      f"algorithmic {alg} B; ratio {hbm / alg:.3f}")
open(out + '/00_bench_command.txt', 'w').write("\n".join(L) + "\n")
print("\n".join(L))
