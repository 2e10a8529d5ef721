#!/usr/bin/env python3
"""gpurun_out/profr2 (tools/gpu_profile_r2.sh) -> profiles/round2/*, profiles/fe_traffic.json."""
import collections, csv, glob, json, os, shutil, sys
out, dst = 'gpurun_out/profr2', 'profiles/round2'
os.makedirs(dst, exist_ok=True)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
def stats(pattern, n=12):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        for i, r in enumerate(csv.DictReader(open(f))):
            if i < n: rows.append("  {Name:.110s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**r))
    return rows
def pmc(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{out}/{name}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
L = []
P = L.append
P(f"# profiles/round2/01_bench_command{tag}.txt -- MI355X, 1 GPU, round-2 build")
P("rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-side-legs --steps 30 --warmup 5")
P("== kernel stats (every launch of the command: the 400 ms settle phase + 5 warm-up + 30 timed steps)")
L += stats(out + '/trace/**/*kernel_stats.csv')
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "mono_fused" in r["Kernel_Name"]]
    if d:
        P(f"== kernel trace, mono_fused_kernel: launches={len(d)} avg_ns(all)={sum(d)/len(d):.0f}  avg_ns(last 30 = the timed region)={sum(d[-30:])/30:.0f}  min={min(d)}")
for name in ("bench_under_rocprof", "bench_default"):
    for l in open(f"{out}/{name}.log"):
        if l.startswith('{'):
            j = json.loads(l); r = j["roofline"]
            P(f"== {name}: value {j['value']} MS/s, ms_per_step {j['ms_per_step']}, HIP-event avg launch {r['avg_launch_ms']} ms over {r['launches_timed']} launches "
              f"(nothing subtracted), algorithmic {r['algorithmic_bytes_per_sample']} B/sample, achieved {r['achieved']} GB/s, frac {r['frac']}")
            if name == "bench_default":
                open(f"{dst}/03_bench_default_run{tag}.json", "w").write(l)
tr = {}
for c, name in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for k, v in pmc(name).items():
        if "mono_fused" in k and c in v:
            tr[c] = sum(v[c]) / len(v[c])
if len(tr) == 2:
    hbm = int(tr["FETCH_SIZE"] * 1024 * 2 + tr["WRITE_SIZE"] * 1024)
    alg = int(262144000 * (2 + 2 / 50))
    t = {"blocks": 256, "output": "s16", "round": "round 2", "kernel": "mono_fused_kernel<101,10,101,5>", "fetch_size_kb": tr["FETCH_SIZE"],
         "write_size_kb": tr["WRITE_SIZE"], "fetch_correction": 2.0, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch of `python3 bench.py --no-cpu-baseline --no-side-legs --steps 5 "
                 "--warmup 2 --settle-ms 0`; gfx950 FETCH_SIZE counts wide streaming reads at half (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact"}
    json.dump(t, open('profiles/fe_traffic.json', 'w'), indent=1)
    P(f"== HBM traffic per launch of mono_fused_kernel (separate --pmc passes): FETCH_SIZE {tr['FETCH_SIZE']:.1f} KB x2 (gfx950 correction) + WRITE_SIZE "
      f"{tr['WRITE_SIZE']:.1f} KB = {hbm} B; algorithmic (S3: 2.04 B/sample) {alg} B; ratio {hbm/alg:.3f}")
P("== SQ / LDS / MFMA counters of mono_fused_kernel, per dispatch (separate --pmc passes of the same command)")
for name in ("pmc_sq1", "pmc_sq2", "pmc_mfma"):
    for k, v in pmc(name).items():
        if "mono_fused" in k:
            for c, vals in sorted(v.items()):
                P(f"   {c:28s} per_dispatch={sum(vals)/len(vals):.6g}  (n={len(vals)})")
open(f"{dst}/01_bench_command{tag}.txt", "w").write("\n".join(L) + "\n")
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, f"{dst}/01_bench_command_kernel_stats{tag}.csv")
S = [f"# profiles/round2/02_stereo_step{tag}.txt -- MI355X, 1 GPU, round-2 build",
     "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/prof_stereo_r2.py",
     "(mode 0 stereo, 12 x 1,024,000-sample blocks per call, s16 L,R out, fresh stream: 1 first call with the serial PLL head + 4 streaming calls)"]
S += stats(out + '/stereo/**/*kernel_stats.csv', 14)
S += [l.rstrip() for l in open(out + "/stereo.log") if l.startswith("done")]
open(f"{dst}/02_stereo_step{tag}.txt", "w").write("\n".join(S) + "\n")
for f in glob.glob(out + '/stereo/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, f"{dst}/02_stereo_step_kernel_stats{tag}.csv")
print("\n".join(L)); print("\n".join(S))

M = [f"# profiles/round2/08_modes23{tag}.txt -- MI355X, 1 GPU: kernel stats of modes 2 and 3 mono, 127 M-sample steps (2 x the bench legs' step), s16 out",
     "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/prof_modes23_r2.py"]
M += stats(out + '/modes23/**/*kernel_stats.csv')
if len(M) > 2:
    open(f"{dst}/08_modes23{tag}.txt", "w").write("\n".join(M) + "\n")
