#!/usr/bin/env python3
"""Turn gpurun_out/profb (output of tools/gpu_profile_bench.sh) + gpurun_out/bench_default_final.log into the
committed artifacts: profiles/round1/09_*, 12_*, profiles/fe_traffic.json."""
import csv, glob, json, os, shutil
out = 'gpurun_out/profb'
t = json.load(open('gpurun_out/fe_traffic.json'))
t["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch of `python3 bench.py --steps 5 --warmup 2 "
             "--settle-ms 0`; gfx950 FETCH_SIZE counts wide streaming reads at half (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact")
t["algorithmic_bytes_per_launch"] = int(262144000 * (2 + 6 / 50))
json.dump(t, open('profiles/fe_traffic.json', 'w'), indent=1)
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, 'profiles/round1/09_bench_command_fused_kernel_stats.csv')
with open('profiles/round1/09_bench_command_fused_stats_and_traffic.txt', 'w') as o:
    def P(*a): print(*a, file=o)
    P("rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline   (MI355X, 1 GPU, final round-1 build)")
    P("== kernel stats (all launches of the command: 400 ms settle phase + 5 warm-up + 30 timed steps of the fused kernel; then the")
    P("   north-star leg: 200 + 30 steps of fe_demod_kernel + audio_fir_kernel; then the copy-bandwidth probe)")
    for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
        for i, r in enumerate(csv.DictReader(open(f))):
            if i < 8: P("  {Name:.100s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**r))
    for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "mono_fused" in r["Kernel_Name"]]
        P(f"== kernel trace, mono_fused_kernel: launches={len(d)} avg_ns(all)={sum(d)/len(d):.0f}  avg_ns(last 30 = the timed region)={sum(d[-30:])/30:.0f}")
        P("   first 10 launches after idle (ns):", d[:10])
    for name in ("bench_under_rocprof", "bench_plain"):
        for l in open(f"{out}/{name}.log"):
            if l.startswith('{'):
                j = json.loads(l); r = j["roofline"]
                P(f"== {name}: value {j['value']} MS/s, ms_per_step {j['ms_per_step']}, HIP-event avg launch {r['avg_launch_ms']} ms (event-pair overhead "
                  f"{r['event_pair_overhead_ms']} ms subtracted) over {r['launches_timed']} sampled launches, achieved {r['achieved']} GB/s, frac {r['frac']}; "
                  f"measured copy {r['measured_copy']}; north_star_form {j.get('north_star_form')}")
    P("== HBM traffic per launch (separate --pmc passes, `bench.py --steps 5 --warmup 2 --settle-ms 0`):")
    P(f"   FETCH_SIZE {t['fetch_size_kb']:.1f} KB x2 (gfx950 correction) + WRITE_SIZE {t['write_size_kb']:.1f} KB = {t['hbm_bytes_per_launch']} B; "
      f"algorithmic {t['algorithmic_bytes_per_launch']} B; ratio {t['hbm_bytes_per_launch']/t['algorithmic_bytes_per_launch']:.3f}")
if os.path.exists('gpurun_out/bench_default_final.log'):
    line = [l for l in open('gpurun_out/bench_default_final.log') if l.startswith('{')][-1]
    open('profiles/round1/12_bench_default_run_final.json', 'w').write(line)
print(open('profiles/round1/09_bench_command_fused_stats_and_traffic.txt').read()[:3000])
