#!/bin/bash
# The committed profile of the bench command itself: kernel trace + stats of
# `python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline`, then FETCH/WRITE PMC passes.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/profb
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o pmc -- python3 bench.py --steps 5 --warmup 2 --settle-ms 0 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_plain.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = 'gpurun_out/profb'
print("== kernel stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline)")
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8: print("  {Name:.80s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**r))
tr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{out}/pmc_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "mono_fused" in k or "fe_mfma" in k or "audio_fir" in k:
            print(f"  {c} {k}: per dispatch {sum(v)/len(v):.1f} KB (n={len(v)})")
            tr[(c, "fe" if "mono_fused" in k else k)] = sum(v) / len(v)
# the --stats average covers every launch of the command, i.e. also the settle phase in front of the
# timed region (first launches after idle run slower); the timed region = the last 30 launches of the trace
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "mono_fused" in r["Kernel_Name"]]
    if d:
        print(f"  kernel trace: mono_fused_kernel launches={len(d)} avg_ns(all)={sum(d)/len(d):.0f} avg_ns(last 30 = timed region)={sum(d[-30:])/30:.0f}")
for l in open(out + '/bench_under_rocprof.log'):
    if l.startswith('{'): d = json.loads(l); print("bench under rocprof: fe avg_launch_ms", d['roofline']['avg_launch_ms'], "value", d['value'])
for l in open(out + '/bench_plain.log'):
    if l.startswith('{'): d = json.loads(l); print("bench plain:         fe avg_launch_ms", d['roofline']['avg_launch_ms'], "value", d['value'])
if ("FETCH_SIZE", "fe") in tr and ("WRITE_SIZE", "fe") in tr:
    # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide streaming reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact
    hbm = tr[("FETCH_SIZE", "fe")] * 1024 * 2 + tr[("WRITE_SIZE", "fe")] * 1024
    json.dump({"blocks": 256, "kernel": "mono_fused_kernel<101,10,101,5>", "fetch_size_kb": tr[("FETCH_SIZE", "fe")],
               "write_size_kb": tr[("WRITE_SIZE", "fe")], "fetch_correction": 2.0, "hbm_bytes_per_launch": int(hbm)},
              open('gpurun_out/fe_traffic.json', 'w'))
    print("traffic per launch:", int(hbm), "bytes")
PY
