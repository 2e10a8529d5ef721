#!/bin/bash
# round 3 acceptance pass + the committed profiles, one GPU call, final build:
#   smoke, the whole -m gpu suite, the bench command under rocprofv3 (kernel trace + stats; FETCH / WRITE PMC passes),
#   the default bench line WITH the traffic figure of this build, a 2-rank rehearsal of `bench.py --gpus 2` on one device,
#   kernel stats + one call's timeline of the exact and the fast receiver bank.
# Output: gpurun_out/final3/ -> copy to profiles/round3/ and profiles/fe_traffic.json (tools/README.md).
set -u
export TMPDIR=/tmp
F=gpurun_out/final3
rm -rf $F; mkdir -p $F
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "$F/$name.log" 2>&1 < /dev/null; local rc=$?; echo "== $name rc=$rc"; tail -n 4 "$F/$name.log" | cut -c1-400; if [ $rc -ge 124 ]; then exit $rc; fi; }
step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke()"
step pytest_gpu 1000 python3 -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider
# the bench command under the profiler; counters in passes of their own
bash tools/gpu_profile_r3.sh > $F/profile_r3.log 2>&1; echo "== profile_r3 rc=$?"; tail -n 8 $F/profile_r3.log | cut -c1-300
cp gpurun_out/profr3/00_bench_command.txt gpurun_out/profr3/00_bench_command_kernel_stats.csv gpurun_out/profr3/fe_traffic.json $F/ 2>/dev/null
# the default run once more with this build's traffic file in place (bench.py reports traffic only for the build it was collected on)
if [ -f gpurun_out/profr3/fe_traffic.json ]; then cp gpurun_out/profr3/fe_traffic.json profiles/fe_traffic.json; fi
step bench_default 600 python3 bench.py
grep '^{' $F/bench_default.log | tail -n 1 > $F/03_bench_default_run.json
step bench_two_ranks 300 python3 bench.py --gpus 2 --all-on-device0 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs
bank() { local out=$1 head=$2; shift 2; { echo "$head"; echo "# rocprofv3 --kernel-trace --stats -- python3 tools/bank_bench.py $*   (tools/gpu_bank_prof.sh)"; bash tools/gpu_bank_prof.sh "$@"; } > $F/$out.txt 2>&1
         cp "$(find gpurun_out/prof_bank -name '*kernel_stats.csv' | head -n 1)" $F/${out}_kernel_stats.csv 2>/dev/null; echo "== $out"; grep channels $F/$out.txt | head -n 2; }
bank 01_bank_exact_16384ch "# receiver bank, exact (fmrx_channels_create_ex audio_channels = 2, exact = 1), mode 0, 16384 receivers x 204800 samples (4 reference blocks) per call: kernel stats over 5 calls (8 chunks each), then the timeline of the last call (q = hardware queue)" --exact 1 --channels 16384 --blocks-per-call 4 --calls 3
bank 01_bank_exact_65536ch "# the same bank, 65536 receivers x 51200 samples (1 reference block) per call: 2 chunks" --exact 1 --channels 65536 --blocks-per-call 1 --calls 3
bank 02_bank_fast_16384ch "# receiver bank, fast (exact = 0), mode 0, 16384 receivers x 204800 samples per call; front end, band-pass pair + output stage, PLL lanes on three internal streams" --exact 0 --channels 16384 --blocks-per-call 4 --calls 3
ls -la $F
