"""bench.py's one-line JSON contract: on a box without a GPU it must say so and exit non-zero
(libfmrx has no CPU fallback); on the GPU a tiny run must carry every field the driver reads plus
the `roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_without_gpu_fails_loudly(fmrx):
    if fmrx.device_count() > 0:
        pytest.skip("a GPU is present")
    rc, out, err = _run("--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert rc != 0 and out is not None and "error" in out and "no GPU" in out["error"]


@pytest.mark.gpu
def test_bench_json_contract(fmrx):
    rc, out, err = _run("--steps", "8", "--warmup", "2", "--blocks", "64", "--settle-ms", "50", "--cpu-seconds", "0.5")
    assert rc == 0, err
    for k, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(out[k], typ), (k, out.get(k))
    assert out["vs_baseline"] is None and out["scaling"] == "weak" and out["n_gpus"] == 1 and out["steps"] == 8
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["avg_launch_ms"] > 0 and r["launches_timed"] == 8
    assert r["algorithmic_bytes_per_sample"] == 2.04 and (r["traffic"] is None or "NOT measured in this run" in r["traffic_source"])
    assert r["avg_launch_ms"] <= out["ms_per_step"] * 1.02          # the kernel cannot take longer than the step it is in
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1 and c["unit"] == "MS/s" and c["sample"]
    # value is whole-job throughput of exactly `steps` steps: consistent with ms_per_step
    assert abs(out["value"] - out["config"]["samples_per_step_per_gpu"] / (out["ms_per_step"] * 1e-3) / 1e6) / out["value"] < 0.01
    assert out["north_star_form"]["fe_variant"] == "valu" and out["north_star_form"]["value"] > 0
    for k in ("two_kernel_path", "s1_if_only_mfma", "s1_if_only_valu", "mode1_mono", "mode2_mono", "mode3_mono", "mode0_stereo"):
        g = out["legs"][k]
        assert g["value"] > 0 and 0 < g["frac"] < 1 and g["algorithmic_bytes_per_sample"] >= 2.0, (k, g)
    assert out["legs"]["small_block"]["us_per_block"] > 0
    assert out["config"]["fmrx_env"] == []
    assert "two_thread_pipeline" in c
    # every stereo number carries the bound it meets; the conforming (bit-exact) paths are measured too
    for k, g in out["legs"].items():
        if "stereo" in k and "error" not in g:
            assert "tolerance" in g, k
    assert out["legs"]["mode0_stereo_exact"]["tolerance"].startswith("bit-exact")
    bank = out["legs"]["stereo_channels_exact"]
    assert bank["tolerance"].startswith("bit-exact") and bank["value"] > 1e4 and bank["channels"] >= 256
    assert out["legs"]["stereo_channels"]["value"] > bank["value"] and "ulp(trigArg" in out["legs"]["stereo_channels"]["tolerance"]
    assert out["legs"]["stereo_channels_exact_mode2"]["tolerance"].startswith("bit-exact") and out["legs"]["stereo_channels_exact_mode2"]["value"] > 1e4
    assert out["legs"]["mono_channels_mode2"]["value"] > 1e4 and "2e-6" in out["legs"]["mono_channels_mode2"]["tolerance"]
    # the CPU figure beside every mode's leg
    for k in ("mode1_mono", "mode2_mono", "mode3_mono", "mode0_stereo", "mode2_stereo"):
        assert c["legs"][k]["value"] > 0.5 and c["legs"][k]["cores"] == 1
    # the line says which library was timed
    lib = out["config"]["library"]
    assert len(lib["sha256"]) == 64 and "src:" in lib["version"] and lib["path"].endswith("libfmrx.so")
    assert "mono_256_blocks" not in out["legs"] or out["legs"]["mono_256_blocks"].get("frac", 1) > 0.2   # (the test runs with --blocks 64)
    hb = out["legs"]["host_buffers"]
    assert "error" not in hb and hb["h2d_GBs"] > 5 and hb["value"] > 1000, hb
    assert out["legs"]["cli_stdin_stdout"]["rc"] == 0 and out["legs"]["cli_stdin_stdout"]["output_bytes"] * 50 == out["legs"]["cli_stdin_stdout"]["input_bytes"]
