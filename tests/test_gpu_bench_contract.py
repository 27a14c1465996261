"""GPU: the one-line JSON contract of bench.py (what the driver parses): keys, types, the roofline and cpu_baseline
objects, and internal consistency (value = steps / time, frac = achieved / peak)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "questions/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]          # batch 1: one question per step
    assert 0.5 < d["value"] < 10.0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["bytes_per_launch"] < 1.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "threads", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == "questions/s" and c["cores"] >= 1
    assert c["cores"] == os.cpu_count() and (c["threads"] is None or 1 <= c["threads"] <= c["cores"])
    assert c["value"] is None or 0 < c["value"] < d["value"]
    assert d["tile_upload_ms"] > 0 and d["weight_broadcast_s"] == 0
    # BASELINE configs[2] in the same line: one timed step of 64 questions about 6 tiles through the scheduler
    b = d["batch64"]
    assert b["questions"] == 256 and b["chain_slots"] == 64 and b["tiles"] == 6 and b["steps"] == 4
    assert abs(b["value"] - 64000.0 / b["ms_per_step"]) < 1e-6 * b["value"]
    assert b["value"] > 5 * d["value"]
    assert 0.75 * 192 <= b["mean_N1"] <= 1.25 * 192 and 0.75 * 96 <= b["mean_N2"] <= 1.25 * 96 and b["mean_L1"] == 802
    assert b["scheduler"]["admitted"] == 512 and b["scheduler"]["chain_steps"] > 40 * b["scheduler"]["steps"]
    rb = b["roofline"]
    assert rb["bound"] == "hbm" and rb["chains"] == 64 and "batched decode" in rb["kernel"]
    assert abs(rb["frac"] - rb["achieved"] / rb["peak"]) < 1e-9 and 0.05 < rb["frac"] < 1.0
    # BASELINE configs[3], one GPU's share of the stream: 1024 questions through 256 chain slots
    w = d["stream256"]
    assert w["questions"] == 1024 and w["chain_slots"] == 256 and w["scheduler"]["admitted"] == 2048
    assert abs(w["value"] - 1024 / w["seconds"]) < 1e-2 * w["value"] and w["value"] > b["value"]
    assert 64 < w["mean_chains_per_step"] <= 256
    ph = d["roofline_phases"]
    assert ph["decode"]["bound"] == "hbm" and ph["vit"]["bound"] == "mfma" and 0 < ph["question"]["frac"] < 1
