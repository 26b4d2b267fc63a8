"""bench.py prints one JSON line with the fields the harness reads (short run, small CPU sample)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_the_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2",
                          "--cpu-images", "1", "--seeds", "2"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 32 * 8 / (d["ms_per_step"] * 8 * 1e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port" and c["cores"] == 1
    assert d["parity"] == {"images": 1, "match_mask_exact": 1, "nms_kept_exact": 1, "floats_close": 1,
                           "ytrue_equals_conf_locs": 1}
    # no per-kernel figure above its own roofline; latency-bound kernels carry no fraction
    for k, v in d["roofline_kernels"].items():
        assert v["frac"] is None or 0.0 < v["frac"] <= 1.0, (k, v)
    assert d["timed_steps"] >= 2000 and d["value_requested_region"] > 0
    assert d["config"]["input_sets"] * 24 * d["config"]["anchors"] * 32 > 256 * 2 ** 20
