"""bench.py's output contract (one JSON line: metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config / roofline /
cpu_baseline), on a short run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--prewarm", "0", "--batch", "4", "--cpu-seconds", "1", "--no-e2e", "--no-pack",
                        "--no-other", "--no-gpu-entropy"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "Mpixel/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["bit_exact_vs_oracle"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / measured launch time
    want = rf["algorithmic_bytes_per_launch"] / rf["kernel_ms_per_launch"] / 1e6
    assert abs(rf["achieved"] - want) < 0.005 * want          # (the JSON rounds the milliseconds)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 0 and d["ms_per_step"] > 0


@pytest.mark.gpu
def test_graft_entry_smoke_runs(gpu):
    """__graft_entry__.smoke(): one small decode on cuda:0 checked against the oracle."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    __graft_entry__.smoke()


def test_graft_entry_build_is_idempotent():
    """build() with everything already built returns quickly and leaves the library loadable."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    __graft_entry__.build()
    assert os.path.exists(os.path.join(ROOT, "jpeg_gpu_amd", "libjpeg_gpu_amd.so"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))
