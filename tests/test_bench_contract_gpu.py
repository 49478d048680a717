"""bench.py prints ONE JSON line with the contract's fields (task statement: metric/value/unit/n_gpus/steps/warmup/ms_per_step/
higher_is_better/scaling/vs_baseline/dtype/data/config + roofline + cpu_baseline).  Small batch so the check takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract_forward():
    d = _run("--cpu-pairs", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3          # value = pairs / time of the timed region
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_per_step"] > 0 and r["avg_launch_ms"] > 0 and "traffic" in r
    assert r["vendor_plain_gemms"]["launches_per_step"] == 0         # every GEMM of the headline step is the hand-written kernel
    v = d["vendor_comparator"]
    assert v["value"] > 0 and v["unit"] == "pairs/s" and v["ms_per_step"] > 0
    assert d["rccl_ranks_seen"] == 1 and d["exchange_ms_per_step"] is None
    k = d["clock"]                  # rocm-smi reading beside the timed region (None only where rocm-smi cannot read the device)
    assert k is None or (500 < k["sclk_mhz_under_load"] <= 2500 and k["socket_power_w"] > 100 and
                         abs(k["mfma_peak_at_this_clock_tflops"] - 2500.0 * k["sclk_mhz_under_load"] / 2400) < 0.1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert c["timed_iterations"] >= 3 and len(c["fixed_length"]["iter_s"]) >= 3 and c["host_hw_threads"] >= c["cores"] and c["cpu_model"]
    seg = c["fixed_length"]["segments_frac"]
    assert set(seg) == {"cnn", "transformer", "vit", "branch", "loss", "mix_and_glue"} and abs(sum(seg.values()) - 1) < 0.02
    assert c["c1_varlen_b16"]["pairs_per_s"] > 0


def test_bench_json_contract_train_and_cascaded():
    t = _run("--cpu-pairs", "0", "--train")
    assert t["config"]["mode"].startswith("train") and t["cpu_baseline"] is None and t["value"] > 0
    c = _run("--cpu-pairs", "0", "--model", "cascaded")
    assert "Cascaded" in c["metric"] and c["value"] > 0
    c = _run("--cpu-pairs", "0", "--model", "cascaded", "--vocab", "49408", "--no-vendor-comparator")
    assert c["value"] > 0 and c["vendor_comparator"] is None


def test_bench_two_ranks_on_one_gpu_exercises_the_multi_rank_path():
    """`bench.py --gpus 2` self-launches two ranks; with the --share-gpu test hook both run the REAL kernels on cuda:0 and exchange over
    gloo, so the N > 1 branch (packed gather, loss on the global batch, exchange timing, max-over-ranks, rank-0 line) runs on hardware
    before the driver's multi-GPU node does it over RCCL."""
    d = _run("--gpus", "2", "--share-gpu", "--cpu-pairs", "0", "--no-vendor-comparator")
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["exchange_ms_per_step"] > 0 and d["value"] > 0 and "share-gpu" in d["data"]
    assert abs(d["loss"] - 2.77) < 0.3                                  # ~ ln(16): the loss saw the GLOBAL batch of 16 pairs
