"""bench.py prints ONE JSON line with the contract's fields (task statement: metric/value/unit/n_gpus/steps/warmup/ms_per_step/
higher_is_better/scaling/vs_baseline/dtype/data/config + roofline + cpu_baseline).  Small batch so the check takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, steps=2):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--batch", "8", *extra],
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
    assert d["ranks_seen"] == 1 and d["backend"] == "none" and d["rccl_ranks_seen"] == 0 and d["exchange_ms_per_step"] is None
    k = d["clock"]                  # rocm-smi reading beside the timed region (None only where rocm-smi cannot read the device)
    assert k is None or (500 < k["sclk_mhz_under_load"] <= 2500 and k["socket_power_w"] > 100 and
                         abs(k["mfma_peak_at_this_clock_tflops"] - 2500.0 * k["sclk_mhz_under_load"] / 2400) < 0.1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert c["timed_iterations"] >= 3 and len(c["fixed_length"]["iter_s"]) >= 3 and c["host_hw_threads"] >= c["cores"] and c["cpu_model"]
    seg = c["fixed_length"]["segments_frac"]
    assert set(seg) == {"cnn", "transformer", "vit", "branch", "loss", "mix_and_glue"} and abs(sum(seg.values()) - 1) < 0.02
    assert c["c1_varlen_b16"]["pairs_per_s"] > 0
    # round 4: the line carries its own parity check (GPU vs the cpu_baseline leg's oracle outputs on the same pairs and weights) ...
    pc = d["parity_check"]
    assert pc["pairs"] == 2 and pc["loss_abs_diff"] <= 2e-2 and pc["audio_raw_cos_min"] > 0.9999 and "ok" in pc and set(pc["thresholds"]) >= {"centred_cos_min", "loss_abs_diff"}
    # ... every other BASELINE.json configuration as a short run beside the headline ...
    oc = d["other_configs"]
    for kind in ("cascaded_v8112", "large_b64", "varlen_packed", "varlen_padded", "train"):
        assert set(oc[kind]) >= {"ms_per_step", "pairs_per_s", "algorithmic_gflop_per_pair", "e2e_frac", "pairs_per_gpu"}, (kind, oc[kind])
        assert oc[kind]["pairs_per_s"] > 0 and 0 < oc[kind]["e2e_frac"] < 1
    assert oc["varlen_packed"]["algorithmic_gflop_per_pair"] < d["config"]["algorithmic_gflop_per_pair"] < oc["large_b64"]["algorithmic_gflop_per_pair"]
    # ... the launch method with the host's enqueue time, and energy next to the clock
    sl = d["step_launch"]
    assert sl["method"].startswith("eager") and sl["host_launch_ms_per_step"]["eager"] > 0
    assert k is None or (k["joules_per_step"] > 0 and abs(k["joules_per_pair"] * 8 - k["joules_per_step"]) < 0.02 * k["joules_per_step"] + 0.01)
    assert r["traffic_launches_per_step"] is None or r["traffic_launches_per_step"] == r["launches_per_step"]


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
    d = _run("--gpus", "2", "--share-gpu", "--graph", "on", "--cpu-pairs", "0", "--no-vendor-comparator")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["rccl_ranks_seen"] == 0 and d["devices_seen"] == 1 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["exchange_ms_per_step"] > 0 and d["value"] > 0 and "share-gpu" in d["data"]
    assert abs(d["loss"] - 2.77) < 0.3                                  # ~ ln(16): the loss saw the GLOBAL batch of 16 pairs
    # --graph on: the local part of the step replays from a captured HIP graph on the non-instrumented steps (the default is eager at every N)
    assert d["step_launch"]["method"].startswith("hip graph"), d["step_launch"]


def test_bench_eight_ranks_on_one_gpu_weak_and_strong():
    """BASELINE.json configs[3] launch shape on the one GPU this box has: `--gpus 8 --share-gpu` runs EIGHT ranks with the real
    kernels (each its own process, HIP graph replay, packed gather over gloo, loss on the global batch, max-over-ranks timing, one rank-0 line) --
    weak scaling (fixed pairs per rank) and strong scaling (`--global-batch`, split over the ranks)."""
    d = _run("--gpus", "8", "--share-gpu", "--graph", "on", "--cpu-pairs", "0", "--no-vendor-comparator", "--no-clock-probe", steps=4)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["backend"] == "gloo" and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp8" and d["scaling"] == "weak"
    assert abs(d["loss"] - 4.16) < 0.3                                  # ~ ln(64): the loss saw all 64 pairs
    assert d["step_launch"]["method"].startswith("hip graph") and d["step_launch"]["host_launch_ms_per_step"]["graph"] > 0
    s = _run("--gpus", "8", "--share-gpu", "--cpu-pairs", "0", "--no-vendor-comparator", "--no-clock-probe", "--global-batch", "64")
    assert s["scaling"] == "strong" and s["config"]["pairs_per_gpu"] == 8 and s["config"]["global_batch"] == 64 and s["ranks_seen"] == 8
    assert s["step_launch"]["method"].startswith("eager")
