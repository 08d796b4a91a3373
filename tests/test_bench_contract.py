"""The bench contract, checked on the committed lines (profiles/r04d_bench.json, profiles/r04d_train_bench.json, r04d_train_e4m3_bench.json): every
field the driver reads is there, the numbers are consistent with each other, and bench.py's own helpers (argument
parsing, PMC lookup by full kernel instantiation) behave -- no GPU needed."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def load(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


@pytest.fixture(scope="module")
def bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_render_line():
    d = load("r04d_bench.json")
    for k in REQUIRED + ("cpu_baseline", "aux"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "ray-samples/s" and d["data"] == "synthetic" and d["dtype"] in ("f16", "bf16", "f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    samples = d["config"]["rays"] * d["config"]["samples_per_ray"]
    assert samples == 800 * 800 * 128
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert abs(r["achieved"] - r["samples_per_launch"] * r["flop_per_sample"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_hbm_bytes"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and abs(c["psnr_delta_vs_teacher_db"]) <= 0.05
    aux = d["aux"]
    assert set(aux) == {"c2", "c4", "c5", "shard8"} and set(aux["c5"]) == {"N64", "N128"}
    for k in ("c2", "c4"):
        assert aux[k]["ms"] > 0 and 0 < aux[k]["frac_of_peak"] < 1 and aux[k]["kernel"]
    # both 16-bit operand types, timed alike, each with the PSNR criterion on the timed weights
    by = d["by_dtype"]
    assert set(by) == {"f16", "bf16"} and by[d["dtype"]].get("headline") is True
    assert by[d["dtype"]]["value"] == d["value"] and by[d["dtype"]]["kernel_ms"] == d["roofline"]["kernel_ms"]
    for k, v in by.items():
        assert v["steps"] == d["steps"] and v["warmup"] == d["warmup"] and 0.4 < v["frac"] < 1 and v["kernel_ms"] <= v["ms_per_step"] * 1.001
        assert v["meets_0.05_db"] == (abs(v["psnr_delta_vs_teacher_db"]) <= 0.05)
    assert by["f16"]["meets_0.05_db"] is True
    # the training iteration selects its batch from the 16 M-ray table inside the timed step
    for n in ("N64", "N128"):
        assert aux["c5"][n]["table_rays"] == 16_000_000 and "select_scan_kernel" in aux["c5"][n]["kernel"]
        # ... and in the 8-bit storage form of the saved tensors, timed alike: faster, and the same training
        e8 = aux["c5"][n]["storage_e4m3"]
        assert e8["steps"] == aux["c5"][n]["steps"] and 0.6 * aux["c5"][n]["ms"] < e8["ms"] < 0.95 * aux["c5"][n]["ms"]
        assert abs(e8["final_loss"] - aux["c5"][n]["final_loss"]) <= 0.02 * aux["c5"][n]["final_loss"]
    # rank 0 of 8 on the wall clock: kernel + collective + host gap = the step; eight such steps no slower than 1.33 full images
    s8 = aux["shard8"]
    assert "error" not in s8
    assert abs(s8["step_ms"] - (s8["kernel_ms"] + s8["collective_ms"] + s8["host_gap_ms"])) < 1e-9
    assert 8 * s8["step_ms"] <= 1.33 * s8["full_image_step_ms"] and s8["host_gap_ms"] < 0.1 * s8["step_ms"]
    assert s8["train"]["step_ms_all_reduce"] >= s8["train"]["step_ms_no_exchange"] * 0.98


@pytest.mark.parametrize("name,storage", [("r04d_train_bench.json", "bf16"), ("r04d_train_e4m3_bench.json", "e4m3")])
def test_train_line(name, storage):
    d = load(name)
    for k in REQUIRED:
        assert k in d, k
    assert d["scaling"] == "weak" and d["dtype"] == "bf16"
    P = d["config"]["rays_per_gpu"] * d["config"]["samples_per_ray"]
    assert abs(d["value"] - d["n_gpus"] * P / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["config"]["table_rays"] == 16_000_000 and "rg.select" in d["config"]["batch"]
    # the storage form of the saved tensors is named, its kernel is the one timed, and both forms are timed alike
    assert ("e4m3" in d["config"]["storage"]) == (storage == "e4m3")
    assert r["kernel"].startswith("dw_gemm_e4m3_kernel" if storage == "e4m3" else "dw_gemm_kernel")
    assert r["algorithmic_bytes_per_point"] == (5856 if storage == "e4m3" else 11776)
    assert r["traffic"] is None or 0.9 * r["algorithmic_bytes_per_point"] * P <= r["traffic"] <= 1.1 * r["algorithmic_bytes_per_point"] * P
    by = d["by_storage"]
    assert by[storage]["headline"] is True and by[storage]["ms_per_step"] == d["ms_per_step"]
    assert by["e4m3"]["steps"] == by["bf16"]["steps"] and by["e4m3"]["ms_per_step"] < 0.9 * by["bf16"]["ms_per_step"]
    assert abs(by["e4m3"]["final_loss"] - by["bf16"]["final_loss"]) <= 0.02 * by["bf16"]["final_loss"]


def test_pmc_lookup_needs_the_exact_instantiation(bench):
    """roofline.traffic comes from a committed PMC summary only for the same mode / precision AND the full template
    instantiation: the training forward of the bf16 kernel must never stand in for the bf16 render."""
    t16, src16 = bench.pmc_traffic(bench.RENDER_KERNEL["fp16"], "render", "fp16")
    tb, srcb = bench.pmc_traffic(bench.RENDER_KERNEL["bf16"], "render", "bf16")
    assert t16 and tb and src16 != srcb and 3.0e7 < t16 < 5.0e7 and 3.0e7 < tb < 5.0e7
    wrong = bench.pmc_traffic(bench.RENDER_KERNEL["bf16"], "render", "fp16")       # an r02 summary without arguments may hold it
    assert wrong == (None, None) or wrong[0] < 5.0e7
    assert bench.pmc_traffic("nerf_mlp_bf16_16_kernel<true, true, true>", "render", "bf16") == (None, None)
    assert bench.pmc_traffic(bench.RENDER_KERNEL["fp32"], "render", "fp32") == (None, None)
    tt, _ = bench.pmc_traffic("dw_gemm_kernel(", "train")
    assert tt and tt > 2.5e9
    t8, src8 = bench.pmc_traffic("dw_gemm_e4m3_kernel(", "train")
    assert t8 and 1.3e9 < t8 < 1.7e9 and "e4m3" in src8
