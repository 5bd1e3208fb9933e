"""Contracts that can be checked without a GPU: the committed bench line carries every field the
driver's bench contract names, the level-tensor cache behaves, profiles exist."""
import json
import os

import pytest
import torch

from conftest import ROOT


def test_committed_bench_line_follows_the_contract():
    path = os.path.join(ROOT, "profiles", "r01_bench_line.json")
    line = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "Gpoints/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # value = points per step / time per step
    assert abs(line["value"] - line["config"]["points_per_step"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3) / 1e9) < 1e-6
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference")


def test_profiles_are_committed():
    for name in ("r01_bench_kernel_stats.csv", "r01_bench_pmc_hbm.json", "r01_bench_line.json"):
        assert os.path.getsize(os.path.join(ROOT, "profiles", name)) > 0, name
    stats = open(os.path.join(ROOT, "profiles", "r01_bench_kernel_stats.csv")).read()
    for kernel in ("msda_fwd_d32_kernel", "msda_bwd_d32_kernel", "msda_bwd_gv"):
        assert kernel in stats, kernel


def test_level_tensors_are_cached_tagged_and_checked_on_the_host():
    from vnext_amd.ops.functions import check_flattened_length, level_tensors
    a = level_tensors([(6, 8), (3, 4)], "cpu")
    b = level_tensors(((6, 8), (3, 4)), "cpu")
    assert a[0] is b[0] and a[1] is b[1]                       # one pair per (shapes, device)
    assert a[0].tolist() == [[6, 8], [3, 4]] and a[1].tolist() == [0, 48]
    assert a[0]._vnx_hw == ((6, 8), (3, 4)) and a[1]._vnx_levels_packed
    check_flattened_length(a[0], 60)
    with pytest.raises(AssertionError):
        check_flattened_length(a[0], 61)
    check_flattened_length(torch.tensor([[6, 8], [3, 4]]), 60)  # untagged: the reference's device-side form
