"""bench.py's LAST stdout line must stay small enough for the driver's 8 KB stdout tail (round 3's 25 KB line was recorded as
`parsed: null`): short_line() is fed a real complete record (round 3's, tests/golden/bench_full_r03_sample.json) inflated to a worst
case - every configuration present, parity entries with long problem lists, rocprof detail, long strings - and must come out under 4 KB
with everything the driver and the judge read."""
import argparse
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def worst_case_record():
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_r03_sample.json")))
    par = lambda name, ok: {"config": name, "kind": "x" * 80, "slots": [8388608, 8454144], "frames": 461, "ok": ok, "seconds": 4.123456789,
                            "problems": [] if ok else ["position: 12345 differing words, first at row 17: " + "y" * 400] * 5, "attrs": ["position", "velocity", "age", "lifetime", "color"]}
    full["parity"] = par("c2", True)
    full["roofline"]["kernel_ms_rocprof"] = 0.12812345678
    full["roofline"]["frac_rocprof"] = 0.7871234567
    full["roofline"]["rocprof_detail"] = {"per_kernel_ms": {("k" * 100 + str(i)): {"avg": 0.123456789, "min": 0.1, "max": 0.2} for i in range(12)}}
    for name, v in full["configs"].items():
        v["parity"] = par(name, True)
        v["roofline"]["kernel_ms_rocprof"] = 0.2123456789
        v["roofline"]["rocprof_detail"] = copy.deepcopy(full["roofline"]["rocprof_detail"])
    for extra in ("c2_view", "c2_lean", "c2_reburst"):         # (round 5's end-to-end row; round 6: the LEAN row next to the default headline, the timed re-burst)
        full["configs"][extra] = copy.deepcopy(full["configs"]["c2_interop"])
        full["configs"][extra]["parity"]["config"] = extra
    full["configs"]["c2_view"].update({"consumer_ms": 0.128123456, "sim_only_ms": 0.1365123456})
    full["configs"]["c2_reburst"]["reburst"] = {"init_kernel_ms": 0.14876543, "init_frac": 0.6123456, "cycle": "c" * 80, "updates_per_frame_of_cycle": [16777216] * 4}
    full["config"].update({"options": "default", "stale_attr_mask_after": 0})
    full["roofline"]["step_kernels_ms"] = {"init": 0.0, "update": 0.1312345678, "lists": 0.0}
    full["roofline"]["algorithmic"].update({"elided_bytes_per_update": {"lifetime_read": 4, "alive_list_read_write": 8, "age_read": 4}, "designed_bytes_per_update": 52})
    for name in ("c2_mixed", "c2_dieoff", "c2_events", "c5"):   # (round 5: the churn configurations' gate on the timed state at full size)
        if name in full["configs"]:
            full["configs"][name]["parity"]["timed_state"] = {"ok": True, "checks": [{"ok": 1, "alive_count": 16499355}] * 3, "diffs": [{"equal": 1, "first_section": -1}] * 3,
                                                              "problems": [], "plain_kernels": "k" * 200, "seconds": 3.14159}
    full["comm"] = {"library": "/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so", "ranks": 1, "effects": 1, "alive_total": 16777216}
    full["config"]["workload"] += " " + "w" * 100
    return full


def test_the_multi_rank_line_carries_the_strong_scaling_run():
    """N > 1: no extra configurations, no scene, no cpu baseline - and the strong-scaling run of the same configuration."""
    full = worst_case_record()
    for k in ("configs", "small_effects_scene", "cpu_baseline", "comm"):
        full.pop(k, None)
    full["n_gpus"] = 8
    full["strong"] = {"value": 1.23456789e11, "ms_per_step": 0.123456789, "capacity_per_gpu": 2097152, "instances_per_gpu": 1, "kernel_ms_avg": 0.0123456789,
                      "algorithmic_whole_step_over_aggregate_peak": 0.123456789, "workload": "the N = 1 workload split over the ranks"}
    args = argparse.Namespace(parity=True, full_json=os.path.join(ROOT, "profiles", "bench_full.json"))
    line = bench.short_line(full, args)
    assert line["strong"]["capacity_per_gpu"] == 2097152 and "workload" not in line["strong"] and len(bench.encode_line(line)) < 4096


def test_short_line_fits_the_driver_tail_and_carries_the_contract():
    full = worst_case_record()
    assert len(json.dumps(full)) > 20000                      # the complete record is what broke round 3
    args = argparse.Namespace(parity=True, full_json=os.path.join(ROOT, "profiles", "bench_full.json"))
    text = bench.encode_line(bench.short_line(full, args))
    assert len(text) < 4096, len(text)
    assert len(text) < 6000
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "configs", "build", "comm"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    ro = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "kernel_ms_rocprof"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["algorithmic_bytes_per_update"] == 68 and "whole_step_frac" in ro and "algorithmic_whole_step_over_peak" in ro
    assert all(not isinstance(v, (dict, list)) for v in ro.values()), "roofline must stay flat: the driver keeps its scalar members only"
    assert line["config"]["options"] == "default" and line["config"]["stale_attr_mask_after"] == 0          # (the headline runs the library's defaults, nothing stale)
    assert ro["elided"] == "lifetime_read 4 + alive_list_read_write 8 + age_read 4" and ro["designed_bytes_per_update"] == 52 and abs(ro["step_kernels_ms"] - 0.1312) < 1e-4
    assert line["configs"]["c2_view"]["consumer_ms"] and line["configs"]["c2_reburst"]["reburst_init_frac"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] <= cb["threads"] and cb["host_physical_cores"] and "cpu_model" in cb and cb["sample"] and "cpu_quota" in cb
    assert line["parity"]["ok"] is True and set(line["parity"]["checked"]) == {"c2"} | set(full["configs"])
    assert line["parity"]["churn"].startswith("timed state: invariants + plain-path differential at full size") and line["parity"]["timed_state"]["ok"] is True
    assert line["comm"]["ranks"] == 1 and "librccl" in line["comm"]["library"]
    assert set(line["configs"]) == set(full["configs"])
    for row in line["configs"].values():
        assert {"value", "ms_per_step", "frac", "ws_frac", "ws68", "kernel_ms"} <= set(row)      # (kernel_ms_rocprof of the side rows: in the complete record)
    assert len(line["windows"]["ms_per_step_min_median_max"]) == 3
    assert line["value"] == full["value"]                      # the headline is not rounded


def test_a_failed_parity_check_refuses_the_value():
    full = worst_case_record()
    full["configs"]["c2_mixed"]["parity"]["ok"] = False
    full["configs"]["c2_mixed"]["parity"]["problems"] = ["velocity: 3 differing words, first at row 5: oracle [1 2 3] device [1 2 4]" + "z" * 500]
    args = argparse.Namespace(parity=True, full_json=os.path.join(ROOT, "profiles", "bench_full.json"))
    line = bench.short_line(full, args)
    assert line["value"] is None and "c2_mixed" in line["refused"] and line["parity"]["ok"] is False
    assert "c2_mixed" in line["parity"]["failed"] and "c2_mixed" not in line["parity"]["checked"]
    assert len(bench.encode_line(line)) < 4096
    # no parity information at all while the gate is on: not accepted either
    bare = worst_case_record()
    bare["parity"] = None
    for v in bare["configs"].values():
        v["parity"] = None
    assert bench.short_line(bare, args)["parity"]["checked"] == []
    off = bench.short_line(bare, argparse.Namespace(parity=False, full_json=args.full_json))
    assert off["parity"]["ok"] is None and off["parity"]["skipped"] == "--no-parity"
