"""The contract line of bench.py: the driver reads the LAST stdout line as JSON and gave up on round 5's 21 KB line.  The line is built by
bench.compact_line from the run's full record; here it is built from a record with every leg present and over-long strings everywhere, and
must stay under bench.LINE_LIMIT bytes with every contract key in place.  (No GPU: pure host logic.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_ray", "valu_issue_frac", "lane_utilisation", "primary")


def full_record(pad=1):
    n = 1 << 24
    pm = {"primary": {"FETCH_SIZE": 9.6e5, "WRITE_SIZE": 3.3e5, "SQ_INSTS_VALU": 2.1e9, "SQ_ACTIVE_INST_VALU": 1.0e9, "SQ_THREAD_CYCLES_VALU": 5.8e10, "SQ_BUSY_CYCLES": 1e9},
          "diffuse": {"FETCH_SIZE": 9.9e6, "WRITE_SIZE": 3.3e5, "SQ_INSTS_VALU": 4.3e9, "SQ_ACTIVE_INST_VALU": 2.0e9, "SQ_THREAD_CYCLES_VALU": 7.8e10, "SQ_BUSY_CYCLES": 2e9},
          "source": "live " * 40 * pad}
    mean = {"primary": 2.301, "diffuse": 4.556, "shadow": 2.2}
    roof = bench.roofline_lines(10, n, mean, {"primary": (26.16, 4.07), "diffuse": (25.65, 5.46)}, pm, {"read_gbps": 6224.0, "copy_gbps": 5000.0, "valu_ginstr_per_s": 1180.0, "source": "x" * 300 * pad})
    vs = {k: {"n": 65536, "hits": 58000, "hitmiss": 0, "prim_real": 0, "t_bad": 0, "uv_differs": 0, "farther_by_ulps": 0, "differ_from_reference": i} for i, k in enumerate(("primary", "diffuse"))}
    detail = {"primary_mrays": 7291.123456, "diffuse_mrays": 3682.123456, "shadow_mrays": 7600.5, "primary_plus_diffuse_kernel_mrays": 4890.9, "kernel_ms": mean, "dispatch_gap_ms": 0.04,
              "coherent_schedule": {"closest_hit": {"decision": "wave packet", "samples": [4, 3]}, "any_hit": {"decision": "deferred+gated", "samples": [3, 3]}, "how": "h" * 500 * pad},
              "config4_strong": {"rays": 1 << 26, "mrays": 3951.85, "workload": "w" * 400 * pad},
              "per_gpu": [{"rank": i, "primary_kernel_ms": 2.3, "diffuse_kernel_ms": 4.5} for i in range(8)],
              "reference_blob": {"kind": "reference", "primary_mrays": 4824.9, "diffuse_mrays": 3549.0, "primary_plus_diffuse_mrays": 4089.7, "steps": 20, "coherent_schedule": "strict",
                                 "vs_real_reference": dict(vs, differ_from_reference=1, rule="r" * 300 * pad)},
              "parity_sample": {"n": 65536, "hitmiss": 0, "prim_real": 0, "t_bad": 0, "uv_bad": 0, "tie": 0, "onsurf": 2, "not_bit_identical": 0, "shadow_flags_differ": 0,
                                "vs_real_reference": dict(vs, differ_from_reference=1), "ok": True, "rule": "p" * 300 * pad},
              "config2": {"rays": 1 << 20, "bvh_gpu_mrays": 4014.1, "ref_opencl_mrays": 2890.5},
              "tlas_1000_instances": {"camera_mrays": 5400.0, "device_tlas_rebuild_ms": 0.2, "device_blas_refit_ms": 0.1, "vs_reference_opencl": {"x": "y" * 2000 * pad}},
              "other_layouts": {"BVH_GPU": {"note": "z" * 5000 * pad}}, "hbm_regime": {"scenes": {"a": "b" * 5000 * pad}}}
    return {"metric": bench.METRIC, "value": 4860.123456789, "unit": "MRays/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 6.9041234, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "procedural street (Bistro-exterior stand-in, 2.83M tris, seed 2); BVH8_CWBVH; per GPU per step 16777216 primary + 16777216 diffuse (depth 1-3) Intersect" + " and more" * 50 * pad,
                       "scene_tris": 2832120, "layout": "BVH8_CWBVH", "rays_per_gpu_per_step": 2 * n, "sharding": "s" * 400 * pad},
            "parity_checked": True, "parity_ok": True, "detail": detail, "roofline": roof,
            "cpu_baseline": {"value": 39.7, "unit": "MRays/s", "cores": 16, "kind": "reference", "sample": "tinybvh BVH8_CPU::Intersect (AVX2), 16 threads, 8388608 primary + 8388608 diffuse rays of the GPU batches" * pad,
                             "threads_1": {"value": 2.58}, "bvh_intersect": {"value": 10.1, "sample": "q" * 300 * pad}},
            "legs_s": [("leg_%d" % i, 1.25) for i in range(12)]}


def test_line_is_small_and_complete():
    for pad in (1, 20):
        line = bench.compact_line(full_record(pad), "gpurun_out/bench_detail.json")
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) < bench.LINE_LIMIT <= 4096, len(text)
        back = json.loads(text)
        for k in CONTRACT_KEYS:
            assert k in back, k
        for k in ROOFLINE_KEYS:
            assert k in back["roofline"], k
        assert back["roofline"]["bound"] == "hbm" and back["roofline"]["unit"] == "GB/s" and back["roofline"]["peak"] == 8000.0
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k
        for k in ("workload", "scene_tris", "layout", "rays_per_gpu_per_step"):
            assert k in back["config"], k
        assert "model" not in back["config"]
        assert back["parity"] == {"checked": True, "ok": True, "rays_sampled": 131072, "differ_from_oracle": 0, "differ_from_reference": 1, "shadow_flags_differ": 0}
        assert back["reference_blob"]["combined"] == 4089.7 and back["reference_blob"]["steps"] == 20
        assert back["detail_file"] == "gpurun_out/bench_detail.json"
        assert "detail" not in back


def test_roofline_follows_the_contract():
    """achieved = ALGORITHMIC bytes per launch / mean launch time (SURVEY par. 8(d)); traffic = PMC bytes per launch with the guide's gfx950 corrections"""
    r = full_record()["roofline"]
    n = 1 << 24
    bpr = 64 + 16 + 80 * 25.65 + 48 * 5.46
    assert abs(r["algorithmic_bytes_per_ray"] - bpr) < 1e-9
    assert abs(r["achieved"] - bpr * n / 4.556e-3 / 1e9) < 1e-6
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic"] == 9.9e6 * 2048.0 + 3.3e5 * 1024.0
    assert abs(r["traffic_frac"] - r["traffic"] / 4.556e-3 / 8e12) < 1e-12
    assert abs(r["lane_utilisation"] - 7.8e10 / (64 * 2.0e9)) < 1e-12
    assert abs(r["valu_issue_frac"] - 4.3e9 / 4.556e-3 / 1e9 / 1180.0) < 1e-12
    assert r["primary"]["nodes_per_ray"] == 26.16


def test_no_counters_means_null_traffic_not_a_guess():
    r = bench.roofline_lines(10, 1 << 24, {"primary": 2.3, "diffuse": 4.5}, {"primary": (26.0, 4.0), "diffuse": (25.0, 5.0)}, None, {})
    assert r["traffic"] is None and r["traffic_frac"] is None and r["achieved"] > 0
    line = bench.compact_line({"metric": bench.METRIC, "value": 1.0, "roofline": r, "detail": {}, "config": {}}, None)
    assert line["roofline"]["traffic"] is None and line["cpu_baseline"] is None


def test_a_run_without_roofline_or_reference_still_prints_a_line():
    line = bench.compact_line({"metric": bench.METRIC, "value": 1.0, "detail": {"reference_blob": {"kind": "n/a"}}, "config": {"workload": "w"}}, None)
    assert line["roofline"] is None and line["reference_blob"] == {"kind": "n/a"}
    assert len(json.dumps(line)) < 1500
