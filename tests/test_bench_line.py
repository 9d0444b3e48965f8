"""The benchmark's result line, assembled from synthetic measurements on the CPU: the fields the measurement contract names are
there and mean what they say (no GPU: bench.result_line and bench.kernel_roofline are pure functions of their inputs)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def fake_inputs(steps=20, per_dispatch=10):
    args = bench.make_parser().parse_args(["--steps", str(steps)])
    sc = {"kind": "street_exterior", "triangles": type("T", (), {"shape": (2_796_540, 9)})()}
    rays = 5_150_000 * steps
    stats = dict(rays=rays, nodes=int(19.8 * rays), triangles=int(3.27 * rays), shaded=int(0.9 * rays), frames=steps,
                 wave_inner_steps=int(19.8 * rays / 64 / 0.7), wave_leaf_steps=1, idle_lane_steps=0)
    launches = 4 * (steps // per_dispatch)
    prof = {"wavefront.extend": dict(launches=launches, total_ms=2.9 * launches)}
    solo = {"wavefront.extend": dict(launches=launches, total_ms=2.07 * launches), "wavefront.shade": dict(launches=launches, total_ms=0.76 * launches),
            "wavefront.accumulate": dict(launches=steps // per_dispatch, total_ms=0.24 * steps // per_dispatch),
            "wavefront.generate": dict(launches=steps // per_dispatch, total_ms=0.15 * steps // per_dispatch)}
    bench.pmc_traffic.samples_per_dispatch = per_dispatch
    return args, sc, stats, prof, solo, rays


def test_one_gpu_line_has_the_contract_fields():
    args, sc, stats, prof, solo, rays = fake_inputs()
    elapsed = 1.09e-3 * args.steps
    line = bench.result_line(args, sc, 1920, 1080, 1, True, elapsed, None, float(rays), stats, prof, solo, args.steps, {})
    json.dumps(line)                                                   # serialisable as it is
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line, key
    assert line["metric"] == "Mrays/s" and line["unit"] == "Mrays/s" and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert abs(line["value"] - rays / elapsed / 1e6) < 1e-6 and abs(line["ms_per_step"] - 1.09) < 1e-9
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # achieved = algorithmic bytes of the dominant kernel's launches / their summed event time
    algo = 64 * stats["nodes"] + 64 * stats["triangles"]
    assert abs(r["achieved"] - algo / (prof["wavefront.extend"]["total_ms"] * 1e-3) / 1e9) < 1e-6
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["algorithmic_bytes_per_launch"] * r["launches"] - algo) < 1.0
    assert 0.69 < r["lane_utilisation"] < 0.71
    assert {k["kernel"] for k in r["other_kernels_solo"]} == {"wavefront.shade", "wavefront.accumulate", "wavefront.generate"}
    v = line["valu_roofline"]
    assert v["bound"] == "valu" and 0.0 < v["frac"] < 1.0 and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-12
    assert line["scaling"] == "strong" and line["config"]["partition"] == "none" and "strong_scaling" not in line


def test_multi_gpu_lines_say_how_the_work_was_shared():
    args, sc, stats, prof, solo, rays = fake_inputs()
    tiles = bench.result_line(args, sc, 1920, 1080, 8, True, 0.004, 0.024, float(rays), stats, prof, solo, args.steps, {})
    assert tiles["scaling"] == "strong" and tiles["config"]["partition"] == "tiles" and tiles["n_gpus"] == 8
    assert abs(tiles["strong_scaling"]["efficiency"] - 0.024 / (8 * 0.004)) < 1e-12
    samples = bench.result_line(args, sc, 1920, 1080, 8, False, 0.004, 0.0, float(rays), stats, prof, solo, args.steps, {})
    assert samples["scaling"] == "weak" and samples["config"]["partition"] == "samples" and "strong_scaling" not in samples


def test_traffic_is_reported_only_for_the_workload_it_was_measured_on():
    with open(os.path.join(bench.ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    recorded = t.get("samples_per_dispatch", 1)
    per_launch = t["kernels"]["wavefront.extend"]["bytes_per_launch"]
    got, _, counters = bench.pmc_traffic("wavefront.extend", t["workload"], recorded)
    assert got == per_launch and "valu_busy" in counters
    # launches of another size: scaled by the samples per launch, and the note says so
    got, note, _ = bench.pmc_traffic("wavefront.extend", t["workload"], recorded + 3)
    assert got == per_launch * (recorded + 3) / recorded and note.startswith("scaled from")
    if recorded > 1:  # one sample per dispatch is another regime (drain tails per sample): not derived from batched launches
        assert bench.pmc_traffic("wavefront.extend", t["workload"], 1)[0] is None
    assert bench.pmc_traffic("wavefront.extend", "another scene", recorded)[0] is None
