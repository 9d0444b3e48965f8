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
    # the dominant kernel is priced against what binds it on a cache-resident scene: vector-instruction issue
    assert r["bound"] == "valu" and r["unit"] == "Tlane-instr/s" and abs(r["peak"] - 256 * 4 * 32 * 2.4e9 / 1e12) < 1e-9
    launches = prof["wavefront.extend"]["launches"]
    solo_s = solo["wavefront.extend"]["total_ms"] / launches * 1e-3
    lane_instr = (bench.VALU_PER_NODE * stats["nodes"] + bench.VALU_PER_TRIANGLE * stats["triangles"]) / launches
    assert abs(r["achieved"] - lane_instr / solo_s / 1e12) < 1e-6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    # the algorithmic bytes are read against the caches that serve them, the measured memory-side bytes against HBM
    algo = 64 * stats["nodes"] + 64 * stats["triangles"]
    c = r["cache_side"]
    assert abs(c["algorithmic_bytes_per_launch"] * launches - algo) < 1.0
    assert abs(c["achieved"] - algo / launches / solo_s / 1e9) < 1e-3
    assert abs(c["vs_l2_peak"] - c["achieved"] / bench.L2_PEAK_GBS) < 1e-12 and abs(c["vs_infinity_cache_peak"] - c["achieved"] / bench.INFINITY_CACHE_GATHER_GBS) < 1e-12
    h = r["hbm"]
    if h["traffic"]:
        assert abs(h["frac"] - h["traffic"] / solo_s / 1e9 / 8000.0) < 1e-9 and h["frac"] < 1.0 and "replayed" in h["traffic_note"]
    assert 0.69 < r["lane_utilisation"] < 0.71
    assert {k["kernel"] for k in r["other_kernels_solo"]} == {"wavefront.shade", "wavefront.accumulate", "wavefront.generate"}

    def no_frac_above_one(node, path="line"):
        if isinstance(node, dict):
            for key, v in node.items():
                if key == "frac" and v is not None:
                    assert 0.0 <= v <= 1.0, (path, v)
                no_frac_above_one(v, path + "." + key)
        elif isinstance(node, list):
            for i, v in enumerate(node):
                no_frac_above_one(v, f"{path}[{i}]")
    no_frac_above_one(line)
    fc = r.get("from_counters")
    if fc:   # present when profiles/traffic.json carries the raw counters of the passes
        assert 0.0 < fc["frac"] <= fc["issued_frac"] < 1.0 and 0.0 < fc["lanes_active"] <= 1.0 and 0.0 < r["vector_memory"]["ta_busy"] <= 1.0
    assert "ms_per_frame_note" in line
    assert line["scaling"] == "strong" and line["config"]["partition"] == "none" and "strong_scaling" not in line


def test_a_plain_python3_run_with_gpus_2_starts_two_ranks():
    """`python3 bench.py --gpus 2` with no launcher environment starts its own ranks (here without a GPU: the dry run, which
    stops after the rendezvous) and prints rank 0's line, with the world size the ranks saw."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PTR_BENCH_DRY_RUN="1", PTR_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--gpus", "2", "--steps", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dry_run"] is True and line["value"] is None


def test_a_rank_that_fails_fails_the_launcher():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PTR_BENCH_DRY_RUN="1", PTR_BENCH_BACKEND="no-such-backend")
    out = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""


def test_the_launcher_does_not_start_ranks_under_a_profiler():
    """A profiler's preloaded library initialises the GPU before main(): starting other programs from such a process is what the
    launcher exists to avoid, so it says how to profile one rank and stops."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PTR_BENCH_DRY_RUN="1", PTR_BENCH_BACKEND="gloo", HSA_TOOLS_LIB="/nonexistent/librocprofiler-sdk-tool.so")
    out = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and out.stdout.strip() == "" and "Profile one rank" in out.stderr


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
    assert got == per_launch and "valu_one_pipe_share" in counters
    # launches of another size: scaled by the samples per launch, and the note says so
    got, note, _ = bench.pmc_traffic("wavefront.extend", t["workload"], recorded + 3)
    assert got == per_launch * (recorded + 3) / recorded and note.startswith("scaled from")
    if recorded > 1:  # one sample per dispatch is another regime (drain tails per sample): not derived from batched launches
        assert bench.pmc_traffic("wavefront.extend", t["workload"], 1)[0] is None
    assert bench.pmc_traffic("wavefront.extend", "another scene", recorded)[0] is None


def test_real_assets_are_used_only_when_they_are_the_objects_the_pointers_name(tmp_path):
    """The probe behind the line's `data` field: a folder whose .gltf, buffer and image all have the size and sha256 the table
    holds is accepted; one changed byte, a missing file or a Git-LFS pointer in place of the file is not.  With the reference's
    own table nothing in this repository qualifies (its checkout holds pointers only), so bench.py renders the stand-in."""
    import hashlib
    sys.path.insert(0, os.path.join(bench.ROOT, "path-tracer_amd", "python"))
    from path_tracer_amd import assets
    folder = tmp_path / "BistroExterior"
    folder.mkdir()
    files = {"BistroExterior.gltf": json.dumps({"asset": {"version": "2.0"}, "buffers": [{"uri": "BistroExterior.bin", "byteLength": 12}],
                                                "images": [{"uri": "a.png"}]}).encode(),
             "BistroExterior.bin": bytes(range(12)), "a.png": b"\x89PNG not really"}
    table = {}
    for name, data in files.items():
        (folder / name).write_bytes(data)
        table[os.path.join("BistroExterior", name)] = dict(sha256=hashlib.sha256(data).hexdigest(), size=len(data))
    path, report = assets.probe(str(tmp_path), "street_exterior", table)
    assert path == str(folder / "BistroExterior.gltf") and report["used"] and report["files_verified"] == 3
    (folder / "a.png").write_bytes(b"\x89PNG not reallY")                      # same size, another object
    assert assets.probe(str(tmp_path), "street_exterior", table)[0] is None
    (folder / "a.png").unlink()
    assert assets.probe(str(tmp_path), "street_exterior", table)[0] is None
    (folder / "BistroExterior.gltf").write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    assert "pointer" in assets.probe(str(tmp_path), "street_exterior", table)[1]["reason"]
    # the table generated from the reference's pointer files names its three scenes
    with open(assets.TABLE) as f:
        real = json.load(f)["files"]
    assert real["sponza/Sponza.gltf"] == dict(sha256="e17c14edbc06b757dafcfed99f8815af9c5f5d9f9de3f0bc7740c37e68e0d829", size=212504)
    assert real["BistroExterior/BistroExterior.bin"]["size"] == 130333472
    assert assets.probe(os.path.join(bench.ROOT, "models"), "street_exterior")[0] is None
