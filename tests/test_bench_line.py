"""The ONE JSON line bench.py leaves on stdout must fit the driver's tail of stdout (round 3's line was 26 KB; the driver keeps
~8 KB and parsed nothing).  CPU-only: the compact line is a pure function of the full record, so the committed round-3 record
(profiles/r03_bench.json, every leg present) — with this round's k-th legs and CPU k-mode baseline added — is shaped here
without a GPU; tests/test_dist.py::test_bench_line_contract_on_a_small_workload does the same on a real run."""
import contextlib
import io
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def full_record():
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    leg = {"max_num_hits": 50, "max_divergence": None, "queries": 10000, "wall_ms": 123.456789, "kernel_ms": 111.111111,
           "queries_per_s_wall": 81000.123456, "scans": 3, "launches": 17, "rows": 612345, "verified": True,
           "roofline": dict(full["besthit_unbounded"]["roofline"])}
    full["kth"] = {"note": "x" * 400, "kth5": leg, "kth50": leg, "kth5_d5": leg, "kth50_d5": leg}
    full["cpu_baseline"]["kmode"] = {"value": 1.2345678, "unit": "query seqs/s", "cores": 1, "queries": 4, "max_num_hits": 5,
                                     "sample": "y" * 300}
    full["stream"]["fetched_over_plane"] = 1.0002761030351117
    ix = {"max_divergence": 5, "served": True, "kernel": "smafa::index_probe_kernel<5, 5, 2>", "kernel_ms": 0.0671234, "verified": True,
          "queries_per_s": 148981234.5, "times_the_scan_kernels": 25.3456, "index": {"blocks": 6, "build_ms_call": 25.41234}}
    full["indexed"] = dict(ix, besthit_planted={"scan_kernels": {"wall_ms": 8.31234}, "with_index": {"wall_ms": 0.36789}, "verified": True})
    full["related"]["besthit_novel_members"]["indexed"] = {"scan_kernels": {"wall_ms": 16.31234}, "with_index": {"wall_ms": 5.6789},
                                                           "verified": True}
    for name in full["configs"]:
        if "cluster" not in name:
            full["configs"][name]["indexed"] = dict(ix)
    full["skipped_for_time"] = [{"leg": "configs[4] cluster", "at_s": 399.0, "needs_s": 60}]
    return full


def test_compact_line_fits_and_keeps_the_contract():
    full = full_record()
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"))
    assert len(line) <= bench.LINE_LIMIT == 4096, len(line)
    out = json.loads(line)
    for key in CONTRACT:
        assert key in out, key
    assert out["value"] == float("%.7g" % full["value"]) and abs(out["ms_per_step"] - full["ms_per_step"]) < 1e-5
    assert abs(out["value"] - 10000 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-5
    assert out["metric"] == full["metric"] and out["unit"] == "query seqs/s" and out["vs_baseline"] is None and out["dtype"] == "u32"
    cfg = out["config"]
    assert len(cfg["workload"]) <= 200 and cfg["db_rows"] == 10_000_000 and cfg["queries_per_gpu"] == 10_000 and cfg["max_divergence"] == 5
    r = out["roofline"]
    assert r["bound"] == "valu" and r["kernel"].startswith("smafa::scan_zone_kernel") and 0 < r["frac"] < 1 and r["traffic"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "Tlane-op/s" and r["insts_source_is_this_build"] is True
    h = r["hbm_stream"]
    assert h["served_by"] == "hbm" and h["bytes_per_pass"] > (256 << 20) and 0.5 < h["frac_wall"] < 1 and 0.5 < h["frac_kernel"] < 1
    assert h["peak"] == 8000.0 and h["unit"] == "GB/s" and abs(h["fetched_over_plane"] - 1.0003) < 1e-3
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["sample"] and c["cpu_model"] and c["kmode"]["value"] > 0
    legs = out["legs"]
    for name in ("unfiltered", "bound8", "bound14", "bound24", "besthit_mixed", "besthit_far", "besthit_novel", "related", "host_api",
                 "kth5", "kth50", "kth5_d5", "kth50_d5", "cfg1", "cfg2", "cfg2N", "cfg3", "cfg4"):
        assert name in legs and legs[name][0] > 0, name
    assert legs["cfg2"][3] is True and legs["kth50"][3] is True and legs["cfg4"][5] > 1000
    assert legs["idx_besthit"] == [8.312, 0.3679, True] and legs["idx_novel"] == [16.31, 5.679, True]
    for name in ("idx", "cfg1i", "cfg2i", "cfg2Ni", "cfg3i"):
        assert legs[name] == [0.06712, 149000000.0, 25.3, True, 25.4], (name, legs[name])
    assert out["full_record"] == "gpurun_out/bench_full.json" and out["skipped_for_time"] == ["configs[4] cluster"]


def test_compact_line_sheds_blocks_rather_than_overrun():
    full = full_record()
    full["config"]["workload"] = "w" * 5000          # cut to 200
    full["cpu_baseline"]["sample"] = "s" * 5000      # cut to 160
    for i in range(40):                              # a leg list nobody planned for
        full["configs"]["extra leg %d" % i] = dict(full["configs"]["configs[1] 1M aa"])
    out = bench.compact_line(full, None)
    assert len(json.dumps(out)) <= bench.LINE_LIMIT
    for key in CONTRACT:
        assert key in out, key


def test_emit_writes_the_full_record_and_prints_the_line_last(tmp_path):
    full = full_record()
    path = str(tmp_path / "sub" / "full.json")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        print("something a library printed earlier")
        bench.emit(full, path)
    lines = buf.getvalue().splitlines()
    assert len(lines[-1]) <= bench.LINE_LIMIT and json.loads(lines[-1])["full_record"] == path
    assert json.load(open(path)) == full
    # an unwritable path costs the file, not the line
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.emit(full, "/proc/nonexistent/dir/full.json")
    assert json.loads(buf.getvalue().splitlines()[-1])["full_record"] is None
