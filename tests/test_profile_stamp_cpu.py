"""CPU: the stamped per-kernel profile (tools/profile_stamp.py) and what bench.py does with it -- host logic only.  The profile
merges two rocprofv3 --pmc passes, a rocprofv3 --stats CSV and the library profiler's raw HIP-event averages of the SAME bench
command; bench.py subtracts the per-kernel event offset from its live event times and prints rocprofv3's own figure beside."""
import csv
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _write_pmc(d, counter, rows):
    os.makedirs(os.path.join(d, "sub"), exist_ok=True)
    with open(os.path.join(d, "sub", "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Correlation_Id", "Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (k, v) in enumerate(rows):
            w.writerow([i, i, k, counter, v])


def _make_profile(tmp_path):
    conv, warp = "void conv_ring_f32_kernel<0, 0, 1, 0>(ConvArgs)", "void warp_sample_kernel<1>(float const*, float const*, int)"
    _write_pmc(str(tmp_path / "f"), "FETCH_SIZE", [(conv, 1000.0), (conv, 3000.0), (warp, 500.0)])       # KB per dispatch
    _write_pmc(str(tmp_path / "w"), "WRITE_SIZE", [(conv, 4000.0), (conv, 4000.0), (warp, 25000.0)])
    stats = tmp_path / "stats.csv"
    with open(stats, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerow([conv, 1700, 47600000, 28000.0, 50.0, 1, 2, 3])
        w.writerow([warp, 100, 1000000, 10000.0, 1.0, 1, 2, 3])
    raw = tmp_path / "raw.json"
    json.dump({"conv_ring_f32_kernel<0, 0, 1, 0>": {"raw_avg_us": 30.5, "launches": 17.0},
               "warp_sample_kernel": {"raw_avg_us": 13.2, "launches": 1.0}}, open(raw, "w"))
    out = tmp_path / "profile.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_stamp.py"), str(tmp_path / "f"), str(tmp_path / "w"), str(stats),
                        str(raw), str(out), "fake command"], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.load(open(out))


def test_profile_stamp_merges_the_four_runs(tmp_path):
    tab = _make_profile(tmp_path)
    assert tab["__meta__"]["csrc_sha16"] == bench.csrc_sha16() and tab["__meta__"]["command"] == "fake command"
    e = tab["conv_ring_f32_kernel<0, 0, 1, 0>"]
    assert e["pmc_launches"] == 2 and e["l2_fabric_bytes_per_launch"] == pytest.approx((2.0 * 2000.0 + 4000.0) * 1024.0)   # FETCH_SIZE x 2
    assert e["rocprofv3_calls"] == 1700 and e["rocprofv3_avg_us"] == pytest.approx(28.0)
    assert e["hip_event_offset_us"] == pytest.approx(2.5) and e["hip_event_name"] == "conv_ring_f32_kernel<0, 0, 1, 0>"
    w = tab["warp_sample_kernel<1>"]                       # the library's profiler names it without its template argument
    assert w["hip_event_name"] == "warp_sample_kernel" and w["hip_event_offset_us"] == pytest.approx(3.2)
    assert "hbm_bytes_per_launch_corrected" not in e       # the old key (an L2-fabric counter is not HBM alone) is gone


def test_bench_uses_the_stamped_profile(tmp_path, monkeypatch):
    tab = _make_profile(tmp_path)
    os.makedirs(tmp_path / "profiles")
    json.dump(tab, open(tmp_path / "profiles" / "p.json", "w"))
    sha = bench.csrc_sha16()
    monkeypatch.setattr(bench, "csrc_sha16", lambda: sha)              # (the hash is of the real csrc/; only the profiles live under tmp_path)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got, note = bench.load_kernel_profile(True, os.path.join("profiles", "p.json"))
    assert got is not None and "csrc" in note
    assert bench.load_kernel_profile(False, os.path.join("profiles", "p.json"))[0] is None            # not a BASELINE workload
    assert "no profiles" in bench.load_kernel_profile(True, os.path.join("profiles", "absent.json"))[1]
    stale = dict(tab, __meta__=dict(tab["__meta__"], csrc_sha16="0" * 16))
    json.dump(stale, open(tmp_path / "profiles" / "stale.json", "w"))
    assert "stale" in bench.load_kernel_profile(True, os.path.join("profiles", "stale.json"))[1]     # another build's profile is not used
    offs = bench.event_offsets(got)
    assert offs == {"conv_ring_f32_kernel<0, 0, 1, 0>": pytest.approx(2.5), "warp_sample_kernel": pytest.approx(3.2)}
    assert bench.profile_entry(got, "warp_sample_kernel")["rocprofv3_avg_us"] == pytest.approx(10.0)
    # live records: raw 30.5 us events of a 2.2624 GF launch; the offset makes the live figure rocprofv3's
    prof = types.SimpleNamespace(offsets_us=offs, idle_pair_ms=4.6e-3, offset_ms=lambda n: 1e-3 * offs.get(n, 2.3))
    recs = [("conv_ring_f32_kernel<0, 0, 1, 0>", (30.5 - 2.5) * 1e-3, 2.2624e9, 38.5e6)] * 17
    roof, table = bench.roofline_from_records(recs, 1)
    bench.annotate_roofline(roof, got, note, prof)
    assert roof["frac"] == pytest.approx(2.2624e9 / 28e-6 / 1e12 / 157.3) and roof["frac_rocprofv3"] == pytest.approx(roof["frac"])
    assert roof["hip_event_offset_us_subtracted"] == pytest.approx(2.5) and "calibrated" in roof["hip_event_offset_source"]
    assert roof["traffic"] == pytest.approx(8000.0 * 1024.0) and roof["algorithmic_bytes_per_launch"] == pytest.approx(38.5e6)
    assert roof["traffic_over_algorithmic"] == pytest.approx(8000.0 * 1024.0 / 38.5e6) and "Infinity Cache" in roof["traffic_counts"]
    # without a profile: the default offset is named as such
    prof2 = types.SimpleNamespace(offsets_us={}, idle_pair_ms=4.6e-3, offset_ms=lambda n: 2.3e-3)
    roof2, _ = bench.roofline_from_records(recs, 1)
    bench.annotate_roofline(roof2, None, "no profile", prof2)
    assert roof2["traffic"] is None and "half an idle event pair" in roof2["hip_event_offset_source"] and "frac_rocprofv3" not in roof2
