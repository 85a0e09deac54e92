#!/usr/bin/env python3
"""Builds the stamped per-kernel profile of one bench workload from runs of the SAME bench command on the SAME build:
  * two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs)  -> bytes that leave the eight XCD L2s per launch
  * one rocprofv3 --kernel-trace --stats run                            -> rocprofv3's average duration per kernel
  * one `bench.py --dump-event-raw` run                                 -> the library's raw HIP-event time per kernel
and writes profiles/<round>_kernel_profile_<workload>.json, stamped with the hash of csrc/ (bench.py ignores a stale file).

What the counters are: FETCH_SIZE / WRITE_SIZE count TCC -> EA requests, i.e. the traffic on the FABRIC side of the eight
private L2s -- served by the 256 MB Infinity Cache or by HBM, the counter cannot tell which.  The key is therefore
`l2_fabric_bytes_per_launch` (it was `hbm_bytes_per_launch_corrected` up to round 3: a mislabel).  FETCH_SIZE is doubled
(gfx950 counts 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is taken as is; both in KB.

`hip_event_offset_us` = raw HIP-event average - rocprofv3 average: what bench.py subtracts from its LIVE event time of that
kernel, so that the live `roofline.frac` and the committed rocprofv3 CSV agree (the offset is the cost of the event pair around
a kernel in a busy stream: 2.5 .. 3.3 us, kernel dependent, stable to +- 0.1 us from run to run).

  python tools/profile_stamp.py <fetch_dir> <write_dir> <kernel_stats.csv | -> <event_raw.json | -> <out.json> [command text]"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pmc_agg(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = per.setdefault(short_name(r["Kernel_Name"]), {})
            key = r.get("Dispatch_Id") or r.get("Correlation_Id")
            a[key] = a.get(key, 0.0) + float(r["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in per.items()}


def short_name(k):
    """rocprofv3 prints `void name<args>(params)`; the library's profiler `name<args>`."""
    return k.replace("void ", "").split("(")[0].strip()


def stats_table(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[short_name(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def match(name, table):
    """The library's profiler names some kernels without their template arguments (warp_sample_kernel, fc_kernel, ...)."""
    if name in table:
        return name
    c = [k for k in table if k.startswith(name + "<")]
    return max(c, key=lambda k: table[k][0]) if c else None


def main():
    from bench import csrc_sha16
    fetch_dir, write_dir, stats_csv, raw_json, out_path = sys.argv[1:6]
    cmd = sys.argv[6] if len(sys.argv) > 6 else "bench.py --steps 20 --warmup 5 (720p, batch 1)"
    fetch, write = pmc_agg(fetch_dir, "FETCH_SIZE"), pmc_agg(write_dir, "WRITE_SIZE")
    stats = stats_table(stats_csv) if stats_csv != "-" else {}          # ("-": traffic only, tools/xcd_traffic.sh)
    raw = json.load(open(raw_json)) if os.path.exists(raw_json) else {}
    out = {"__meta__": {"csrc_sha16": csrc_sha16(), "command": cmd,
                        "l2_fabric_bytes": "FETCH_SIZE x2 (gfx950 counts 128-B reads at 64 B) + WRITE_SIZE, both in KB: TCC -> EA traffic of the "
                                           "eight XCD L2s (Infinity Cache or HBM behind it), per launch",
                        "hip_event_offset_us": "raw HIP-event average of the library's profiler minus rocprofv3's average duration of the same "
                                               "kernel on the same box: subtracted by bench.py from its live event times"}}
    for k in sorted(set(fetch) | set(stats), key=lambda k: -(stats.get(k, (0, 0))[0] * stats.get(k, (0, 0))[1])):
        e = {}
        if k in fetch:
            n, fkb = fetch[k]
            wkb = write.get(k, (0, 0.0))[1]
            e.update({"pmc_launches": n, "fetch_size_kb_raw": fkb, "write_size_kb": wkb, "l2_fabric_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0})
        if k in stats:
            e.update({"rocprofv3_calls": stats[k][0], "rocprofv3_avg_us": stats[k][1]})
        out[k] = e
    for name, r in raw.items():                      # library profiler name -> {"raw_avg_us", "launches"}
        k = match(name, {kk: (v.get("rocprofv3_calls", 0), 0) for kk, v in out.items() if kk != "__meta__" and "rocprofv3_avg_us" in v})
        if k is None:
            continue
        out[k]["hip_event_name"] = name
        out[k]["hip_event_raw_avg_us"] = r["raw_avg_us"]
        out[k]["hip_event_offset_us"] = r["raw_avg_us"] - out[k]["rocprofv3_avg_us"]
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in [kv for kv in out.items() if kv[0] != "__meta__"][:14]:
        print("%-62s %6s calls  %8.2f us  offset %5s us  %7s MB/launch" % (
            k[:62], v.get("rocprofv3_calls"), v.get("rocprofv3_avg_us", float("nan")),
            "%.2f" % v["hip_event_offset_us"] if "hip_event_offset_us" in v else "-",
            "%.1f" % (v["l2_fabric_bytes_per_launch"] / 1e6) if "l2_fabric_bytes_per_launch" in v else "-"))


if __name__ == "__main__":
    main()
