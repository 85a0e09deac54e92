#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into HBM bytes
per launch per kernel.  FETCH_SIZE is doubled (gfx950 counts 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM /
rocprofv3 section); WRITE_SIZE is taken as is.  Both counters are in KB.
  python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [command text for the file's header]"""
import csv, glob, json, os, sys


def agg(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            a = per.setdefault(k, {})
            key = r.get("Dispatch_Id") or r.get("Correlation_Id")
            a[key] = a.get(key, 0.0) + float(r["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in per.items()}


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # noqa: E402  (the stamp bench.py checks before it reports `traffic`)

fetch, write = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {"__meta__": {"csrc_sha16": csrc_sha16(), "command": sys.argv[4] if len(sys.argv) > 4 else "bench.py --steps 20 --warmup 5 (720p, batch 1)",
                    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B reads at 64 B), WRITE_SIZE as is; both in KB"}}
for k, (n, fkb) in sorted(fetch.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    wkb = write.get(k, (0, 0.0))[1]
    out[k] = {"launches": n, "fetch_size_kb_raw": fkb, "write_size_kb": wkb,
              "hbm_bytes_per_launch_corrected": (2.0 * fkb + wkb) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in [kv for kv in out.items() if kv[0] != "__meta__"][:12]:
    print("%-70s n=%5d  %.1f MB/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch_corrected"] / 1e6))
