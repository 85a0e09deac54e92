import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("stack_assemble")]
per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
per = per[-100:]
print("frames", len(per), "period us: median %.1f mean %.1f min %.1f max %.1f" % (sorted(per)[len(per)//2], sum(per)/len(per), min(per), max(per)))
# busy time per frame window
i0 = starts[-101]
busy = 0; last_end = i0; gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < i0: continue
    if s > last_end: gaps.append(((s - last_end) / 1e3, r["Kernel_Name"][:40]))
    busy += e - max(s, last_end) if e > last_end else 0
    last_end = max(last_end, e)
print("busy fraction %.3f" % (busy / (last_end - i0)))
from collections import Counter
c = Counter(); tot = Counter()
for g, n in gaps:
    if g > 3: c[n] += 1; tot[n] += g
for n, k in c.most_common(8): print("gap>3us before %-40s x%d total %.0f us (%.1f us per frame)" % (n, k, tot[n], tot[n] / 100))
