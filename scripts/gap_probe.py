"""busy time vs span of the kernels of the last run in a rocprofv3 kernel trace csv: python gap_probe.py <csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run starts at the last k_reset
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_reset"))
run = rows[idx:]
t0, t1 = int(run[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in run)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in run)
print("last run: %d launches, span %.3f ms, sum of kernel durations %.3f ms (%.0f %%)" % (len(run), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
d = collections.defaultdict(lambda: [0, 0])
for r in run:
    k = r["Kernel_Name"].split("(")[0]
    d[k][0] += 1; d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print("  %-18s %4d launches  %8.1f us total  %6.2f us avg" % (k, c, t / 1e3, t / 1e3 / c))
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(run, run[1:])]
print("gap between consecutive launches: median %.2f us, mean %.2f us" % (sorted(gaps)[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3))
