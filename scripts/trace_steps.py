"""per-launch durations of the factorisation kernels from a rocprofv3 kernel trace csv: python trace_steps.py <csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in d.items():
    if "chol" not in k:
        continue
    durs = [b - a for a, b in v]
    mx = max(durs)
    act = [x for x in durs if x > 0.2 * mx]
    print("%-16s launches %5d  active %5d  active total %.2f ms  avg %.1f us  max %.1f us" % (k, len(durs), len(act), sum(act) / 1e6, sum(act) / len(act) / 1e3, mx / 1e3))
# one factorisation: the launches between two consecutive k_schur_off launches
so = sorted(a for a, b in d.get("k_schur_off", []))
if len(so) > 3:
    t0, t1 = so[2], so[3]
    seq = [(r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])) for r in rows if t0 <= int(r["Start_Timestamp"]) < t1 and "chol" in r["Kernel_Name"]]
    print("one iteration: %d factor launches, busy %.2f ms, span %.2f ms" % (len(seq), sum(x[1] for x in seq) / 1e6, (seq[-1][2] + seq[-1][1] - seq[0][2]) / 1e6))
    print(" ".join("%s%.0f" % ("d" if "diag" in n else ("c" if "col" in n else "p"), t / 1e3) for n, t, _ in seq))
