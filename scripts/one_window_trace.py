"""kernel timeline of ONE vba_solve of a C3 window: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/one_window_trace.py
   then: python3 scripts/one_window_trace.py --summarise DIR/*/*_kernel_trace.csv"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, collections
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_init_pads")]
    sel = rows[marks[-1]:]          # the last solve
    by = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        e = by[r["Kernel_Name"].split("(")[0]]; e[0] += 1; e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in by.values())
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(sel[:-1], sel[1:])]
    print("last solve: %d kernels, span %.0f us, kernel time %.0f us, gaps %.0f us (median %.2f us, max %.1f us)" % (len(sel), span, busy, sum(gaps), sorted(gaps)[len(gaps) // 2], max(gaps)))
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]): print("   %-22s x%4d  %8.1f us  (%.2f us each)" % (k, v[0], v[1], v[1] / v[0]))
    big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]
    for g, i in big: print("   gap %.1f us between %s and %s" % (g, sel[i]["Kernel_Name"].split("(")[0], sel[i + 1]["Kernel_Name"].split("(")[0]))
    sys.exit(0)
from mc_slam_amd import backend, synth
p = synth.config_c3(3)
ba = backend.LocalBA(0)
for _ in range(4): ba.solve(p)
