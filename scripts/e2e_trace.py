"""one resident run and one end-to-end call of the same 4096 windows under rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/e2e_trace.py ;  python3 scripts/e2e_trace.py --summarise DIR/*/*_kernel_trace.csv"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, collections
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_init_pads")]
    # phases are separated by the uploads: [upload+warm run+timed run] then the end-to-end call (several uploads)
    t0 = int(rows[0]["Start_Timestamp"])
    def summarise(sel, title):
        by = collections.defaultdict(float)
        for r in sel: by[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        span = (max(int(r["End_Timestamp"]) for r in sel) - min(int(r["Start_Timestamp"]) for r in sel)) / 1e6
        print("%s: span %.1f ms, kernel time %.1f ms" % (title, span, sum(by.values())))
        for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]: print("   %-28s %8.1f ms" % (k, v))
    nch = int(sys.argv[3]) if len(sys.argv) > 3 else 5   # chunks (= uploads) of one end-to-end call
    def busy(sel):
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
        tot, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > cur_e: tot += cur_e - cur_s; cur_s, cur_e = a, b
            else: cur_e = max(cur_e, b)
        return (tot + cur_e - cur_s) / 1e6
    res = rows[:marks[1]]
    last = rows[marks[-nch]:]
    summarise(res, "upload + 2 resident runs")
    print("   device busy %.1f ms" % busy(res))
    summarise(last, "last end-to-end call")
    print("   device busy %.1f ms" % busy(last))
    for i in range(nch):
        a = marks[-nch + i]; b = marks[-nch + i + 1] if i + 1 < nch else len(rows)
        print("   chunk %d: first kernel at %.1f ms" % (i, (int(rows[a]["Start_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6))
    sys.exit(0)
import bench
from mc_slam_amd import backend
n, nd = 4096, 64
wins = bench.make_windows([("c3", 100 + i, False, "caller") for i in range(nd)], 1)
batch = [wins[i % nd] for i in range(n)]
ba = backend.LocalBA(0)
ba.upload(batch); ba.run(); ba.run()
packed = ba.pack(batch)
ba.solve_packed(packed)
ba.pack_reset(packed)
ba.solve_packed(packed)
ba.pack_reset(packed)
t = time.perf_counter(); ba.solve_packed(packed); print("e2e call %.1f ms" % (1e3 * (time.perf_counter() - t)), file=sys.stderr)
