"""per-block-column durations of the left-looking factorisation kernels from a rocprofv3 kernel trace:
python scripts/ll_columns.py <kernel_trace.csv>   (one window group / one stream: VBA_STREAMS=1)"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)   # kernel -> durations in launch order
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if n in ("k_chol_panel_ll", "k_chol_diag_ll", "k_chol_diag_ll2"):
        seq[n].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
for n, v in seq.items():
    # one factorisation = a run of launches until the grid size pattern repeats; print the second factorisation seen (warm)
    grids = [g for _, g in v]
    per = next((p for p in range(1, len(grids)) if grids[p:2 * p] == grids[:p]), len(grids))
    if n.startswith("k_chol_diag") and "k_chol_panel_ll" in seq:   # the diagonal kernel has one grid for every column: one more launch per factorisation than the panel kernel
        gp = [g for _, g in seq["k_chol_panel_ll"]]
        per = next((p for p in range(1, len(gp)) if gp[p:2 * p] == gp[:p]), len(gp)) + 1
    print(n, "launches per factorisation:", per)
    nf = len(v) // per
    for j in range(per):
        ds = [v[f * per + j][0] for f in range(nf)]
        act = [d for d in ds if d > 8.0]
        print("  col %2d grid %8d  median active %.1f us (%d of %d launches active)" % (j, grids[j], sorted(act)[len(act) // 2] if act else 0.0, len(act), nf))
