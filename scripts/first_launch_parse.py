import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    seen.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in ("k_lin2", "k_schur_all", "k_update", "k_chol_panel_ll", "k_chol_diag_ll"):
    if k in seen:
        v = seen[k]
        print("%-16s first %.1f us  second %.1f us  (%d launches)" % (k, v[0], v[1] if len(v) > 1 else 0, len(v)))
