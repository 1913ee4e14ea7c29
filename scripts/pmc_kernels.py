"""mean counter value per kernel (active launches) from a rocprofv3 --pmc counter_collection csv"""
import csv, sys, collections, glob, os
d = collections.defaultdict(lambda: collections.defaultdict(list))
src = sys.argv[1]
files = glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True) if os.path.isdir(src) else [src]
for f in files:   # a directory: every PMC pass below it (one pass per counter group)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2:] or ["k_schur_off", "k_schur_diag", "k_lin2", "k_chol_panel_ll", "k_update"]
for k in want:
    if k not in d: continue
    print(k)
    for c, v in sorted(d[k].items()):
        mx = max(v); act = [x for x in v if x > 0.2 * mx] or v
        print("   %-34s mean(active) %.4g   (n=%d)" % (c, sum(act) / len(act), len(act)))
