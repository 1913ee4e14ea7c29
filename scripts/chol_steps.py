"""per-column durations of the factorisation launches of the FIRST factorisation of the last solve in a rocprofv3 kernel trace"""
import csv, glob, sys
rows = sorted(csv.DictReader(open(glob.glob(sys.argv[1])[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_init_pads")]
sel = rows[marks[-1]:]
ch = [r for r in sel if r["Kernel_Name"].startswith("k_chol_step")]
nb = 23
for it in (0, 1, 9):
    c = ch[it * nb:(it + 1) * nb]
    print("factorisation %d:" % it, [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in c])
print("grid:", [int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) for r in ch[:nb]])
tr = [r for r in sel if r["Kernel_Name"].startswith("k_trsv")]
print("trsv:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in tr])
