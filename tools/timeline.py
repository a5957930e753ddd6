"""Print the kernel timeline of one MALA* iteration from a rocprofv3 kernel trace CSV (development aid)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 250
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44], r["Queue_Id"]) for r in rows)
idx = [i for i, k in enumerate(ks) if "propose" in k[2]]
i0, i1 = idx[which], idx[which + 1]
t0 = ks[i0][0]
for k in ks[i0:i1 + 1]:
    print(f"{(k[0]-t0)/1e3:8.1f} {(k[1]-t0)/1e3:8.1f} {(k[1]-k[0])/1e3:6.1f}  q={k[3]} {k[2]}")
