"""Dev tool: print the kernel timeline between two occurrences of a marker kernel in a rocprofv3 kernel trace."""
import csv, glob, sys
root, marker, occ = sys.argv[1], sys.argv[2], int(sys.argv[3])
tr = list(csv.DictReader(open(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if marker in r["Kernel_Name"]]
i0, i1 = idx[occ], idx[occ + 1]
base = int(tr[i0]["Start_Timestamp"]); prev = None
for r in tr[i0:i1 + 1]:
    s = int(r["Start_Timestamp"]) - base; e = int(r["End_Timestamp"]) - base
    gap = (s - prev) if prev is not None else 0; prev = e
    name = r["Kernel_Name"][:50]
    print(f"{name:50s} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>4s} start={s/1e3:8.1f} dur={(e-s)/1e3:7.1f} gap={gap/1e3:5.1f}")
