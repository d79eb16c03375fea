"""Dev tool: print the kernels of the LAST pass in a rocprofv3 kernel-trace csv (argv[1]), argv[2] = passes."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // passes
last = rows[-n:]
tot = 0
for r in last:
    t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += t
    if t > 15:
        print(f"{r['Kernel_Name'][:58]:58s} {t:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print('kernels', n, 'total us', round(tot))
