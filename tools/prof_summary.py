"""Dev tool: aggregate a rocprofv3 kernel-trace csv (argv[1]) by (kernel, grid): calls, mean us."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
    key = (r['Kernel_Name'][:70], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v2 = v[len(v) // 3:] if len(v) >= 3 else v
    m = sum(v2) / len(v2)
    if m * len(v) > 100:
        print(f"{k[0]:70s} grid {k[1]}x{k[2]}x{k[3]:3s} calls {len(v):3d} mean {m:8.1f} us")
