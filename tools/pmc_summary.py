"""Dev tool: summarise rocprofv3 --pmc CSV output for kernels matching a substring."""
import csv, glob, collections, sys
root, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in agg.items():
        print(f"   {k:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
for f in sorted(glob.glob(f"{root}/**/*_kernel_trace.csv", recursive=True)):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    print(f, "dur us", [round(x) for x in d])
