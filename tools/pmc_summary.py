import collections, csv, glob, sys
for d in sys.argv[2:]:
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
