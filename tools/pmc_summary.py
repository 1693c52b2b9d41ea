import csv, glob, collections, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
for p in sorted(glob.glob(f"{d}/p*/")):
    for f in glob.glob(f"{p}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = v[3:] if len(v) > 3 else v
            print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
