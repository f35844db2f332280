"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import csv, glob, sys, collections, re
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "uno::" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
