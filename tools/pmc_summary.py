"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter per dispatch."""
import csv, glob, sys, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in out.items():
    if not k.startswith("gngf::"):
        continue
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(res, indent=1))
