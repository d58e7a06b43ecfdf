"""Summarise rocprofv3 counter_collection.csv files: mean counter value per kernel name."""
import csv, glob, sys, collections, json, re
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{root}/*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "eval_kernel" not in k: continue
        m = re.search(r"eval_kernel<\w+, \d, (true|false)", k)
        k = "grad" if (m and m.group(1) == "true") else "nograd"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v)/len(v) for c, v in d.items()} for k, d in agg.items()}
print(json.dumps(out, indent=1))
