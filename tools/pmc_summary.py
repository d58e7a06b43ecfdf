"""Summarise rocprofv3 counter_collection.csv files: mean counter value per launch, per kernel class.
Classes: eval_kernel -> "grad" / "nograd" (+ "_pre" for the split-mode instance that reads the clearance tables),
clearance_kernel -> "clearance"."""
import csv, glob, sys, collections, json, re
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
names = collections.defaultdict(set)
for f in sorted(glob.glob(f"{root}/*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"eval_kernel<\w+, \d, (true|false), \d, (true|false)", k)
        if m:
            cls = ("grad" if m.group(1) == "true" else "nograd") + ("_pre" if m.group(2) == "true" else "")
        elif "clearance_kernel" in k:
            cls = "clearance"
        else:
            continue
        names[cls].add(k.split("(")[0].replace("void ", ""))
        agg[cls][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k in out:
    out[k]["kernel_names"] = sorted(names[k])
# provenance: the counters are a property of the kernels they were taken on (bench.py refuses a summary of other sources)
import hashlib, pathlib
root_dir = pathlib.Path(__file__).resolve().parents[1]
h = hashlib.sha256()
for rel in ("motion_planning_amd/csrc/mp_kernels.hpp", "motion_planning_amd/csrc/mp_device.hpp", "motion_planning_amd/csrc/mp_api.hip"):
    h.update((root_dir / rel).read_bytes())
out["_provenance"] = {"kernel_sources_sha256": h.hexdigest(), "sources": "mp_kernels.hpp + mp_device.hpp + mp_api.hip"}
print(json.dumps(out, indent=1))
