"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per (kernel, grid, workgroup)."""
import csv
import glob
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
rows = defaultdict(lambda: [0.0, 0])
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            short = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
            key = (short, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("Counter_Name", ""))
            rows[key][0] += float(r.get("Counter_Value", 0) or 0)
            rows[key][1] += 1
with open(out, "w") as f:
    f.write("kernel,grid_size,workgroup_size,counter,dispatches,mean_value\n")
    for (k, g, w, c), (tot, n) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
        f.write(f"\"{k}\",{g},{w},{c},{n},{tot / max(n, 1):.3f}\n")
print(open(out).read()[:3000])
