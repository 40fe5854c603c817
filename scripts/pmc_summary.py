"""Summarise rocprofv3 --pmc counter_collection CSVs.
    pmc_summary.py DIR OUT.csv           mean counter value per (kernel, grid, workgroup, counter)
    pmc_summary.py --raw DIR OUT.csv     the same (kept as a name for SQ counter sets)
    pmc_summary.py --mfma DIR OUT.csv    DIR/pmc_<clips>/...: matrix-pipe busy fraction of the largest kernel per clip count
                                         (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), MI355X_MICROARCH.md)"""
import csv
import glob
import os
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
mode = next((a for a in sys.argv[1:] if a.startswith("--")), "")
root, out = args[0], args[1]


def collect(folder):
    rows = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(folder + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name", "")
                short = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
                key = (short, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("Counter_Name", ""))
                rows[key][0] += float(r.get("Counter_Value", 0) or 0)
                rows[key][1] += 1
    return rows


if mode == "--mfma":
    with open(out, "w") as f:
        f.write("clips,kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_frac\n")
        for folder in sorted(glob.glob(os.path.join(root, "pmc_*"))):
            if not os.path.isdir(folder):
                continue
            clips = os.path.basename(folder)[4:]
            per = defaultdict(dict)
            for (k, g, w, c), (tot, n) in collect(folder).items():
                per[k][c] = tot
                per[k]["n"] = n
            for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:2]:
                gui = v.get("GRBM_GUI_ACTIVE", 0)
                frac = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui / 8) if gui else 0.0
                f.write(f"{clips},\"{k}\",{v['n']},{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{v.get('SQ_BUSY_CYCLES', 0):.0f},{gui:.0f},{frac:.4f}\n")
else:
    rows = collect(root)
    with open(out, "w") as f:
        f.write("kernel,grid_size,workgroup_size,counter,dispatches,mean_value\n")
        for (k, g, w, c), (tot, n) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
            f.write(f"\"{k}\",{g},{w},{c},{n},{tot / max(n, 1):.3f}\n")
print(open(out).read()[:3000])
