"""durations of the tiled GEMM launches of the last generate step in the newest rocprofv3 kernel trace under gpurun_out/prof_s2s_cfg5"""
import csv, glob, os, sys
fs = sorted(glob.glob("gpurun_out/prof_s2s_cfg5/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows[-31:]:
    if "gemm_bias" in r["Kernel_Name"]:
        out.append(("wide" if "wide" in r["Kernel_Name"] else "64x64", r["Grid_Size_X"], r.get("Grid_Size_Z"), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, 1)))
print(sys.argv[1] if len(sys.argv) > 1 else "", out)
