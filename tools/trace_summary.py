#!/usr/bin/env python3
"""Per (kernel, grid, LDS) summary of a rocprofv3 --kernel-trace csv directory."""
import csv, sys, glob, collections, re
d = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not ("ccn" in name): continue
        short = re.sub(r"_ZN3ccn\d+", "", name)[:44]
        key = (short, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"total ccn kernel time {tot/1e6:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/1e6:8.3f} ms {100*sum(v)/tot:5.1f}%  n={len(v):4d} avg={sum(v)/len(v)/1e3:8.1f}us min={min(v)/1e3:8.1f}  blocks={k[1]:5d} wg={k[2]} lds={k[3]} vgpr={k[4]}+{k[5]}  {k[0]}")
