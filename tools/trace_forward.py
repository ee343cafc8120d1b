#!/usr/bin/env python3
"""One UNet forward, launch by launch, from a rocprofv3 --kernel-trace csv directory of tools/prof_sample.py (last DDIM step)."""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "ccn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
stems = [i for i, n in enumerate(names) if "stem_kernel" in n]
lo = stems[-1]; seq = rows[lo:]
hi = next(i for i, r in enumerate(seq) if "head_kernel" in r["Kernel_Name"]) + 1
seq = seq[:hi]
t0 = int(seq[0]["Start_Timestamp"]); tot = 0
for r in seq:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    n = re.sub(r"\(.*", "", re.sub(r"^void |ccn::|_ZN3ccn\d+", "", r["Kernel_Name"]))[:40]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f} us  {n}")
print(f"sum of kernel times {tot:.1f} us; span {(int(seq[-1]['End_Timestamp']) - t0) / 1e3:.1f} us; launches {len(seq)}")
