#!/usr/bin/env python3
"""One UNet forward, launch by launch, from a rocprofv3 --kernel-trace csv directory of tools/prof_sample.py (last DDIM step).

    python tools/trace_forward.py <trace dir> [ddim steps of the sample = 3]
The conditioning kernels (temb / linear) run once in front of the step loop; what follows them is `steps` repetitions of the same
launch sequence, of which the last is printed (works for every architecture: C4's stem / head run on the generic kernel)."""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "ccn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cond = [i for i, r in enumerate(rows) if "linear_kernel" in r["Kernel_Name"] or "temb_kernel" in r["Kernel_Name"]]
body = rows[cond[-1] + 1:] if cond else rows
per = len(body) // steps
seq = body[len(body) - per:]
t0 = int(seq[0]["Start_Timestamp"]); tot = 0
for r in seq:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    n = re.sub(r"\(.*", "", re.sub(r"^void |ccn::|_ZN3ccn\d+", "", r["Kernel_Name"]))[:40]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f} us  {n}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
print(f"sum of kernel times {tot:.1f} us; span {(int(seq[-1]['End_Timestamp']) - t0) / 1e3:.1f} us; launches {len(seq)}")
