#!/usr/bin/env python3
"""One training step, launch by launch, from a rocprofv3 --kernel-trace csv directory of `tools/train_bench.py --steps S --warmup 0`.

    python tools/trace_train.py <trace dir> [steps = 3]
Prints the LAST step: start offset, duration, stream (queue) and kernel of every launch -- the side stream's weight-gradient kernels
overlap the main stream, so the sum of kernel times exceeds the span -- then the totals per kernel family with the share of the
span during which each queue was busy."""
import csv, glob, re, sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# a step starts at the first launch of the forward's conditioning (temb) kernel
starts = [i for i, r in enumerate(rows) if "temb" in r["Kernel_Name"]]
first = starts[-1] if starts else 0
nxt = len(rows)
seq = rows[first:nxt]
t0 = int(seq[0]["Start_Timestamp"])
queues = sorted({r.get("Queue_Id", "?") for r in seq})
fam = defaultdict(lambda: [0, 0.0])
busy = defaultdict(float)
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    n = re.sub(r"\(.*", "", re.sub(r"^void |ccn::|at::native::", "", r["Kernel_Name"]))[:58]
    q = queues.index(r.get("Queue_Id", "?"))
    fam[re.sub(r"<.*", "", n)][0] += 1; fam[re.sub(r"<.*", "", n)][1] += d
    busy[q] += d
    print(f"{(s - t0) / 1e3:8.1f} {d:7.1f} us  q{q}  {n}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
span = (max(int(r["End_Timestamp"]) for r in seq) - t0) / 1e3
print(f"\nspan {span:.1f} us; launches {len(seq)}; busy per queue: " + ", ".join(f"q{q} {v:.0f} us" for q, v in sorted(busy.items())))
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us  {n:4d} x  {k}")
