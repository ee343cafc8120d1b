#!/usr/bin/env python3
"""HBM traffic of the dominant kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KiB).

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming
read (16 B / lane), so the read side is doubled; WRITE_SIZE reads exactly for 16-byte stores.  Output: JSON with
per-launch averages, consumed by bench.py (`roofline.traffic`)."""
import csv, glob, json, sys
fetch_dir, write_dir, out, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
def total(d, counter, pred):
    tot, n = 0.0, 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and pred(r["Kernel_Name"]):
                tot += float(r["Counter_Value"]); n += 1
    return tot, n
# 3x3 stride-1 family: the persistent kernel (conv_pr_kernel<9, ...>) plus the warp-specialised / free-running kernels
# instantiated with NTAPS = 9 (the 32-pixel-level layers and any shape the persistent kernel does not take)
is_c3 = lambda k: ("conv_pr_kernel" in k and ("<9," in k or "ILi9E" in k)) or (("conv_ws_kernel" in k or "conv_fr_kernel" in k) and "Li9E" in k)
f, nf = total(fetch_dir, "FETCH_SIZE", is_c3)
w, nw = total(write_dir, "WRITE_SIZE", is_c3)
assert nf == nw and nf > 0, (nf, nw)
res = {"kernel_family": "conv3x3_s1_igemm", "launches_profiled": nf, "ddim_steps_profiled": steps,
       "fetch_size_kib_raw_per_launch": f / nf, "write_size_kib_per_launch": w / nw,
       "read_bytes_per_launch_corrected_x2": 2 * f * 1024 / nf, "write_bytes_per_launch": w * 1024 / nw,
       "hbm_bytes_per_launch": (2 * f + w) * 1024 / nf,
       "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), batch 8, 256 px, bf16; FETCH_SIZE doubled per the gfx950 note"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
