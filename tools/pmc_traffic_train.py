#!/usr/bin/env python3
"""HBM traffic per launch of the training step's kernel families from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_tr_fetch --output-format csv -- python3 bench_train.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_tr_write --output-format csv -- python3 bench_train.py ... (same)
    python tools/pmc_traffic_train.py gpurun_out/pmc_tr_fetch gpurun_out/pmc_tr_write profiles/r01_pmc_traffic_train.json

FETCH_SIZE is doubled for the wide (16 B / lane) streaming reads, as in tools/pmc_traffic.py (MI355X_MICROARCH.md, HBM)."""
import csv, glob, json, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
FAMS = {"conv_weight_grad": lambda k: "wgrad_bf16" in k, "weight_grad_reduce": lambda k: "wgrad_reduce" in k,
        "gn_bwd_reduce": lambda k: "gn_bwd_reduce" in k, "gn_bwd_apply": lambda k: "gn_bwd_apply" in k,
        "conv_pr": lambda k: "conv_pr_kernel" in k, "conv_ws_fr": lambda k: "conv_ws_kernel" in k or "conv_fr_kernel" in k,
        "gn_act": lambda k: "gn_act_kernel" in k or "gn_act_fused" in k, "stem_head": lambda k: "stem_kernel" in k or "head_kernel" in k,
        "weight_repack": lambda k: "pack_group_kernel" in k, "adamw": lambda k: "adamw_kernel" in k}
def totals(d, counter):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for fam, pred in FAMS.items():
                if pred(r["Kernel_Name"]):
                    tot[fam][0] += float(r["Counter_Value"]); tot[fam][1] += 1
    return tot
f, w = totals(fetch_dir, "FETCH_SIZE"), totals(write_dir, "WRITE_SIZE")
res = {}
for fam in FAMS:
    if f[fam][1] == 0:
        continue
    n = f[fam][1]
    res[fam] = {"launches_profiled": n, "read_mb_per_launch_corrected_x2": round(2 * f[fam][0] * 1024 / n / 1e6, 2),
                "write_mb_per_launch": round(w[fam][0] * 1024 / max(w[fam][1], 1) / 1e6, 2)}
json.dump({"note": "separate --pmc passes, training step at 256 px / batch 4 / bf16; FETCH_SIZE doubled per the gfx950 note; per launch, averaged over every launch "
                   "of the family in the profiled steps (all layer shapes)", "families": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
