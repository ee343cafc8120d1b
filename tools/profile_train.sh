#!/bin/bash
# Training-step evidence for profiles/ (run via gpurun from the repo root):  bash tools/profile_train.sh r03 [older worktree for the A/B]
set -u
R=${1:-rXX}; CMP=${2:-}; O=gpurun_out/prof_train_$R; mkdir -p $O
export TMPDIR=/tmp
python bench_train.py > $O/${R}_bench_train_bf16.json 2> $O/bench.err
if [ -n "$CMP" ] && [ -d "$CMP" ]; then
  (cd $CMP && python bench_train.py --no-cpu-baseline) > $O/${R}_bench_train_bf16_older_build_same_box.json 2>> $O/bench.err   # the worktree's own package + library
fi
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stats -- python3 $OLDPWD/bench_train.py --no-cpu-baseline --no-roofline > /dev/null 2>&1)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_bench_train_bf16_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/trace -- python3 $OLDPWD/tools/train_bench.py --steps 3 --warmup 0 > /dev/null 2>&1)
python tools/trace_train.py $O/trace > $O/${R}_train_step_launch_by_launch.txt
rm -rf $O/stats $O/trace
ls -la $O
# HBM bytes per launch of the kernel families (two --pmc passes, kernel trace only)
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python3 $OLDPWD/bench_train.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python3 $OLDPWD/bench_train.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
python tools/pmc_traffic_train.py $O/pmc_fetch $O/pmc_write $O/${R}_pmc_traffic_train.json > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write
ls -la $O
