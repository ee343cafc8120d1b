#!/bin/bash
# C4 (BASELINE configs[3]: 512 px, base 192, ch_mult (1,2,2,4), 100 DDIM steps, batch 4) on one GPU: bench line, rocprof kernel stats,
# launch-by-launch trace of one forward.   bash tools/profile_c4.sh r03
set -u
R=${1:-rXX}; O=gpurun_out/prof_c4_$R; mkdir -p $O; export TMPDIR=/tmp
C4="--size 512 --base 192 --ch-mult 1,2,2,4 --ddim-steps 100 --batch 4"
python bench.py $C4 --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/${R}_c4_bench_bf16.json 2> $O/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stats -- python3 $OLDPWD/bench.py $C4 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-roofline > /dev/null 2>&1)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_c4_bench_bf16_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/trace -- python3 $OLDPWD/tools/prof_sample.py --size 512 --base 192 --ch-mult 1,2,2,4 --batch 4 --steps 2 > /dev/null 2>&1)
python tools/trace_forward.py $O/trace 2 > $O/${R}_c4_forward_launch_by_launch.txt
rm -rf $O/stats $O/trace
ls -la $O
