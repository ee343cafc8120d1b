#!/bin/bash
# One profiling session on the GPU box: everything under profiles/ for a round comes from this script (run via gpurun from the repo root).
#   bash tools/profile_round.sh r02
set -u
R=${1:-rXX}; O=gpurun_out/prof_$R; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 5 --warmup 1 > $O/${R}_bench_bf16.json 2> $O/bench.err
if [ -d .cmp_r02 ]; then (cd .cmp_r02 && python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline > ../$O/${R}_bench_bf16_round2_build_same_box.json 2>> ../$O/bench.err); fi
python bench.py --steps 2 --warmup 1 --dtype fp32 --no-cpu-baseline --no-parity > $O/${R}_bench_fp32.json 2>> $O/bench.err
python bench_train.py > $O/${R}_bench_train_bf16.json 2>> $O/bench.err
python tools/eval_e2e.py --n 256 > $O/${R}_eval_e2e.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stats -- python3 $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_bench_bf16_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/trace -- python3 $OLDPWD/tools/prof_sample.py --steps 3 > /dev/null 2>&1)
python tools/trace_forward.py $O/trace > $O/${R}_forward_launch_by_launch.txt
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python3 $OLDPWD/tools/prof_sample.py --steps 2 > /dev/null 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python3 $OLDPWD/tools/prof_sample.py --steps 2 > /dev/null 2>&1)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/${R}_pmc_traffic.json 2 > /dev/null
python tools/pmc_summary.py $O/pmc_fetch > $O/${R}_pmc_fetch_size_by_kernel.txt
python tools/pmc_summary.py $O/pmc_write > $O/${R}_pmc_write_size_by_kernel.txt
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OLDPWD/$O/pmc_sq -- python3 $OLDPWD/tools/prof_sample.py --steps 1 > /dev/null 2>&1)
python tools/pmc_summary.py $O/pmc_sq > $O/${R}_pmc_sq_counters.txt
rm -rf $O/stats $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
