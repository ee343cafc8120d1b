#!/bin/bash
# extra PMC passes of one DDIM step (tools/prof_sample.py): LDS conflicts / stalls and vector-memory issue of the conv kernels
export TMPDIR=/tmp
O=gpurun_out/pmc_extra; rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OLDPWD/$O/lds -- python3 $OLDPWD/tools/prof_sample.py --steps 1 > /dev/null 2>&1) || exit 1
python tools/pmc_summary.py $O/lds > $O/r02_pmc_lds_counters.txt
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OLDPWD/$O/vmem -- python3 $OLDPWD/tools/prof_sample.py --steps 1 > /dev/null 2>&1) || exit 1
python tools/pmc_summary.py $O/vmem > $O/r02_pmc_vmem_counters.txt
(cd /tmp && rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_sum --output-format csv -d $OLDPWD/$O/tcp -- python3 $OLDPWD/tools/prof_sample.py --steps 1 > /dev/null 2>&1) || exit 1
python tools/pmc_summary.py $O/tcp > $O/r02_pmc_tcp_counters.txt
rm -rf $O/lds $O/vmem $O/tcp
head -12 $O/r02_pmc_lds_counters.txt; head -12 $O/r02_pmc_vmem_counters.txt; head -8 $O/r02_pmc_tcp_counters.txt
