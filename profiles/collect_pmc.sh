#!/bin/bash
# Recipe behind profiles/r03_pmc_trace.json (round 2: r02_pmc_trace.json) (run on the GPU box through gpurun):
#   gpurun -- 'bash profiles/collect_pmc.sh pmc_r3_bench'
# Five separate rocprofv3 passes of the SAME command (python3 bench.py, N = 1, default
# workload): --kernel-trace --stats alone, then counters only (--pmc with --kernel-trace; never
# with the sys/hip/hsa trace domains).  FETCH_SIZE and WRITE_SIZE need a pass each (TCC slots).
# Raw CSVs land in gpurun_out/<tag>/; profiles/make_pmc_json.py condenses them.
tag=${1:-pmc_r3_bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$tag
sha256sum $R/grace-devel_amd/csrc/trace_kernel.hpp | cut -d' ' -f1 > $R/gpurun_out/$tag/trace_kernel.sha256
CMD="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/stats -o p --output-format csv -- $CMD > $R/gpurun_out/$tag/stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/$tag/insts -o p --output-format csv -- $CMD > $R/gpurun_out/$tag/insts.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/$tag/cycles -o p --output-format csv -- $CMD > $R/gpurun_out/$tag/cycles.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/$tag/fetch -o p --output-format csv -- $CMD > $R/gpurun_out/$tag/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/$tag/write -o p --output-format csv -- $CMD > $R/gpurun_out/$tag/write.log 2>&1 || exit 1
tail -1 $R/gpurun_out/$tag/stats.log
