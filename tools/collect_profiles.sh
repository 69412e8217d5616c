#!/bin/bash
# Round evidence on the MI355X box (run through gpurun from the repo root): default bench line, rocprofv3 kernel stats of the
# same workload (serial grouped launches, so that kernel names / durations match bench.py's HIP-event leg), PMC passes.
set -x
R=${1:-r02}
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-e2e --no-train"
SOME_AMD_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats -- $B > $O/${R}_bench_under_rocprof.json 2> $O/${R}_stats.log
cp $(ls $O/${R}_stats/*/*kernel_stats.csv | head -1) $O/${R}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats2 -- $B --no-kernel-profile > /dev/null 2> $O/${R}_stats2.log
cp $(ls $O/${R}_stats2/*/*kernel_stats.csv | head -1) $O/${R}_kernel_stats_dual_stream.csv
P="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --no-f32-leg --no-secondary --no-e2e --no-train"
SOME_AMD_DUAL_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/${R}_pmc_mfma -- $P > /dev/null 2>&1
python tools/pmc_summary.py $O/${R}_pmc_mfma > $O/${R}_pmc_mfma_busy.json
SOME_AMD_DUAL_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_pmc_fetch -- $P > /dev/null 2>&1
SOME_AMD_DUAL_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_pmc_write -- $P > /dev/null 2>&1
python tools/pmc_hbm_summary.py $O/${R}_pmc_fetch $O/${R}_pmc_write > $O/${R}_pmc_hbm_traffic.json
rm -rf $O/${R}_stats $O/${R}_stats2 $O/${R}_pmc_mfma $O/${R}_pmc_fetch $O/${R}_pmc_write
tail -c 1500 $O/${R}_bench.json
