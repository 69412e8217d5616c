#!/bin/bash
# round 6, call 11: closing evidence - the whole GPU suite, smoke(), the default bench line + rocprofv3 kernel stats + PMC passes
# (tools/collect_profiles.sh), the e2e leg with 10 000 DISTINCT files
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/r06z_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06z_smoke.txt 2>&1
bash tools/collect_profiles.sh r06z > $O/r06z_collect.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --no-f32-leg --no-fast-leg --no-secondary --no-live-pmc --no-train --e2e-distinct 10000 > $O/r06z_bench_e2e_distinct10000.json 2> $O/r06z_bench_e2e_distinct10000.err
cat $O/r06z_pytest_gpu_tail.txt $O/r06z_smoke.txt; tail -c 3000 $O/r06z_bench.json; echo; tail -c 2500 $O/r06z_bench_e2e_distinct10000.json
