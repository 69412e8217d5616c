#!/bin/bash
# round 3, GPU call 14: FETCH_SIZE of the FFN1 / QKV GEMMs with and without the XCD column-split tile map
O=gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 1 3; do
  SOME_AMD_GEMM_FLAGS=$f rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$f -- python tools/gemm_bench.py --iters 3 > /dev/null 2>&1
  python tools/pmc_hbm_summary.py $O/pmc_fetch_$f /nonexistent > $O/fetch_flags$f.json
  rm -rf $O/pmc_fetch_$f
done
python - <<'PY'
import json
for f in (1, 3):
    d = json.load(open(f'gpurun_out/r03p/fetch_flags{f}.json'))
    for k, v in d.items():
        if 'hgemm3' in k:
            print(f, k[:66], {c: round(x['avg_KiB'] * 2 / 1024, 1) for c, x in v.items()}, 'MiB (x2 corrected)', [x['dispatches'] for x in v.values()])
PY
