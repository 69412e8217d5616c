D=/tmp/some_amd_bench/hs
H="python tools/host_scaling_bench.py --dir $D --files 10000"
$H --ranks 8 > gpurun_out/r04f_hs_default.txt 2> gpurun_out/r04f_hs_build.err
$H --ranks 8 --prefetch 1024 > gpurun_out/r04f_hs_prefetch1024.txt 2>&1
$H --ranks 8 --io-threads 16 --prefetch 1024 > gpurun_out/r04f_hs_io16_prefetch1024.txt 2>&1
$H --ranks 8 --bind > gpurun_out/r04f_hs_bind.txt 2>&1
$H --ranks 8 --align-workers 0 > gpurun_out/r04f_hs_noalign.txt 2>&1
$H --ranks 8 --flush-batches 2 > gpurun_out/r04f_hs_flush2.txt 2>&1
$H --ranks 8 --flush-batches 2 --prefetch 512 --bind > gpurun_out/r04f_hs_flush2_prefetch512_bind.txt 2>&1
$H --ranks 2 > gpurun_out/r04f_hs_ranks2.txt 2>&1
$H --ranks 4 > gpurun_out/r04f_hs_ranks4.txt 2>&1
for f in default prefetch1024 io16_prefetch1024 bind noalign flush2 flush2_prefetch512_bind ranks2 ranks4; do echo "== $f"; sed -n 3,4p gpurun_out/r04f_hs_$f.txt; tail -1 gpurun_out/r04f_hs_$f.txt; done
