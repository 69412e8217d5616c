D=/tmp/some_amd_bench/hs
H="python tools/host_scaling_bench.py --dir $D --files 10000"
$H --ranks 8 > gpurun_out/r04h_hs_unbound.txt 2> gpurun_out/r04h_hs_build.err
$H --ranks 8 --bind > gpurun_out/r04h_hs_bind.txt 2>&1
$H --ranks 8 --bind --cold > gpurun_out/r04h_hs_bind_cold.txt 2>&1
$H --ranks 8 --bind > gpurun_out/r04h_hs_bind2.txt 2>&1
$H --ranks 8 --bind --io-threads 4 > gpurun_out/r04h_hs_bind_io4.txt 2>&1
$H --ranks 8 --bind --align-workers 4 > gpurun_out/r04h_hs_bind_align4.txt 2>&1
$H --ranks 8 --bind --align-workers 4 --io-threads 4 > gpurun_out/r04h_hs_bind_align4_io4.txt 2>&1
$H --ranks 1 > gpurun_out/r04h_hs_ranks1.txt 2>&1
for f in unbound bind bind_cold bind2 bind_io4 bind_align4 bind_align4_io4 ranks1; do echo "== $f"; sed -n 3,3p gpurun_out/r04h_hs_$f.txt; tail -1 gpurun_out/r04h_hs_$f.txt; done
