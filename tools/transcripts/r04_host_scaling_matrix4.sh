D=/tmp/some_amd_bench/hs
H="python tools/host_scaling_bench.py --dir $D --files 10000"
$H --ranks 8 > gpurun_out/r04i_hs_unbound.txt 2> gpurun_out/r04i_hs_build.err
$H --ranks 8 --bind > gpurun_out/r04i_hs_bind.txt 2>&1
$H --ranks 8 --bind --io-threads 4 --align-workers 4 > gpurun_out/r04i_hs_bind_io4_align4.txt 2>&1
$H --ranks 8 --bind --io-threads 8 --align-workers 4 > gpurun_out/r04i_hs_bind_io8_align4.txt 2>&1
$H --ranks 8 --bind --io-threads 4 --align-workers 8 > gpurun_out/r04i_hs_bind_io4_align8.txt 2>&1
$H --ranks 8 --bind --io-threads 2 --align-workers 3 > gpurun_out/r04i_hs_bind_io2_align3.txt 2>&1
$H --ranks 8 --bind --io-threads 4 --align-workers 4 --cold > gpurun_out/r04i_hs_bind_io4_align4_cold.txt 2>&1
$H --ranks 1 > gpurun_out/r04i_hs_ranks1.txt 2>&1
for f in unbound bind bind_io4_align4 bind_io8_align4 bind_io4_align8 bind_io2_align3 bind_io4_align4_cold ranks1; do echo "== $f"; sed -n 3,3p gpurun_out/r04i_hs_$f.txt; tail -1 gpurun_out/r04i_hs_$f.txt; done
