D=/tmp/some_amd_bench/hs
H="python tools/host_scaling_bench.py --dir $D --files 10000"
$H --ranks 8 --repeat 3 > gpurun_out/r04j_hs_unbound_r3.txt 2> gpurun_out/r04j_hs_build.err
$H --ranks 8 --bind --repeat 3 > gpurun_out/r04j_hs_bind_r3.txt 2>&1
$H --ranks 8 --bind --repeat 3 --io-threads 2 --align-workers 3 > gpurun_out/r04j_hs_bind_io2_align3_r3.txt 2>&1
$H --ranks 8 --bind --repeat 3 --io-threads 8 --align-workers 8 > gpurun_out/r04j_hs_bind_io8_align8_r3.txt 2>&1
$H --ranks 8 --bind > gpurun_out/r04j_hs_bind_r1.txt 2>&1
$H --ranks 8 --bind --cold > gpurun_out/r04j_hs_bind_cold.txt 2>&1
$H --ranks 4 --bind --repeat 2 > gpurun_out/r04j_hs_ranks4_bind_r2.txt 2>&1
$H --ranks 2 --bind > gpurun_out/r04j_hs_ranks2_bind.txt 2>&1
$H --ranks 1 > gpurun_out/r04j_hs_ranks1.txt 2>&1
for f in unbound_r3 bind_r3 bind_io2_align3_r3 bind_io8_align8_r3 bind_r1 bind_cold ranks4_bind_r2 ranks2_bind ranks1; do echo "== $f"; sed -n 3,3p gpurun_out/r04j_hs_$f.txt; tail -1 gpurun_out/r04j_hs_$f.txt; done
