#!/bin/bash
# round 4, visit c: the latency-hiding form of the matrix-core Lanczos kernel (double-buffered stage / out tiles, row-weight preload): parity, same-box A/B against round 3's library
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or resize" > gpurun_out/r04c_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04c_pytest.txt
for rep in 1 2; do
  for lib in tools/lab/ab/libvpfhip_r03.so videoprocessingframework_amd/libvpfhip.so; do
    timeout 300 python tools/lab/ablate/time_one.py $lib 2>&1 | grep ablate | tee -a gpurun_out/r04c_ab.txt
  done
done
