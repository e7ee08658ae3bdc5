#!/bin/bash
# round 4, visit a: first contact of the re-scheduled matrix-core Lanczos kernel (instruction-level MFMA / VALU interleave, buffer-descriptor fetch, 8-lane staging rows):
# parity, MFMA issue probes (K = 32 form, interleaved streams), same-box A/B against round 3's library
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or resize" > gpurun_out/r04a_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04a_pytest.txt
timeout 120 tools/lab/probes/probe_mfma_rate > gpurun_out/r04a_probe_mfma_rate.txt 2>&1; cat gpurun_out/r04a_probe_mfma_rate.txt
for rep in 1 2; do
  for lib in tools/lab/ab/libvpfhip_r03.so videoprocessingframework_amd/libvpfhip.so; do
    timeout 300 python tools/lab/ablate/time_one.py $lib 2>&1 | grep ablate | tee -a gpurun_out/r04a_ab.txt
  done
done
