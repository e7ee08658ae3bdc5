#!/bin/bash
# round 4, visit o: the two-role Lanczos kernel (pass 1 and pass 2 on different waves, three workgroups per CU) against the one-role kernel, band heights swept
mkdir -p gpurun_out
VPF_HIP_LOG=2 VPF_BENCH_MFMA=0x20004 VPF_BENCH_ONLY=lanczos timeout 120 python tools/resize_batch_bench.py 2>&1 | grep -m3 "launch k_lanczos"
for m in 0 0x20002 0x20003 0x20004 0x20005 0x20006 0x20008 0x2000c 0x20010 0x20017; do
  echo "knob $m" | tee -a gpurun_out/r04o_pair_sweep.txt
  VPF_BENCH_MFMA=$m VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | grep -E "RGB    (1920x1080->1280x720|3840x2160->1920|1280x720->1920|1920x1080->3840)|NV12   1920|YUV420 1920" | cut -c1-150 | tee -a gpurun_out/r04o_pair_sweep.txt
done
