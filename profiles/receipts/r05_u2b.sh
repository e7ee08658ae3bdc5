#!/bin/bash
# round 5, visit u2b: SQ counters of the Lanczos up-scale RGB 1080p -> 4K, ring of two (policy) against the ring of four (| 0x80000)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
bash tools/gpu_pmc_resize_batch.sh 1920 1080 3840 2160 2 > $O/r05_u2b_pmc_ring2.txt 2>&1
VPF_PMC_MFMA=0x80000 bash tools/gpu_pmc_resize_batch.sh 1920 1080 3840 2160 2 > $O/r05_u2b_pmc_ring4.txt 2>&1
paste $O/r05_u2b_pmc_ring2.txt $O/r05_u2b_pmc_ring4.txt | cut -c1-230
