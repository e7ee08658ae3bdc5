#!/bin/bash
# round 5, visit u2m: SQ counters of RGB 1080p -> 720p Lanczos, the classic 8-tile form against the shared-column form (| 0x100000)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 > $O/r05_u2m_pmc_classic.txt 2>&1
VPF_PMC_MFMA=0x100000 bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 > $O/r05_u2m_pmc_sc.txt 2>&1
paste $O/r05_u2m_pmc_classic.txt $O/r05_u2m_pmc_sc.txt | cut -c1-200
