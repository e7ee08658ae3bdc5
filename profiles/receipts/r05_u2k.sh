cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD SWEEP_INTERP=2 SWEEP_N=32 SWEEP_CASES="RGB:1920x1080:1280x720,Y:1920x1080:1280x720,NV12:1920x1080:1280x720,RGB:3840x2160:1920x1080,RGB:1920x1080:3840x2160"
for pad in 0 14000 0 14000; do echo "== LDS pad $pad (4-tile strips: 39.5 KB -> three per CU; + 14000 -> 53.5 KB: two per CU)"; VPF_LZM_LDS_PAD_EXPERIMENT=$pad timeout 300 python tools/band_knob_sweep.py 0x400 0x800 2>&1 | grep knobs | tail -3; done > $O/r05_u2k_occupancy_experiment.txt; cat $O/r05_u2k_occupancy_experiment.txt
