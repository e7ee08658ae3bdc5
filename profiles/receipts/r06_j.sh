cd $GRAFT_REPO_ROOT && export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "row_band_kernels" 2>&1 | tail -5
SWEEP_CASES="RGB:1920x1080:1280x720,RGB:3840x2160:2560x1440,RGB:3840x2160:1920x1081,RGB:1280x720:1920x1080,RGB:1920x1080:3840x2160" SWEEP_PASSES=5 timeout 900 python tools/band_knob_sweep.py 0 0x100100 0x100200 0x100400 0x100800 0x100208 0x100408 2>&1 | grep knobs | tee gpurun_out/r06_j_stream_knobs.txt
SWEEP_N=128 SWEEP_CASES="RGB:1920x1080:1280x720,RGB:1280x720:1920x1080" SWEEP_PASSES=5 timeout 900 python tools/band_knob_sweep.py 0 0x100200 0x100400 0x100800 2>&1 | grep knobs | tee gpurun_out/r06_j_stream_knobs_n128.txt
