#!/bin/bash
# round 3, visit ax: tile-shape sweep of the tiled Lanczos kernel on strong down-scales (batched / per frame)
mkdir -p gpurun_out
timeout 900 python tools/lanczos_tile_sweep.py 2>&1 | grep lz-tile-sweep | tee gpurun_out/r03ax_tile_sweep.txt
