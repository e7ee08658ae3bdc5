#!/bin/bash
# round 3, visit ao: the weight-table cache under eight concurrent host threads
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "concurrent_threads or across_streams or arena" 2>&1 | tail -5
