#!/bin/bash
# round 3, visit be: whole GPU suite + smoke on the final code
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r03be_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r03be_smoke.txt
