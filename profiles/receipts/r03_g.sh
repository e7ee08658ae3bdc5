#!/bin/bash
# round 3, visit g: the whole GPU suite (as the driver runs it) + smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03g_pytest.txt 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r03g_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
