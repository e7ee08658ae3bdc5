#!/bin/bash
# round 6, visit f: the product library without the forms no policy selects (persistent band launch, two-role Lanczos, per-wave fused strips -> tools/lab/libvpfhip_forms.so):
# the whole GPU suite (both libraries are loaded by it), then the fused shapes the per-wave strips used to take by policy, product vs lab build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -15) > $O/r06_f_pytest.txt; tail -6 $O/r06_f_pytest.txt
(timeout 600 python tools/fused_forms_ab.py 2>&1 | grep fused_ab) > $O/r06_f_fused_forms_ab.txt; cat $O/r06_f_fused_forms_ab.txt
