"""pytest_with_lib.py LIB [pytest arguments]: the parity tests against ANOTHER build of the kernel library (capi.LIB_PATH swapped before the first
call, pytest run in this process) — e.g. the Lanczos cases against a lab variant before its timing is believed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from videoprocessingframework_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
rc = pytest.main(sys.argv[2:])
print(f"[pytest_with_lib] {os.path.basename(capi.LIB_PATH)}: exit {int(rc)}", flush=True)
sys.exit(int(rc))
