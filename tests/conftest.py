import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def native_test_build():
    """(extra compiler flags, build directory) for the C / C++ harnesses the CPU tests compile (tests/c/*, oracle/): `make sanitize`
    (tools/sanitize.sh) sets VPF_TEST_CFLAGS to the sanitizer flags and VPF_TEST_BUILD_TAG so that instrumented objects never mix with
    the plain ones."""
    import shlex

    flags = shlex.split(os.environ.get("VPF_TEST_CFLAGS", ""))
    tag = os.environ.get("VPF_TEST_BUILD_TAG", "")
    out = os.path.join(ROOT, "tests", "_build", tag) if tag else os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    return flags, out


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test may hang a run: where pytest-timeout is installed (it is in the project image) every test without a timeout of its own
    gets 20 minutes — far above the slowest (the pin-kit rehearsal, which builds and runs a second pytest), far below a stuck GPU call."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1200))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/): test infrastructure only."""
    import oracle as o

    o.lib()
    return o


@pytest.fixture(scope="session")
def capi_forms():
    """The same ctypes binding over tools/lab/libvpfhip_forms.so: libvpfhip's sources compiled with -DVPF_LAB_FORMS (tools/lab/build_lab.py), i.e.
    WITH the kernel forms no policy selects — the persistent band launch, the two-role Lanczos form, the fused kernel's per-wave strips — and the
    knob values that force them.  The product library contains none of them (tests/test_product_kernels_cpu.py)."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools", "lab"))
    import build_lab

    so = build_lab.FORMS_OUT
    if not os.path.exists(so):
        so = build_lab.build_forms()
    import videoprocessingframework_amd  # noqa: F401  (the package the copy's relative imports resolve in)
    spec = importlib.util.spec_from_file_location("videoprocessingframework_amd._capi_forms", os.path.join(root, "videoprocessingframework_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    m.LIB_PATH = so
    m.lib()
    return m


@pytest.fixture(scope="session")
def capi():
    from videoprocessingframework_amd import capi as c

    c.lib()
    return c
