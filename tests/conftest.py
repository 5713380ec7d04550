import os

# errors of the HIP runtime are silent at its default log level; a run that dies should say why
os.environ.setdefault("AMD_LOG_LEVEL", "1")
# products below 8 M non-zeros split their long rows after the multiply by default; the tests' small inputs must take the
# direct-row path (osp_split.h, direct_plan_kernel) wherever it is not switched off on purpose
os.environ.setdefault("OSP_DIRECT_MIN_NNZ", "0")
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _install_abort_trace(fd):
    """SIGABRT -> native backtrace on the real stderr (tests/abort_trace.c), then the previous handler.  Best effort."""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_abort_trace.so")
    src = os.path.join(here, "abort_trace.c")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src], check=True, capture_output=True)
        ctypes.CDLL(so).abort_trace_install(int(fd))
    except Exception:  # no compiler, read-only tree ...: the tests do not depend on it
        pass


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pytest points fd 2 at a capture file while a test runs; its faulthandler plugin keeps a copy of the real stderr
    fd = None
    try:
        from _pytest.faulthandler import fault_handler_stderr_fd_key
        fd = config.stash[fault_handler_stderr_fd_key]
    except Exception:
        pass
    _install_abort_trace(fd if fd is not None else os.dup(2))
    # the in-tree library is a build product (git-ignored): build it once if a fresh checkout lacks it
    lib = os.path.join(ROOT, "outerspace_amd", "libouterspace_spgemm.so")
    cli = os.path.join(ROOT, "outerspace_amd", "osp_spgemm")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "outerspace_amd", "csrc"), "all"], check=True,
                       stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def port():
    """The plain-C oracle (checker only)."""
    from oracle import oracle
    oracle.build(ref=os.path.exists("/root/reference/simulator/SimSpGEMM.cpp"))
    return oracle.port()


@pytest.fixture(scope="session")
def _ctx_shared():
    from outerspace_amd import spgemm as S
    c = S.Context(0)
    yield c
    c.close()


_VARIANT_ENV = {"outer-split": ("OSP_DIRECT", "0"), "outer-written": ("OSP_GATHER", "0")}


@pytest.fixture(params=["outer", "rowwise", "outer-split", "outer-written"])
def ctx(request, _ctx_shared):
    """Every test that multiplies runs once per formulation: the outer product with its planned long rows GATHERED -- never
    written: the merge kernel forms their partial products from the plan's run descriptors (the default since round 5) --,
    the same with those rows written into their column ranges by the multiply phase (OSP_GATHER=0: "direct" rows as until
    round 4, still the path of rows with an over-long range), the row-wise variant that forms short rows inside the merge
    kernel (osp_config_t.algorithm; its long rows are written), and the outer product with every long row split after the
    multiply instead (OSP_DIRECT=0).  All must equal the oracle bit for bit.

    ONE library context (device 0: stream + buffer pool) for the whole session: a process normally keeps one, and a
    context per test module only meant freeing every pooled device buffer and allocating it again a moment later."""
    _ctx_shared.algorithm = "rowwise" if request.param == "rowwise" else "outer"
    env = _VARIANT_ENV.get(request.param)
    had = os.environ.get(env[0]) if env else None
    if env:
        os.environ[env[0]] = env[1]
    yield _ctx_shared
    _ctx_shared.algorithm = "outer"
    if env:
        if had is None:
            os.environ.pop(env[0], None)
        else:
            os.environ[env[0]] = had
