import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    # a sanitizer build of the emulated library (GMAT_TEST_EMU_LIBRARY) reports on file descriptor 2, which pytest captures and xdist drops:
    # GMAT_SANITIZER_LOG=<prefix> sends each process's descriptor 2 to <prefix>.<pid>
    if os.environ.get("GMAT_SANITIZER_LOG"):
        os.dup2(os.open("%s.%d" % (os.environ["GMAT_SANITIZER_LOG"], os.getpid()), os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644), 2)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`python -m pytest tests -m "not gpu"` as the driver types it is ONE process: 17 minutes of kernel sources on CPU fibers (5 750 tests),
    3 on eight.  Where no GPU is visible (the build container, the driver's CPU check) and the caller chose nothing (-n, --dist, -p no:xdist,
    GMAT_TEST_SERIAL=1), the run is spread over pytest-xdist workers — what the option `-n 8` would do.  With a GPU in sight nothing is touched:
    one process per GPU."""
    opt = config.option
    # never inside a worker: xdist runs this hook there too with numprocesses reset to None — without this line every worker spread
    # itself over eight more (550 processes within 25 s of `pytest tests/test_abi.py`, the container out of memory within minutes)
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", None) is not None or getattr(opt, "dist", "no") != "no":
        return None
    if os.environ.get("GMAT_TEST_SERIAL") or os.path.exists("/dev/kfd") or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    n = min(8, os.cpu_count() or 1)
    if n > 1:
        opt.numprocesses, opt.dist, opt.tx = n, "load", ["popen"] * n
    return None


def _make(directory, target):
    """always ask make (a no-op when up to date): a stale .so must never pass for the sources"""
    import fcntl
    # one make per directory at a time: pytest-xdist workers all arrive here, and two makes rebuilding the same objects
    # truncate each other's outputs (seen as 2860 errors after a source edit once failures were no longer swallowed)
    with open(os.path.join(directory, ".make.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-C", directory], capture_output=True, text=True)
    if r.returncode != 0:               # also when an older binary exists: the suite would pass against a stale build
        raise RuntimeError(f"make -C {directory} failed:\n{r.stdout}{r.stderr}")
    return os.path.join(directory, target)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liborc.so) — the checker, never the thing under test."""
    import harness
    return harness.load_oracle(_make(os.path.join(ROOT, "oracle"), "liborc.so"))


def _gpu_available():
    try:
        from gmat_amd.lib import load
        return load().gmat_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session", params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    """A device back-end exposing the C ABI: 'hip' = the product library on a real GPU,
    'emu' = the same sources compiled against tests/hipemu (kernel logic on CPU fibers)."""
    import harness
    from gmat_amd.lib import load
    if request.param == "hip":
        if os.path.exists("/opt/rocm/bin/hipcc") and not os.environ.get("GMAT_TEST_HIP_LIBRARY"):       # keep the product library in step with its sources
            _make(os.path.join(ROOT, "gmat_amd", "csrc"), "../lib/libgmat_hip.so")
        # GMAT_TEST_HIP_LIBRARY: another BUILD of the product sources under test (tests/test_layout_variants.py: the asm-bearing kernels
        # under -O2 / -Os).  Test infrastructure; the package's own loader has no such switch.
        alt = os.environ.get("GMAT_TEST_HIP_LIBRARY")
        lib = load(alt) if alt else load()            # raises loudly if the product library is missing
        if lib.gmat_device_count() <= 0:
            pytest.fail("-m gpu test selected but no HIP device is visible")
        return harness.Dev(lib, "hip")
    # GMAT_TEST_EMU_LIBRARY: another BUILD of the emulated library (tools/ubsan_emu.sh: every kernel and the host code under
    # -fsanitize=undefined — sanitizers run on the CPU build only)
    path = os.environ.get("GMAT_TEST_EMU_LIBRARY") or _make(os.path.join(ROOT, "tests", "hipemu"), "build/libgmat_hip_emu.so")
    return harness.Dev(load(path), "emu")


@pytest.fixture(autouse=True)
def _fresh_knobs(request):
    """a test that changed a GMAT_* knob leaves the library's cached value behind (the environment itself is restored by monkeypatch):
    every test that has a device starts from the environment as it is now"""
    if "dev" in request.fixturenames:
        request.getfixturevalue("dev").lib.gmat_knobs_reload()
    yield


@pytest.fixture(params=["strip", "tiled"])
def kern(request):
    """Which 2:1 kernel a 4:2:0 -> packed RGB context gets: the strip-walking one (default) or, with
    GMAT_SCALE_NO_STRIP=1 at context creation, the tiled one it superseded (still the path of Lanczos, of filters whose
    borders are not edge replication, and of 4:2:0 destinations).  Yields the kernel name to expect."""
    old = os.environ.get("GMAT_SCALE_NO_STRIP")
    if request.param == "tiled":
        os.environ["GMAT_SCALE_NO_STRIP"] = "1"
    else:
        os.environ.pop("GMAT_SCALE_NO_STRIP", None)
    # (one frame per call: the block-cooperative form of the strip kernel; launches of more than three frames take the walker,
    # scale_yuv2s_kernel — tests/test_parity_strip.py holds both forms to the oracle at every launch size)
    yield "scale_yuv2x_kernel" if request.param == "tiled" else "scale_yuv2s_blk_kernel"
    if old is None:
        os.environ.pop("GMAT_SCALE_NO_STRIP", None)
    else:
        os.environ["GMAT_SCALE_NO_STRIP"] = old
