"""Short runs of the randomised differential fuzzers (tests/fuzz/*.py) as part of the suites: the kernel sources
on the CPU emulation here, the real kernels with -m gpu.  Longer runs: `python tests/fuzz/fuzz_parity.py 2000 <seed>
[--hip]` etc."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FUZZERS = ["fuzz_parity.py", "fuzz_transforms.py", "fuzz_yuvopts.py", "fuzz_unit.py", "fuzz_filters.py", "fuzz_strip.py", "fuzz_walker.py"]


def _run(name, ncases, seed, hip):
    cmd = [sys.executable, os.path.join(HERE, "fuzz", name), str(ncases), str(seed)] + (["--hip"] if hip else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    tail = "\n".join(l for l in r.stdout.splitlines() if "MISMATCH" in l or "ERROR" in l or l.startswith("cases"))
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]


@pytest.mark.parametrize("name", FUZZERS)
def test_fuzz_on_the_emulator(dev, name):
    if dev.kind != "emu":
        pytest.skip("emulator run")
    _run(name, 80, 20260927, hip=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FUZZERS)
def test_fuzz_on_the_gpu(dev, name):
    if dev.kind != "hip":
        pytest.skip("GPU run")
    _run(name, 600, 20260928, hip=True)
