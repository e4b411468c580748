"""Layout robustness of the kernels that issue memory instructions from inline assembly: the compiler cannot
check hazards inside an asm string, so what surrounds it — schedule, register allocation, code size — decides whether a latent one
bites (round 3: a GPU fault one rebuild away, FINDINGS.md R3-walker-bands).  tools/build_layout_variants.sh rebuilds those kernel files
under -O2 and -Os (__graft_entry__.build() runs it); here every variant is (a) linted at the ISA level in the build container and
(b) run over the walkers' parity tests on the GPU, as a member of `-m gpu`."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "layout_*", "libgmat_hip.so")))
SUBSET = ["tests/test_parity_generic_walker.py", "tests/test_parity_quad_walker.py", "tests/test_parity_down3.py", "tests/test_parity_down32.py",
          "tests/test_parity_scale.py::test_yuv_single_context_bicubic"]


def test_layout_variants_exist_where_they_can_be_built():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    assert len(VARIANTS) >= 2, "run tools/build_layout_variants.sh (__graft_entry__.build() does)"
    newest = max(os.path.getmtime(f) for f in glob.glob(os.path.join(ROOT, "gmat_amd", "csrc", "*")) if os.path.isfile(f))
    for v in VARIANTS:                                  # a variant older than the sources tests yesterday's kernels
        assert os.path.getmtime(v) >= newest, "%s is older than gmat_amd/csrc: run tools/build_layout_variants.sh" % v


@pytest.mark.parametrize("lib", VARIANTS, ids=lambda p: os.path.basename(os.path.dirname(p)))
def test_layout_variant_lints_clean(lib):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    bad = m.lint_file(lib)
    assert not bad, bad[:6]


@pytest.mark.gpu
@pytest.mark.parametrize("lib", VARIANTS, ids=lambda p: os.path.basename(os.path.dirname(p)))
def test_layout_variant_passes_the_walkers_on_the_gpu(lib):
    """the band walker, the 3:1 / 3:2 plane walkers and the tiled 2:1 kernel (the four files with asm stores) through their parity
    tests against this build of them"""
    env = dict(os.environ, GMAT_TEST_HIP_LIBRARY=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + SUBSET,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
