"""One case of tests/fuzz/fuzz_walker.py by hand: KEY=VALUE arguments are environment knobs, SF DF SW SH DW DH ALGO NF ALIGN NS the case.
PROBE_HIP=1 runs the product library on a GPU, otherwise the CPU emulation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
for kv in sys.argv[1:]:
    k, v = kv.split("="); os.environ[k] = v
import harness
from harness import SWS
from gmat_amd.lib import load
from test_batch_api import _run_batch
E = os.environ.get
orc = harness.load_oracle(os.path.join(ROOT, "oracle", "liborc.so"))
HIP = E("PROBE_HIP") == "1"
lib = load() if HIP else load(os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so"))
dev = harness.Dev(lib, "hip" if HIP else "emu")
try:
    k = _run_batch(dev, orc, E("SF", "nv12"), E("DF", "yuv420p"), int(E("SW", 1428)), int(E("SH", 248)), int(E("DW", 272)), int(E("DH", 58)),
                   nframes=int(E("NF", 3)), nstreams=int(E("NS", 1)), align=int(E("ALIGN", 16)), flags=SWS[E("ALGO", "bicubic")])
    print("ok", k)
except AssertionError as e:
    print("MISMATCH", str(e)[:300])
