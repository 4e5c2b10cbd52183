"""Diagnostic: phase times of cdae_wino4_kernel (-DXSQ_WINO4_STAMPS=1 build, default) or of cdae_wino_kernel (WHICH=wn, a
-DXSQ_WINO_STAMPS=1 build), XSQ_LIB=...: per tile, in microseconds."""
import ctypes as C, os, sys
os.environ.setdefault("XSQ_WINO4", "1")
WN = os.environ.get("WHICH", "w4") == "wn"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
sep.xumx_model.set_winograd(7 if WN else 15)
sep.overlap_tail = False
x = synth_audio(4 * 2_621_440, seed=20260101).to(dev)
for _ in range(2): sep(x)
fn = _lib.lib.xsq_debug_wn_stamps if WN else _lib.lib.xsq_debug_w4_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
fn(buf, 1)
for _ in range(3): sep(x)
fn(buf, 1)
v = list(buf)
n = max(v[3], 1)
names = ["prologue", "chunk loop", "epilogue", "tiles", "  loop: at barriers (wave 0)", ]
for i, nm in enumerate(names):
    print(f"{nm:36s} {v[i] if i == 3 else v[i] * 0.01 / n:12.3f}" + ("" if i == 3 else " us / tile"))
print(f"sum per tile {sum(v[:3]) * 0.01 / n:.3f} us; x tiles / 256 CUs / 6 launches = {sum(v[:3]) * 0.01 / 256 / 6 / 1e3:.3f} ms per launch")
