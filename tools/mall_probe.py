"""Does the inverse transform speed up when its intermediates fit the 256 MB Infinity Cache?
Per-kernel microseconds per (channel, slice) row for several batch sizes."""
import os, sys, json
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.transforms import NSGTBase

with redirect_stdout(sys.stderr):
    base = NSGTBase("bark", 262, 32.9, device="cuda")
eng = base.nsgt
for BC, S in [(8, 16), (8, 32), (8, 64), (8, 128), (8, 292), (32, 292)]:
    n = eng.ncoefs if hasattr(eng, "ncoefs") else None
    arena = torch.randn(2 * BC * S * int(_lib.lib.xsq_plan_coefs_per_slice(eng.handle(torch.device("cuda")))), device="cuda")
    length = (2 * S - 2) * 4515
    for _ in range(2):
        eng.backward(arena, BC, S, length)
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_reset()
    for _ in range(5):
        eng.backward(arena, BC, S, length)
    torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    rows = BC * S
    print(json.dumps({"BC": BC, "S": S, "rows": rows, "Z_MB": round(rows * 18640 * 8 / 2**20),
                      "us_per_row": {k: round(ms / 5 / rows * 1e3, 4) for k, (ms, c) in prof.items()},
                      "ms": {k: round(ms / 5, 3) for k, (ms, c) in prof.items()}}))
    del arena
