"""The C ABI without Python on the call path: tools/abi_demo/demix_c.cpp -- a plain C++ host program, hipMalloc'd buffers, no
torch -- links libxumx_slicq_hip.so, uploads the plan and the seeded model through include/xumx_slicq_hip.h and demixes a
clip with ONE call (xsq_separator_forward).  Its stems must be bitwise those of the Python Separator on the same input."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from xumx_slicq_amd.synth import synth_audio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("wiener", [0, 1])
def test_c_host_program_matches_the_python_separator(tmp_path, wiener):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available on this box")
    from xumx_slicq_amd.separator import seeded_separator
    libdir = os.path.join(ROOT, "xumx_slicq_amd")
    exe = str(tmp_path / "demix_c")
    subprocess.run([HIPCC, "-O2", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "abi_demo", "demix_c.cpp"),
                    "-L", libdir, "-lxumx_slicq_hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, capture_output=True, timeout=600)
    sep = seeded_separator(realtime=False, wiener=bool(wiener))
    p = sep.nsgt.nsgt.plan
    with open(tmp_path / "plan.bin", "wb") as f:
        f.write(struct.pack("<3i", p.L, p.tr, p.nbands))
        for a, dt in ((p.Lg, "<i4"), (p.c, "<i4"), (p.g, "<f4"), (p.gd, "<f8"), (p.tw, "<f4")):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    m = sep.xumx_model
    params = m.packed_parameters()
    with open(tmp_path / "model.bin", "wb") as f:
        f.write(struct.pack("<2i", len(m.table), 0))
        f.write(np.ascontiguousarray(m._F, dtype="<i4").tobytes() + np.ascontiguousarray(m._T, dtype="<i4").tobytes())
        f.write(struct.pack("<q", params.size) + np.ascontiguousarray(params, dtype="<f4").tobytes())
    chunk, N, nb = 60000, 60000 * 3 + 12345, 2
    x = synth_audio(N, seed=123, nb_samples=nb)
    with open(tmp_path / "audio.bin", "wb") as f:
        f.write(struct.pack("<iq", nb, N) + x.numpy().astype("<f4").tobytes())
    r = subprocess.run([exe, str(tmp_path / "plan.bin"), str(tmp_path / "model.bin"), str(tmp_path / "audio.bin"),
                        str(tmp_path / "stems.bin"), str(chunk), str(wiener)], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stderr
    got = torch.from_numpy(np.fromfile(tmp_path / "stems.bin", dtype="<f4").reshape(4, nb, 2, N))
    try:
        sep.chunk_size = chunk
        want = sep(x.cuda()).cpu()
    finally:
        sep.chunk_size = 2621440
    assert torch.equal(got, want)
