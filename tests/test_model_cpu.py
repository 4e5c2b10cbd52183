"""CPU-side checks of the model mirrors: state_dict layout identical to the
reference's (so real checkpoints drop in), packing order, loud failures."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def unmix():
    from xumx_slicq_amd.separator import build_models
    m, enc, sr = build_models(device="cpu")
    return m


def test_state_dict_layout_equals_reference(unmix):
    g = load_golden("state_dict_layout.npz")
    sd = unmix.state_dict()
    assert len(sd) == int(g["nkeys"]) == 5740
    assert list(sd.keys())[:30] == list(g["first_keys"])
    desc = "\n".join(f"{k} {tuple(v.shape)}" for k, v in sd.items())
    assert hashlib.sha256(desc.encode()).hexdigest() == str(g["sha256"])
    assert sum(v.numel() for v in sd.values() if v.dtype.is_floating_point) == int(g["nparams"])


def test_seeded_weights_follow_the_layout_and_pack(unmix, seeded_sd):
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.weights import state_dict_spec
    keys = [k for k, _, _ in state_dict_spec(unmix.table.shapes)]
    assert keys == list(unmix.state_dict().keys())
    unmix.load_state_dict(seeded_sd, strict=True)
    packed = unmix.packed_parameters()
    F = np.asarray([s[0] for s in unmix.table.shapes], dtype=np.int32)
    T = np.asarray([s[1] for s in unmix.table.shapes], dtype=np.int32)
    assert packed.dtype == np.float32
    assert packed.size == _lib.lib.xsq_model_num_params(len(F), F.ctypes.data, T.ctypes.data)
    assert sum(p.numel() for p in unmix.parameters()) == 15010446          # README: 60 MB fp32
    assert packed.size == 15010446 + 70 * 4 * 2 * (50 + 51 + 50)             # + BN running stats
    # first entries are block 0's input_mean / input_scale
    assert np.array_equal(packed[:1], seeded_sd["sliced_umx.0.input_mean"].numpy())


def test_cpu_and_training_mode_are_refused(unmix):
    from xumx_slicq_amd import _lib
    X = [torch.zeros(1, 2, F, 3, T, 2) for F, T in unmix.table.shapes]
    unmix.eval()
    with pytest.raises(_lib.XsqError):
        unmix(X)
    with pytest.raises(NotImplementedError):
        unmix.sliced_umx[0](X[0], X[0][..., 0])
    from xumx_slicq_amd.phase import blockwise_wiener
    with pytest.raises(_lib.XsqError):
        blockwise_wiener(torch.zeros(1, 2, 3, 4, 8, 2), torch.zeros(4, 1, 2, 3, 4, 8))
    with pytest.raises(ValueError):
        blockwise_wiener(torch.zeros(1, 2, 3, 4, 8, 2), torch.zeros(4, 1, 2, 3, 4, 9))


def test_separator_api_surface():
    from xumx_slicq_amd.separator import Separator, load_target_models
    with pytest.raises(ValueError):
        Separator.load(runtime_backend="tensorrt")
    with pytest.raises(ValueError):
        load_target_models("/nonexistent", runtime_backend="torch-cpu")
    assert Separator.sources == ["bass", "vocals", "other", "drums"]
    est = torch.arange(4 * 1 * 2 * 5, dtype=torch.float32).view(4, 1, 2, 5)
    d = Separator.to_dict(est)
    assert list(d) == Separator.sources and torch.equal(d["drums"], est[3])
    agg = Separator.to_dict(est, {"acc": ["bass", "other", "drums"], "v": ["vocals"]})
    assert torch.equal(agg["acc"], est[0] + est[2] + est[3])
