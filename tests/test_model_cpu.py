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
    with pytest.raises(_lib.XsqError):                      # the per-block call is a HIP path too (no CPU fallback)
        unmix.sliced_umx[0](X[0], X[0][..., 0])
    with pytest.raises(ValueError):                         # block 1 handed block 0's shapes
        unmix.sliced_umx[1](X[0], X[0][..., 0])
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


# the args the reference's training run stores next to the checkpoint (pretrained_model/xumx_slicq_v2.json)
REF_ARGS = {"batch_size": 64, "batch_size_valid": 1, "cuda_device": -1, "debug": False, "epochs": 1000, "fbins": 262,
            "fgamma": 15.0, "fmin": 32.9, "fscale": "bark", "lr": 0.001, "lr_decay_gamma": 0.3, "lr_decay_patience": 80,
            "nb_channels": 2, "nb_workers": 8, "patience": 1000, "quiet": False, "realtime": False,
            "samples_per_track": 64, "sample_rate": 44100.0, "seed": 42, "seq_dur": 2.0, "weight_decay": 1e-05}


def write_checkpoint(path, state, realtime=False):
    """A model directory as the reference's training writes it (training.py:419-448): xumx_slicq_v2.pth = the
    state_dict, xumx_slicq_v2.json = {"args": ..., histories}."""
    import json
    torch.save(state, path / "xumx_slicq_v2.pth")
    (path / "xumx_slicq_v2.json").write_text(json.dumps({"args": dict(REF_ARGS, realtime=realtime), "best_epoch": 1,
                                                         "best_loss": 0.1, "epochs_trained": 1, "num_bad_epochs": 0,
                                                         "train_loss_history": [0.1], "train_time_history": [1.0],
                                                         "valid_loss_history": [0.1]}))


def test_separator_load_reads_a_reference_style_checkpoint(tmp_path, seeded_sd):
    """The drop-in entry inference.py uses: Separator.load(model_path=...) (separator.py:50-93, 262-356)."""
    from xumx_slicq_amd.separator import Separator
    write_checkpoint(tmp_path, seeded_sd, realtime=False)
    sep = Separator.load(model_path=str(tmp_path), device="cpu")
    assert sep.runtime_backend == "hip-rocm" and sep.chunk_size == 2621440 and float(sep.sample_rate) == 44100.0
    assert sep.nsgt.nsgt.sllen == 18060 and not sep.xumx_model.training
    got = sep.xumx_model.state_dict()
    assert list(got.keys()) == list(seeded_sd.keys())
    assert all(torch.equal(got[k], seeded_sd[k]) for k in got)
    assert all(not blk.realtime and not blk.causal for blk in sep.xumx_model.sliced_umx)
    # the realtime flag comes from the JSON (separator.py:330-345), causal first layers included
    write_checkpoint(tmp_path, seeded_sd, realtime=True)
    rt = Separator.load(model_path=str(tmp_path), device="cpu")
    assert all(blk.realtime and blk.causal for blk in rt.xumx_model.sliced_umx)


def test_separator_load_failures_are_loud(tmp_path, seeded_sd):
    from xumx_slicq_amd.separator import Separator
    with pytest.raises(AssertionError):                     # no JSON next to the weights (separator.py:276)
        Separator.load(model_path=str(tmp_path), device="cpu")
    write_checkpoint(tmp_path, seeded_sd)
    # a Git-LFS pointer instead of the weights (what /root/reference/pretrained_model holds offline)
    (tmp_path / "xumx_slicq_v2.pth").write_text("version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 60123456\n")
    with pytest.raises(RuntimeError, match="LFS"):
        Separator.load(model_path=str(tmp_path), device="cpu")
    # a truncated checkpoint
    blob = tmp_path / "full.pth"
    torch.save(seeded_sd, blob)
    (tmp_path / "xumx_slicq_v2.pth").write_bytes(blob.read_bytes()[:1_000_000])
    with pytest.raises(Exception):
        Separator.load(model_path=str(tmp_path), device="cpu")
    # a checkpoint of another architecture: strict loading reports it (the reference drops it silently, quirk A7)
    bad = dict(seeded_sd)
    bad.pop("sliced_umx.3.input_mean")
    torch.save(bad, tmp_path / "xumx_slicq_v2.pth")
    with pytest.raises(RuntimeError, match="input_mean"):
        Separator.load(model_path=str(tmp_path), device="cpu")


def test_copies_and_mode_switches_do_not_share_or_rebuild_device_state(unmix):
    import copy
    unmix.eval()
    v = unmix._version()
    unmix.eval()                                            # idempotent: no repack
    assert unmix._version() == v
    unmix.train()
    assert unmix._version() == v + 1
    unmix.eval()
    unmix._handles[7] = (0, "raw-handle")                   # stand-ins for ctypes handles / workspaces
    unmix._ws[(7, 0)] = "workspace"
    try:
        c = copy.deepcopy(unmix)
        assert c._handles == {} and c._ws == {} and c is not unmix
        assert all(torch.equal(a, b) for a, b in zip(c.state_dict().values(), unmix.state_dict().values()))
    finally:
        unmix._handles.pop(7), unmix._ws.pop((7, 0))
