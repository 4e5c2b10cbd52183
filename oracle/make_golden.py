"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from
/root/reference, never copied) on seeded inputs.  Development container only:
the reference does not travel to the GPU box; the fixtures do.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Rules: torch.no_grad(); clone block lists before decoding
(the reference decoder overwrites its input, quirk A1); clone X before the
phasemix path (quirk A2); fixed thread count.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from xumx_slicq_amd.synth import synth_audio
from xumx_slicq_amd.weights import seeded_state_dict

OUT = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 1234


def checksums(t: torch.Tensor):
    a = t.detach().double().flatten()
    return np.array([float(a.sum()), float((a * a).sum()), float(a.abs().max())])


def main():
    torch.set_num_threads(8)
    from xumx_slicq_v2.transforms import NSGTBase, make_filterbanks, ComplexNorm
    from xumx_slicq_v2.model import Unmix
    from xumx_slicq_v2.phase import blockwise_wiener
    from xumx_slicq_v2.separator import Separator

    os.makedirs(OUT, exist_ok=True)
    base = NSGTBase("bark", 262, 32.9, fs=44100.0, device="cpu")
    enc, dec = make_filterbanks(base, 44100.0)
    ns = base.nsgt
    nb = ns.fbins_actual

    # ---- (1) plan ---------------------------------------------------------
    Lg = np.array([len(g) for g in ns.g[:nb]], dtype=np.int32)
    assert np.array_equal(Lg, np.array([int(m) for m in ns.M[:nb]]))
    c = np.array([int(w[len(w) // 2]) for w in ns.wins[:nb]], dtype=np.int32)  # centre bin mod L
    from xumx_slicq_v2.nsgt.slicing import makewnd
    tw = makewnd(base.sllen, base.trlen).numpy()
    with torch.no_grad():
        jag, _ = base.predict_input_size(1, 2, 2.0)
    blocks = np.array([[b.shape[2], b.shape[4]] for b in jag], dtype=np.int32)
    np.savez_compressed(
        os.path.join(OUT, "plan.npz"),
        L=base.sllen, tr=base.trlen, nbands=nb, Lg=Lg, c=c, blocks=blocks, tw=tw,
        g=np.concatenate([g.numpy() for g in ns.g[:nb]]).astype(np.float32),
        gd=np.concatenate([g.numpy() for g in ns.gd[:nb]]).astype(np.float64),
        seq_dur_slices=jag[0].shape[3],
    )

    # ---- (2)+(3) forward / inverse ---------------------------------------
    keep_blocks = [0, 1, 2, 4, 33, 69]
    for n in (9031, 70000):
        x = synth_audio(n, seed=20260101 + n)
        with torch.no_grad():
            C = enc(x)
            rng = np.random.default_rng(n)
            P = [cb + torch.from_numpy(
                (0.1 * rng.standard_normal(cb.shape)).astype(np.float32)) for cb in C]
            y = dec([p.clone() for p in P], n)
        d = dict(n=n, S=C[0].shape[3],
                 fwd_sums=np.stack([checksums(cb) for cb in C]),
                 pert_sums=np.stack([checksums(p) for p in P]),
                 inv=y.numpy())
        if n == 9031:
            for i, cb in enumerate(C):
                d[f"fwd_{i}"] = cb.numpy()
        else:
            for i in keep_blocks:
                d[f"fwd_{i}"] = C[i].numpy()
        np.savez_compressed(os.path.join(OUT, f"slicqt_{n}.npz"), **d)

    # ---- model with seeded weights -----------------------------------------
    sd = seeded_state_dict([tuple(b) for b in blocks.tolist()], seed=WEIGHT_SEED)
    cnorm = ComplexNorm()

    def build(realtime_conv: bool, phasemix: bool):
        m = Unmix(cnorm(jag), realtime=realtime_conv)
        missing, unexpected = m.load_state_dict(sd, strict=True), None
        m.freeze()
        for blk in m.sliced_umx:         # flag read at model.py:264
            blk.realtime = phasemix
        return m

    import hashlib
    ref_sd = build(False, False).state_dict()
    desc = "\n".join(f"{k} {tuple(v.shape)}" for k, v in ref_sd.items())
    np.savez_compressed(os.path.join(OUT, "state_dict_layout.npz"),
                        nkeys=len(ref_sd), nparams=sum(v.numel() for v in ref_sd.values() if v.dtype.is_floating_point),
                        sha256=hashlib.sha256(desc.encode()).hexdigest(),
                        first_keys=np.array(list(ref_sd.keys())[:30]))

    models = {
        "realtime": build(True, True),       # config 1: causal conv + phasemix
        "offline_phasemix": build(False, True),   # config 2: offline conv + phasemix
        "offline_wiener": build(False, False),    # config 3: offline conv + Wiener-EM
    }

    # ---- (4) per-block CDAE masks ------------------------------------------
    n = 70000
    x = synth_audio(n, seed=20260101 + n)
    d = dict(n=n, blocks=np.array(keep_blocks))
    with torch.no_grad():
        C = enc(x)
        for name in ("realtime", "offline_wiener"):
            _, masks = models[name]([cb.clone() for cb in C], return_masks=True)
            for i in keep_blocks:
                d[f"mask_{'causal' if name == 'realtime' else 'offline'}_{i}"] = masks[i].numpy()
            d[f"mask_sums_{'causal' if name == 'realtime' else 'offline'}"] = np.stack(
                [checksums(m) for m in masks])
    np.savez_compressed(os.path.join(OUT, "cdae_masks_70000.npz"), **d)

    # ---- (5) blockwise_wiener ----------------------------------------------
    rng = np.random.default_rng(5)
    mix = torch.from_numpy(rng.standard_normal((1, 2, 2, 26, 200, 2)).astype(np.float32))
    mag = torch.from_numpy(np.abs(rng.standard_normal((4, 1, 2, 2, 26, 200))).astype(np.float32))
    with torch.no_grad():
        yw = blockwise_wiener(mix.clone(), mag.clone(), 5000)
    rng = np.random.default_rng(6)   # the reference's own test shape, tests/test_phase.py:6-12
    mix2 = torch.from_numpy(rng.standard_normal((1, 2, 14, 257, 37, 2)).astype(np.float32))
    mag2 = torch.from_numpy(rng.standard_normal((4, 1, 2, 14, 257, 37)).astype(np.float32))
    with torch.no_grad():
        yw2 = blockwise_wiener(mix2.clone(), mag2.clone(), 5000)
    assert yw2.shape == (4, 1, 2, 14, 257, 37, 2) and torch.all(torch.isfinite(yw2))
    np.savez_compressed(os.path.join(OUT, "wiener.npz"),
                        out_5200=yw.numpy(),
                        out_testphase_sums=checksums(yw2),
                        out_testphase_sub=yw2.flatten()[::97].numpy())

    # ---- (6) end-to-end stems ----------------------------------------------
    for n in (9031, 100000):
        x = synth_audio(n, seed=20260101 + n)
        d = dict(n=n)
        for name, m in models.items():
            sep = Separator(xumx_model=m, encoder=(enc, dec, cnorm), runtime_backend="torch-cpu",
                            chunk_size=2621440 if n < 50000 else 60000, quiet=True)
            sep.freeze()
            with torch.no_grad():
                est = sep(x.clone())
            assert est.shape == (4, 1, 2, n)
            d[f"{name}_sums"] = np.stack([checksums(est[t]) for t in range(4)])
            d[f"{name}"] = est.numpy() if n == 9031 else est[..., ::7].contiguous().numpy()
        d["chunk_size"] = 2621440 if n < 50000 else 60000
        np.savez_compressed(os.path.join(OUT, f"stems_{n}.npz"), **d)

    # ---- (7) validation half of one training.loop step (training.py:66-103, train=False, SDR term off):
    #      B = 2 clips of 2 s, mix = sum of four seeded sources, offline model in eval mode
    import types
    sys.modules.setdefault("auraloss", types.SimpleNamespace(time=types.SimpleNamespace(SDSDRLoss=lambda: None)))
    from xumx_slicq_v2.loss import ComplexMSELossCriterion, MaskSumLossCriterion
    n = 88200
    y_t = torch.stack([0.5 * synth_audio(n, seed=500 + j, nb_samples=2) for j in range(4)])      # (4, 2, 2, n)
    x = y_t.sum(0)
    with torch.no_grad():
        Xc = enc(x)
        Yest, Ymask = models["offline_wiener"]([c.clone() for c in Xc], return_masks=True)
        Ytgt = enc(y_t)
        mse = float(ComplexMSELossCriterion()(Yest, Ytgt))
        msk = float(MaskSumLossCriterion()(Ymask))
    np.savez_compressed(os.path.join(OUT, "validation_step.npz"), n=n, mse=mse, mask=msk, loss=mse + msk)
    print("validation step: mse", mse, "mask", msk)

    # ---- (7b) one TRAINING step of the reference (training.py:66-108): unmix.train(), loss.backward().
    #       B = 2 clips of 1 s (S = 6), realtime (causal + mix-phase) and offline (Wiener-EM) models.
    n = 44100
    y_t = torch.stack([0.5 * synth_audio(n, seed=600 + j, nb_samples=2) for j in range(4)])
    x = y_t.sum(0)
    d = dict(n=n)
    keep_keys = ["sliced_umx.0.input_mean", "sliced_umx.0.input_scale", "sliced_umx.0.cdaes.1.0.weight",
                 "sliced_umx.0.cdaes.1.1.weight", "sliced_umx.0.cdaes.1.1.bias", "sliced_umx.0.cdaes.2.3.weight",
                 "sliced_umx.0.cdaes.2.6.weight", "sliced_umx.0.cdaes.3.9.weight", "sliced_umx.0.cdaes.3.9.bias",
                 "sliced_umx.1.cdaes.0.3.weight", "sliced_umx.2.cdaes.1.6.weight", "sliced_umx.33.cdaes.2.0.weight",
                 "sliced_umx.69.cdaes.3.9.weight", "sliced_umx.1.input_scale", "sliced_umx.69.cdaes.0.7.bias"]
    for tag, rt in (("realtime", True), ("offline", False)):
        m = Unmix(cnorm(jag), realtime=rt)
        m.load_state_dict(sd, strict=True)
        m.train()
        Xc = enc(x)
        Yest, Ymask = m([c.clone() for c in Xc], return_masks=True)
        with torch.no_grad():
            Ytgt = enc(y_t)
        mse = ComplexMSELossCriterion()(Yest, Ytgt)
        msk = MaskSumLossCriterion()(Ymask)
        (mse + msk).backward()
        d[f"{tag}_mse"], d[f"{tag}_mask"] = float(mse), float(msk)
        names, norms = [], []
        for k, p_ in m.named_parameters():
            names.append(k); norms.append(float(p_.grad.double().norm()))
        d[f"{tag}_grad_norms"] = np.array(norms)
        d["param_names"] = np.array(names)
        for k in keep_keys:
            d[f"{tag}_grad::{k}"] = dict(m.named_parameters())[k].grad.numpy()
        rm = dict(m.named_buffers())
        d[f"{tag}_running_mean::sliced_umx.1.cdaes.0.4"] = rm["sliced_umx.1.cdaes.0.4.running_mean"].numpy()
        d[f"{tag}_running_var::sliced_umx.1.cdaes.0.4"] = rm["sliced_umx.1.cdaes.0.4.running_var"].numpy()
        print("training step", tag, "mse", float(mse), "mask", float(msk))
    np.savez_compressed(os.path.join(OUT, "training_step.npz"), **d)

    # ---- (9) dataset statistics: the reference's training.get_statistics on three seeded "tracks"
    # training.py pulls in packages that are absent offline; none of them is touched by get_statistics
    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Stub(self.__name__ + "." + k)
        def __call__(self, *a, **k):
            return None
    for name in ("torchaudio", "torchaudio.transforms", "torchinfo", "torch.utils.tensorboard", "tensorboard",
                 "musdb", "museval"):
        sys.modules.setdefault(name, _Stub(name))
    import importlib
    try:
        tr = importlib.import_module("xumx_slicq_v2.training")
        have_training = True
    except Exception as e:  # training.py drags in data.py / torchaudio / torchinfo / tensorboard
        have_training = False
        print("training.py not importable here (", type(e).__name__, e, "): restating get_statistics' loop inline")
    lens = (50000, 80000, 30000)

    class _DS:
        random_chunks = True; seq_duration = 1.0; samples_per_track = 2; augmentations = 1
        random_track_mix = True; random_interferer_mix = True
        def __len__(self): return len(lens)
        def __getitem__(self, i): return synth_audio(lens[i], seed=900 + i)[0], None

    if have_training:
        means, stds = tr.get_statistics(types.SimpleNamespace(quiet=True), (enc, dec, cnorm), _DS(), len(jag))
    else:
        import sklearn.preprocessing
        scalers = [sklearn.preprocessing.StandardScaler() for _ in jag]
        for i in range(len(lens)):
            X = cnorm(enc(_DS()[i][0][None, ...]))
            for k, Xb in enumerate(X):       # training.py:141-149, verbatim semantics
                flat = np.squeeze(torch.flatten(Xb, start_dim=-2, end_dim=-1).mean(1, keepdim=False).permute(0, 2, 1), axis=0)
                scalers[k].partial_fit(flat)
        stds = [np.maximum(sc.scale_, 1e-4 * np.max(sc.scale_)) for sc in scalers]
        means = [sc.mean_ for sc in scalers]
    np.savez_compressed(os.path.join(OUT, "statistics.npz"), lens=np.array(lens), via_reference_function=have_training,
                        means=np.concatenate(means), stds=np.concatenate(stds))

    # ---- (8) second plan: Mel-32 (the reference's small streaming models,
    #      .github/pretrained_models_other/*/xumx_slicq_v2.json: fscale mel, fbins 32, fmin 115.5),
    #      one demixui-sized chunk of next_pow2(sllen) = 32768 samples (demixui.py:49-51)
    mbase = NSGTBase("mel", 32, 115.5, fs=44100.0, device="cpu")
    menc, mdec = make_filterbanks(mbase, 44100.0)
    mns = mbase.nsgt
    mnb = mns.fbins_actual
    n = 32768
    x = synth_audio(n, seed=20260101 + n)
    with torch.no_grad():
        C = menc(x)
        rng = np.random.default_rng(n)
        P = [cb + torch.from_numpy((0.1 * rng.standard_normal(cb.shape)).astype(np.float32)) for cb in C]
        y = mdec([p.clone() for p in P], n)
    d = dict(L=mbase.sllen, tr=mbase.trlen, nbands=mnb, n=n, S=C[0].shape[3],
             Lg=np.array([len(g) for g in mns.g[:mnb]], dtype=np.int32),
             c=np.array([int(w[len(w) // 2]) for w in mns.wins[:mnb]], dtype=np.int32),
             blocks=np.array([[b.shape[2], b.shape[4]] for b in C], dtype=np.int32),
             g=np.concatenate([g.numpy() for g in mns.g[:mnb]]).astype(np.float32),
             gd=np.concatenate([g.numpy() for g in mns.gd[:mnb]]).astype(np.float64),
             inv=y.numpy())
    for i, cb in enumerate(C):
        d[f"fwd_{i}"] = cb.numpy()
    np.savez_compressed(os.path.join(OUT, "mel32_32768.npz"), **d)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
