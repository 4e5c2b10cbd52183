"""Training step (BASELINE config 5 shape family): loss + gradients of
training.loop (training.py:66-108) vs fixtures produced by the reference's own autograd
(tests/golden/training_step.npz: B = 2 clips of 1 s, realtime and offline models)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from xumx_slicq_amd.synth import synth_audio


def _inputs(n):
    y_t = torch.stack([0.5 * synth_audio(n, seed=600 + j, nb_samples=2) for j in range(4)])
    return y_t.sum(0), y_t


# Each fixture has ONE pre-activation of a small block sitting ON a ReLU kink: realtime model, block 32 / target 1,
# BatchNorm-2 output -1.28e-6 in the reference's fp32 and -4.7e-7 in fp64; offline model, block 36 / target 0,
# BatchNorm-3 output 5.5e-8 in fp32 and 1.2e-7 in fp64 (test_fixture_has_a_relu_kink pins both).  Any other summation
# order can land on the other side, which switches one of that group's 36-48 rows on or off: the gradients upstream of
# that ReLU (the group's earlier layers and the block's whitening) then move by a few percent -- a subgradient choice,
# not an error.  Those tensors get a loose bound for the HIP path; every other tensor keeps rtol.
KINKS = {
    "realtime": (32, 1, 2, ("sliced_umx.32.input_",) + tuple(f"sliced_umx.32.cdaes.1.{i}." for i in (0, 1, 3, 4))),
    "offline": (36, 0, 3, ("sliced_umx.36.input_",) + tuple(f"sliced_umx.36.cdaes.0.{i}." for i in (0, 1, 3, 4, 6, 7))),
}


def _inputs_b(g):
    n, seed0 = int(g["n"]), int(g["seed0"])
    y_t = torch.stack([0.5 * synth_audio(n, seed=seed0 + j, nb_samples=2) for j in range(4)])
    return y_t.sum(0), y_t


def _loose_b(g, tag):
    """Fixture B (oracle/make_golden_train2.py) stores min |BatchNorm output| per group; tensors of a (block, target)
    with a pre-activation within RISK of a ReLU kink -- and that block's whitening -- may legitimately land on the
    other subgradient.  By construction none of them is in a block fixture A treats loosely."""
    groups = [str(x) for x in g[f"{tag}_risk_groups"] if str(x)]
    pref = []
    for grp in groups:                       # "sliced_umx.<b>.cdaes.<t>"
        pref.append(grp + ".")
        pref.append(grp.split(".cdaes.")[0] + ".input_")
    return tuple(pref)


def _check(g, tag, mse, msk, grads, rtol, kink_rtol=None, loose=None):
    KINK = KINKS[tag][3] if loose is None else loose
    assert abs(mse - float(g[f"{tag}_mse"])) < 1e-4 * float(g[f"{tag}_mse"])
    assert abs(msk - float(g[f"{tag}_mask"])) < 1e-4 * float(g[f"{tag}_mask"])
    names = [str(k) for k in g["param_names"]]
    norms = dict(zip(names, g[f"{tag}_grad_norms"]))
    worst = 0.0
    for k in names:
        got = float(grads[k].double().norm())
        tol = kink_rtol if (kink_rtol and k.startswith(KINK)) else rtol
        # (absolute term: a few tensors have gradient norms of ~4e-5, where the transforms' own rounding -- 1e-7 of the coefficients --
        #  moves the norm by ~3e-7; the pair-contracted band DFTs of round 5 shifted one of them by 3.4e-7)
        assert abs(got - norms[k]) <= tol * norms[k] + 6e-7, (k, got, norms[k])
    for key in g.files:
        if key.startswith(f"{tag}_grad::"):
            k = key.split("::", 1)[1]
            ref = torch.from_numpy(g[key])
            err = float((grads[k].cpu() - ref).abs().max())
            scale = float(ref.abs().max()) + 1e-12
            if kink_rtol and k.startswith(KINK):
                assert err <= kink_rtol * scale + 2e-7, (k, err, scale)
                continue
            if scale > 1e-6:
                worst = max(worst, err / scale)
            # (with batch-statistics BN the loss is invariant to input_scale of single-bin blocks: those
            #  gradients are pure rounding noise around 1e-9, hence the absolute floor)
            assert err <= rtol * scale + 2e-7, (k, err, scale)
    return worst


@pytest.mark.parametrize("tag,causal,wiener", [("realtime", True, False), ("offline", False, True)])
def test_oracle_training_gradients_match_reference(oracle_plan, seeded_sd, tag, causal, wiener):
    from oracle import loss as oloss
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    loss, mse, msk, grads = oloss.training_gradients(oracle_plan, seeded_sd, x, y_t, causal=causal, wiener=wiener)
    _check(g, tag, mse, msk, grads, rtol=2e-3)


@pytest.mark.parametrize("tag,causal", [("realtime", True), ("offline", False)])
def test_fixture_has_a_relu_kink(oracle_plan, seeded_sd, tag, causal):
    """Pins the premise of KINKS above on the oracle (one block, fp32 and fp64)."""
    import torch.nn.functional as F
    from oracle import model as omodel
    from oracle import slicqt as oslicqt
    g = load_golden("training_step.npz")
    x, _ = _inputs(int(g["n"]))
    b, t, layer, _ = KINKS[tag]
    pre_b, p = f"sliced_umx.{b}.", f"sliced_umx.{b}.cdaes.{t}."
    Xb = oslicqt.forward(oracle_plan, x)[b]
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt) for k, v in seeded_sd.items() if k.startswith(pre_b) and v.dtype.is_floating_point}
        mag = omodel.abs_of_real_complex(Xb.to(dt))
        B, C, Fb, S, T = mag.shape
        xx = (mag.reshape(B, C, Fb, S * T) + sd[pre_b + "input_mean"][None, None, :, None]) \
            * sd[pre_b + "input_scale"][None, None, :, None]
        y = F.conv2d(F.pad(xx, (T - 1, 0)) if causal else xx, sd[p + "0.weight"], stride=(1, T // 2))
        y = omodel._bn(F.conv2d(F.relu(omodel._bn(y, sd, p + "1", True)), sd[p + "3.weight"]), sd, p + "4", True)
        if layer == 3:
            y = omodel._bn(F.conv_transpose2d(F.relu(y), sd[p + "6.weight"]), sd, p + "7", True)
        pre = y.abs().flatten().sort().values
        assert pre[0] < 5e-6 and pre[1] > 1e-4, pre[:3]


def _trainer(realtime, precision="fp32"):
    from xumx_slicq_amd.separator import seeded_separator
    from xumx_slicq_amd.training import Trainer
    sep = seeded_separator(realtime=realtime)
    return sep, Trainer(sep.xumx_model, (sep.nsgt, sep.insgt, sep.cnorm), precision=precision)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,realtime", [("realtime", True), ("offline", False)])
def test_hip_training_gradients_match_reference(tag, realtime):
    """xsq_train_step (gradients only) vs the reference's loss.backward(): loss terms, the gradient norm of
    every one of the 3500 trainable tensors and fifteen full gradient tensors."""
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    sep, tr = _trainer(realtime)
    before = tr.state_dict()
    loss, mse, msk = tr.step(x, y_t, apply_update=False)
    worst = _check(g, tag, mse, msk, tr.gradients(), rtol=2e-3, kink_rtol=0.15)
    assert worst < 2e-3
    after = tr.state_dict()
    for k in before:      # gradients-only leaves parameters and running statistics alone
        assert torch.equal(before[k], after[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("tag,realtime", [("realtime", True), ("offline", False)])
def test_hip_training_gradients_bf16x6(tag, realtime):
    """The same check with the forward / data-gradient contractions on the bf16x6 matrix path (exact three-way
    operand cut, six MFMAs per product; the reference itself trains these layers under bf16 autocast)."""
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    sep, tr = _trainer(realtime, precision="bf16x6")
    loss, mse, msk = tr.step(x, y_t, apply_update=False)
    worst = _check(g, tag, mse, msk, tr.gradients(), rtol=2e-3, kink_rtol=0.15)
    assert worst < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("tag,realtime", [("realtime", True), ("offline", False)])
def test_hip_training_step_updates_like_adamw(tag, realtime):
    """One full step: running statistics follow nn.BatchNorm2d.train() (reference fixture), the parameter
    update equals torch.optim.AdamW(lr=1e-3, weight_decay=1e-5) applied to the same gradients, and a few
    steps on one batch bring the loss down."""
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    sep, tr = _trainer(realtime)
    before = tr.state_dict()
    loss0, _, _ = tr.step(x, y_t, apply_update=True)
    grads = tr.gradients()
    after = tr.state_dict()
    k = "sliced_umx.1.cdaes.0.4"
    assert np.allclose(after[k + ".running_mean"].numpy(), g[f"{tag}_running_mean::{k}"], rtol=1e-4, atol=1e-6)
    assert np.allclose(after[k + ".running_var"].numpy(), g[f"{tag}_running_var::{k}"], rtol=1e-4, atol=1e-7)
    keys = ["sliced_umx.0.input_mean", "sliced_umx.0.cdaes.1.0.weight", "sliced_umx.1.cdaes.0.3.weight",
            "sliced_umx.2.cdaes.1.6.weight", "sliced_umx.69.cdaes.3.9.weight", "sliced_umx.69.cdaes.0.7.bias",
            "sliced_umx.33.cdaes.2.9.bias"]
    ps = [torch.nn.Parameter(before[k_].clone()) for k_ in keys]
    opt = torch.optim.AdamW(ps, lr=1e-3, weight_decay=1e-5)
    for p, k_ in zip(ps, keys):
        p.grad = grads[k_].clone()
    opt.step()
    for p, k_ in zip(ps, keys):
        # (the first AdamW step moves every weight by ~lr * sign(g); tiny |g| next to eps = 1e-8 is where
        #  fp32 rounding of the gradient shows, hence the absolute term)
        assert torch.allclose(after[k_], p.detach(), rtol=1e-5, atol=2e-6), k_
    losses = [loss0] + [tr.step(x, y_t)[0] for _ in range(4)]
    assert losses[-1] < losses[0], losses
    # the trained tensors load back into the inference module
    tr.sync_to(sep.xumx_model)
    est = sep(x[:1].cuda())
    assert torch.isfinite(est).all()


# ---- fixture B: another seeded batch whose near-kink groups lie in other blocks than fixture A's -----------------
def test_fixture_b_covers_what_fixture_a_leaves_loose():
    g = load_golden("training_step_b.npz")
    for tag in ("realtime", "offline"):
        loose_b = _loose_b(g, tag)
        names = [str(k) for k in g["param_names"]]
        loose_a = [k for k in names if k.startswith(KINKS[tag][3])]
        assert loose_a and not any(k.startswith(loose_b) for k in loose_a), tag      # every tensor is strict in A or in B
        # the fixture carries the full gradients of A's loose tensors
        b = KINKS[tag][0]
        assert any(key.startswith(f"{tag}_grad::sliced_umx.{b}.") for key in g.files)
        mins = dict(zip([str(x) for x in g[f"{tag}_bn_names"]], g[f"{tag}_bn_min_abs"]))
        assert all(v >= float(g["risk"]) for k, v in mins.items() if k.startswith(f"sliced_umx.{b}."))


@pytest.mark.parametrize("tag,causal,wiener", [("realtime", True, False), ("offline", False, True)])
def test_oracle_training_gradients_match_reference_fixture_b(oracle_plan, seeded_sd, tag, causal, wiener):
    from oracle import loss as oloss
    g = load_golden("training_step_b.npz")
    x, y_t = _inputs_b(g)
    loss, mse, msk, grads = oloss.training_gradients(oracle_plan, seeded_sd, x, y_t, causal=causal, wiener=wiener)
    _check(g, tag, mse, msk, grads, rtol=2e-3, kink_rtol=0.15, loose=_loose_b(g, tag))


@pytest.mark.gpu
@pytest.mark.parametrize("tag,realtime", [("realtime", True), ("offline", False)])
def test_hip_training_gradients_match_reference_fixture_b(tag, realtime):
    """The tensors upstream of fixture A's kinks (block 32 / target 1, block 36 / target 0) held at 2e-3."""
    g = load_golden("training_step_b.npz")
    x, y_t = _inputs_b(g)
    sep, tr = _trainer(realtime)
    loss, mse, msk = tr.step(x, y_t, apply_update=False)
    grads = tr.gradients()
    worst = _check(g, tag, mse, msk, grads, rtol=2e-3, kink_rtol=0.15, loose=_loose_b(g, tag))
    assert worst < 2e-3
    strict = [k for k in (str(n) for n in g["param_names"]) if k.startswith(KINKS[tag][3])]
    norms = dict(zip([str(k) for k in g["param_names"]], g[f"{tag}_grad_norms"]))
    for k in strict:                             # explicit: A's loose set, strict here
        got = float(grads[k].double().norm())
        assert abs(got - norms[k]) <= 2e-3 * norms[k] + 2e-7, (k, got, norms[k])


@pytest.mark.gpu
def test_optimizer_state_round_trip_and_incremental_sync():
    """AdamW moments + step counter through the ABI (what optimizer.state_dict() checkpoints in the reference,
    training.py:419-430): a resumed trainer continues bit for bit; sync_to counts only the steps since the last sync."""
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    sep, tr = _trainer(False)
    for _ in range(2):
        tr.step(x, y_t)
    opt, params = tr.optimizer_state_dict(), tr.state_dict()
    # torch.optim.AdamW's own layout: the reference's optimizer loads it and hands it back unchanged
    keys = [k for k in params if not k.endswith(("running_mean", "running_var"))]
    assert sorted(opt) == ["param_groups", "state"] and opt["param_groups"][0]["params"] == list(range(len(keys)))
    i3 = keys.index("sliced_umx.1.cdaes.0.3.weight")
    assert float(opt["state"][i3]["step"]) == 2.0 and float(opt["state"][i3]["exp_avg_sq"].abs().sum()) > 0
    topt = torch.optim.AdamW([torch.nn.Parameter(params[k].clone()) for k in keys], lr=1e-3, weight_decay=1e-5)
    topt.load_state_dict(opt)
    back = topt.state_dict()
    assert all(torch.equal(back["state"][i]["exp_avg"], opt["state"][i]["exp_avg"]) for i in (0, i3, len(keys) - 1))
    opt = back                                    # resume from what torch re-emits
    tr.step(x, y_t)
    want = tr.state_dict()
    sep2, tr2 = _trainer(False)
    tr2.load_state_dict(params)
    tr2.load_optimizer_state_dict(opt)
    tr2.step(x, y_t)
    got = tr2.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want)
    # without the moments the third step differs (bias correction restarts): the restore is what made it equal
    sep3, tr3 = _trainer(False)
    tr3.load_state_dict(params)
    tr3.step(x, y_t)
    assert not torch.equal(tr3.state_dict()["sliced_umx.1.cdaes.0.3.weight"], want["sliced_umx.1.cdaes.0.3.weight"])
    k = "sliced_umx.1.cdaes.0.4.num_batches_tracked"
    base = int(sep.xumx_model.state_dict()[k])
    tr.sync_to(sep.xumx_model)
    tr.sync_to(sep.xumx_model)                   # a second sync without new steps adds nothing
    assert int(sep.xumx_model.state_dict()[k]) == base + 3
    tr.step(x, y_t)
    tr.sync_to(sep.xumx_model)
    assert int(sep.xumx_model.state_dict()[k]) == base + 4


# ---- BASELINE configs[4] at its stated size: batch of 16 two-second chunks (S = 11), offline model + Wiener-EM -------
RISK16 = 3e-6      # |BatchNorm output| below this may land on the other side of the ReLU kink in another summation order


def _inputs16(n=88200, B=16):
    from oracle import precompute
    return precompute.inputs16(n, B)


@pytest.fixture(scope="module")
def oracle_step16(oracle_plan, seeded_sd):
    """The oracle's autograd on the host for the bench's own B = 16 batch (~45 s on 8 cores), shared by the tests below;
    computed by the background job tests/conftest.py started at collection (oracle/precompute.py train16) or inline."""
    from conftest import precomputed
    from oracle import precompute
    assert precompute.RISK16 == RISK16
    r = precomputed("train16", lambda: precompute.train16(oracle_plan, seeded_sd))
    x, y_t = _inputs16()
    return x, y_t, r["mse"], r["msk"], r["grads"], r["risk"]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x6"])
def test_hip_training_step_at_config_size_matches_the_oracle(oracle_step16, precision):
    """One Trainer.step at B = 16 x 88,200 samples (BASELINE configs[4]; training.py:34-112) against the oracle's
    loss.backward(): both loss terms at 1e-4 relative and the gradient norm of EVERY trainable tensor at 2e-3.  Tensors
    upstream of a ReLU whose closest pre-activation lies within RISK16 of the kink (and the whitening of their block)
    carry a subgradient choice -- one row of the group switched on or off.  Every tensor is first held at 2e-3; the ones
    that exceed it must belong to such a group, stay within 15 % (fixtures A / B use the same bound) and number < 2 %."""
    x, y_t, mse_o, msk_o, grads_o, risk = oracle_step16
    sep, tr = _trainer(False, precision=precision)
    loss, mse, msk = tr.step(x, y_t, apply_update=False)
    assert abs(mse - mse_o) < 1e-4 * mse_o and abs(msk - msk_o) < 1e-4 * msk_o, (mse, mse_o, msk, msk_o)
    loose = tuple(p for g in risk for p in (g + ".", g.split(".cdaes.")[0] + ".input_"))
    grads = tr.gradients()
    assert sorted(grads) == sorted(grads_o)
    # strict bound first; what exceeds it must sit upstream of a near-kink ReLU, stay inside 15 %, and be rare
    worst, worst_key, over = 0.0, None, []
    for k, ref in grads_o.items():
        want, got = float(ref.double().norm()), float(grads[k].double().norm())
        err = abs(got - want)
        if err <= 2e-3 * want + 2e-7:
            if want > 1e-6 and err / want > worst:
                worst, worst_key = err / want, k
            continue
        assert k.startswith(loose) and err <= 0.15 * want + 2e-7, (precision, k, got, want, k.startswith(loose))
        over.append(k)
    assert len(over) <= 0.02 * len(grads_o), (len(over), over[:8])
    # a few full tensors, elementwise
    for k in ("sliced_umx.0.cdaes.1.0.weight", "sliced_umx.40.cdaes.2.3.weight", "sliced_umx.69.cdaes.3.9.weight",
              "sliced_umx.50.cdaes.0.6.weight", "sliced_umx.69.input_mean"):
        if k in over:
            continue
        ref = grads_o[k]
        err, scale = float((grads[k].cpu() - ref).abs().max()), float(ref.abs().max()) + 1e-12
        assert err <= 2e-3 * scale + 2e-7, (precision, k, err, scale)
    print(f"\n[B=16 step, {precision}] worst strict gradient-norm error {worst:.2e} ({worst_key}); {len(over)} of "
          f"{len(grads_o)} tensors past 2e-3 (all upstream of a near-kink ReLU, all within 15 %): {over[:6]}")


@pytest.mark.gpu
def test_pending_loss_equals_the_waited_step_and_expires_after_four_steps():
    """Trainer.step(wait=False): same step, same numbers, the loss just read later (by ticket, xsq_train_loss)."""
    from xumx_slicq_amd import _lib
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    _, tr_a = _trainer(False)
    _, tr_b = _trainer(False)
    waited = [tr_a.step(x, y_t) for _ in range(3)]
    pend = [tr_b.step(x, y_t, wait=False) for _ in range(3)]          # three steps in flight before any loss is read
    assert [tuple(p) for p in pend] == waited
    assert pend[1][0] == waited[1][0] and pend[2].result() == waited[2]
    sa, sb = tr_a.state_dict(), tr_b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    old = tr_b.step(x, y_t, wait=False)
    for _ in range(4):
        tr_b.step(x, y_t, wait=False)
    with pytest.raises(_lib.XsqError):
        old.result()                                                  # five steps later its ring slot has been reused


@pytest.mark.gpu
def test_hip_training_step_bf16_arm_sits_inside_the_reference_autocast_spread(oracle_plan, seeded_sd):
    """BASELINE configs[4] as written: "training.py step ... bf16".  The reference runs forward + loss under
    torch.autocast("cpu", dtype=torch.bfloat16) (training.py:66-108,473-476); tests/golden/training_step_bf16.npz
    (oracle/make_golden_train_bf16.py, the reference's own autograd) holds that step on fixture A's batch next to the same
    step in fp32.  The reference's bf16 step is NOT close to its fp32 step -- loss terms 2.4e-4 apart, per-tensor gradients
    7.6 % apart in the median, ~100 % at the 90th percentile -- so the tolerances of the bf16 arm (Trainer(precision=
    "bf16"): forward / data-gradient / weight-gradient operands rounded to bf16, one MFMA per product, fp32 everything else) come from that
    spread, not from a guess:
      * both loss terms within 1e-3 of the autocast reference's and of the fp32 reference's (4x the reference's own gap);
      * over all trainable tensors, the arm's relative distance to the fp32 gradients (the oracle's autograd, pinned to the
        reference at 2e-3 by fixture A) has a median and a 90th percentile no larger than 1.25x the autocast reference's own;
      * twelve full gradient tensors lie within 2.5x their own fp32-vs-autocast distance (+ 2 % of the tensor's norm) of the
        autocast reference's."""
    import numpy as np
    from oracle import loss as oloss
    g = load_golden("training_step_bf16.npz")
    x, y_t = _inputs(int(g["n"]))
    sep, tr = _trainer(False, precision="bf16")
    loss, mse, msk = tr.step(x, y_t, apply_update=False)
    grads = tr.gradients()
    for ref in ("bf16", "fp32"):
        assert abs(mse - float(g[f"{ref}_mse"])) < 1e-3 * float(g["fp32_mse"]), (ref, mse, float(g[f"{ref}_mse"]))
        assert abs(msk - float(g[f"{ref}_mask"])) < 1e-3 * float(g["fp32_mask"]), (ref, msk, float(g[f"{ref}_mask"]))
    # the fp32 gradients of this batch: the HIP fp32 arm's, held to the reference's own autograd on this very batch at 2e-3
    # per tensor by test_hip_training_gradients_match_reference[offline] (fixture A) -- 30x below the spread measured here.
    # (Until round 6 the oracle's autograd was recomputed for this: 55 s of host CPU; the B = 16 test below keeps the
    # oracle as its fp32 side.)
    _sep32, tr32 = _trainer(False, precision="fp32")
    tr32.step(x, y_t, apply_update=False)
    grads_o = {k: v.cpu() for k, v in tr32.gradients().items()}
    names = [str(k) for k in g["param_names"]]
    rel_ref = dict(zip(names, g["rel_diff_bf16_vs_fp32"].tolist()))
    norm32 = dict(zip(names, g["fp32_grad_norms"].tolist()))
    mine, theirs = [], []
    for k in names:
        if norm32[k] <= 1e-6:
            continue
        want = grads_o[k].double()
        mine.append(float((grads[k].cpu().double() - want).norm() / want.norm()))
        theirs.append(rel_ref[k])
    mine, theirs = np.asarray(mine), np.asarray(theirs)
    print(f"\n[bf16 arm] relative distance to the fp32 gradients over {len(mine)} tensors: median {np.median(mine):.3e} "
          f"(reference autocast {np.median(theirs):.3e}), 90 % {np.quantile(mine, 0.9):.3e} ({np.quantile(theirs, 0.9):.3e}); "
          f"loss {mse:.6f} / {msk:.6f} against autocast {float(g['bf16_mse']):.6f} / {float(g['bf16_mask']):.6f}")
    assert np.median(mine) <= 1.25 * np.median(theirs) and np.quantile(mine, 0.9) <= 1.25 * np.quantile(theirs, 0.9)
    assert np.median(mine) > 1e-4                    # ... and it IS the bf16 arithmetic (the fp32 arm sits at ~1e-6 here)
    for key in g.files:
        if not key.startswith("bf16_grad::"):
            continue
        k = key.split("::", 1)[1]
        if norm32[k] <= 1e-6:                     # (a tensor whose gradient is rounding noise in fp32 already: input_mean of block 0)
            continue
        ref16 = torch.from_numpy(g[key]).double()
        d = float((grads[k].cpu().double() - ref16).norm())
        assert d <= (2.5 * rel_ref[k] + 0.02) * norm32[k] + 1e-7, (k, d, rel_ref[k], norm32[k])


def test_oracle_step_at_config_size_matches_the_reference_fixture(oracle_step16):
    """The B = 16 fixture (oracle/make_golden_train_bf16.py --b16: the REFERENCE's own step on bench.py's configs[4] batch,
    in fp32 and under bf16 autocast) pins the oracle at the config's stated size: both loss terms at 1e-4, every tensor's
    gradient norm at 2e-3 (near-kink groups at 15 %, < 2 % of the tensors) -- the bounds the HIP step is held to."""
    g = load_golden("training_step_bf16_b16.npz")
    _x, _y, mse_o, msk_o, grads_o, risk = oracle_step16
    assert int(g["n"]) == 88200
    assert abs(mse_o - float(g["fp32_mse"])) < 1e-4 * float(g["fp32_mse"]) and abs(msk_o - float(g["fp32_mask"])) < 1e-4 * float(g["fp32_mask"])
    loose = tuple(p for grp in risk for p in (grp + ".", grp.split(".cdaes.")[0] + ".input_"))
    over = []
    for k, want in zip((str(k) for k in g["param_names"]), g["fp32_grad_norms"].tolist()):
        got = float(grads_o[k].double().norm())
        if abs(got - want) <= 2e-3 * want + 2e-7:
            continue
        assert k.startswith(loose) and abs(got - want) <= 0.15 * want + 2e-7, (k, got, want)
        over.append(k)
    assert len(over) <= 0.02 * len(grads_o), over[:8]


@pytest.mark.gpu
def test_hip_training_step_bf16_arm_at_config_size(oracle_step16):
    """BASELINE configs[4] AS WRITTEN AND AT ITS STATED SIZE: "training.py step ... bf16, batch=16 chunks" -- the shape and
    arithmetic of bench.py's `train_step_bf16` number.  tests/golden/training_step_bf16_b16.npz holds the reference's own
    step on this batch (B = 16 x 88,200, seeds 700..703) under torch.autocast("cpu", dtype=torch.bfloat16)
    (training.py:66-108,473-476) next to the same step in fp32; at this size the reference's bf16 step is 1.4e-4 / 7.9e-5
    from its fp32 loss terms and 5.8 % (median; 98 % at the 90th percentile) from its fp32 gradients.  Same spread-derived
    bounds as the B = 2 test: loss terms within 1e-3 of both references; the arm's per-tensor distance to the fp32 gradients
    (the oracle's, pinned to the reference by the test above) no larger than 1.25x the autocast reference's own in the median
    and at the 90th percentile; the stored full tensors within 2.5x their own fp32-vs-autocast distance (+ 2 %) of the
    autocast reference's."""
    import numpy as np
    g = load_golden("training_step_bf16_b16.npz")
    x, y_t, _mse_o, _msk_o, grads_o, _risk = oracle_step16
    sep, tr = _trainer(False, precision="bf16")
    loss, mse, msk = tr.step(x, y_t, apply_update=False)
    grads = tr.gradients()
    for ref in ("bf16", "fp32"):
        assert abs(mse - float(g[f"{ref}_mse"])) < 1e-3 * float(g["fp32_mse"]), (ref, mse, float(g[f"{ref}_mse"]))
        assert abs(msk - float(g[f"{ref}_mask"])) < 1e-3 * float(g["fp32_mask"]), (ref, msk, float(g[f"{ref}_mask"]))
    names = [str(k) for k in g["param_names"]]
    rel_ref = dict(zip(names, g["rel_diff_bf16_vs_fp32"].tolist()))
    norm32 = dict(zip(names, g["fp32_grad_norms"].tolist()))
    mine, theirs = [], []
    for k in names:
        if norm32[k] <= 1e-6:
            continue
        want = grads_o[k].double()
        mine.append(float((grads[k].cpu().double() - want).norm() / want.norm()))
        theirs.append(rel_ref[k])
    mine, theirs = np.asarray(mine), np.asarray(theirs)
    print(f"\n[bf16 arm, B = 16] relative distance to the fp32 gradients over {len(mine)} tensors: median {np.median(mine):.3e} "
          f"(reference autocast {np.median(theirs):.3e}), 90 % {np.quantile(mine, 0.9):.3e} ({np.quantile(theirs, 0.9):.3e}); "
          f"loss {mse:.6f} / {msk:.6f} against autocast {float(g['bf16_mse']):.6f} / {float(g['bf16_mask']):.6f}")
    assert np.median(mine) <= 1.25 * np.median(theirs) and np.quantile(mine, 0.9) <= 1.25 * np.quantile(theirs, 0.9)
    assert np.median(mine) > 1e-4                    # it IS the bf16 arithmetic
    for key in g.files:
        if not key.startswith("bf16_grad::"):
            continue
        k = key.split("::", 1)[1]
        if norm32[k] <= 1e-6:
            continue
        ref16 = torch.from_numpy(g[key]).double()
        d = float((grads[k].cpu().double() - ref16).norm())
        assert d <= (2.5 * rel_ref[k] + 0.02) * norm32[k] + 1e-7, (k, d, rel_ref[k], norm32[k])


@pytest.mark.gpu
def test_packed_bf16_weight_gradient_kernel_is_bitwise_the_lane_converting_one():
    """Round 6: the bf16 arm's weight gradients on `wgrad_bf16p_kernel` (csrc/wgrad.h: operands rounded to bf16 ONCE at staging
    time and held k-pair-packed in LDS, one 16-byte fragment read per MFMA operand; XSQ_TRAIN_WGRAD_PACKED=1, an A/B arm that
    measured slower and stays off by default) against round 5's `wgrad_kernel<Op, true>` (fp32 K-steps in LDS, eight 4-byte reads
    + four conversions per fragment).  The same values are
    rounded by the same instruction and meet in the same MFMA in the same order: every gradient tensor must come out BITWISE
    equal -- on fixture A's batch (B = 2) and on the config's B = 16 batch (ragged last chunks, all four weight-gradient launches)."""
    import os
    g = load_golden("training_step_bf16.npz")
    for (x, y_t) in (_inputs(int(g["n"])), _inputs16()):
        res = {}
        for packed in ("1", "0"):
            os.environ["XSQ_TRAIN_WGRAD_PACKED"] = packed
            try:
                _sep, tr = _trainer(False, precision="bf16")
                res[packed] = (tr.step(x, y_t, apply_update=False), {k: v.clone() for k, v in tr.gradients().items()})
            finally:
                os.environ.pop("XSQ_TRAIN_WGRAD_PACKED", None)
        assert res["1"][0] == res["0"][0]
        bad = [k for k in res["1"][1] if not torch.equal(res["1"][1][k], res["0"][1][k])]
        assert not bad, (len(bad), bad[:5])
