"""Pins the CPU oracle (oracle/*.py) against golden vectors produced by the real reference
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import torch

from _golden import load, sub, t, rel_err
from oracle import critic_oracle as CO
from oracle import ge2e_oracle as GO
from oracle import tts_oracle as TO

TOL = 2e-6


def test_highway_fwd_bwd():
    g = load("highway.npz")
    for i, (k, d, causal) in enumerate(g["configs"]):
        pre = "c%d/" % i
        sd = {("hc." + n): v.clone().requires_grad_(True) for n, v in sub(g, pre + "sd/").items()}
        x = t(g[pre + "x"]).requires_grad_(True)
        y = TO.highway_conv(x, sd, "hc", int(k), int(d), bool(causal))
        assert rel_err(y, t(g[pre + "y"])) < TOL
        y.backward(t(g[pre + "dy"]))
        assert rel_err(x.grad, t(g[pre + "dx"])) < 1e-5
        for n, gr in sub(g, pre + "grad/").items():
            assert rel_err(sd["hc." + n].grad, gr) < 1e-5, n


def test_melsyn_train_and_losses():
    g = load("melsyn_train.npz")
    sd = {n: v.clone().requires_grad_(True) for n, v in sub(g, "sd/").items()}
    Y, A = TO.melsyn_train(t(g["mel_in"]), t(g["text"]), t(g["spk"]), sd)
    assert rel_err(Y, t(g["Y"])) < TOL
    assert rel_err(A, t(g["A"])) < TOL
    gaw = TO.guided_attention_mat(24, 32)
    assert torch.equal(gaw, t(g["gaw"]))
    l1, bd, att = TO.text2mel_losses(Y, A, t(g["mel_gt"]), gaw)
    for mine, ref in ((l1, "l1"), (bd, "bd"), (att, "att")):
        assert abs(float(mine) - float(g[ref])) < 1e-6 * max(1.0, abs(float(g[ref])))
    (l1 + bd + att).backward()
    worst = max(rel_err(sd[n].grad, gr) for n, gr in sub(g, "grad/").items())
    assert worst < 2e-4, worst


def test_melsyn_eval_loop_indices_exact():
    g = load("melsyn_eval.npz")
    sd = sub(g, "sd/")
    with torch.no_grad():
        Y, A, pma = TO.synthesize_loop(t(g["text"]), t(g["spk"]), sd, int(g["steps"]))
    assert torch.equal(pma, t(g["pma"]))          # attention indices: bit-exact
    assert rel_err(Y, t(g["Y"])) < 1e-5
    assert rel_err(A, t(g["A"])) < 1e-5
    assert float(g["margins"].min()) > 1e-2        # recorded top-2 margin >> rounding


def test_ssrn_small_and_full():
    g = load("ssrn_small.npz")
    sd = {n: v.clone().requires_grad_(True) for n, v in sub(g, "sd/").items()}
    mel = t(g["mel"]).requires_grad_(True)
    P = TO.ssrn(mel, sd)
    assert rel_err(P, t(g["P"])) < TOL
    l1, bd = TO.ssrn_losses(P, t(g["lin"]))
    assert abs(float(l1) - float(g["l1"])) < 1e-6 and abs(float(bd) - float(g["bd"])) < 1e-6
    (l1 + bd).backward()
    assert rel_err(mel.grad, t(g["dmel"])) < 2e-4
    worst = max(rel_err(sd[n].grad, gr) for n, gr in sub(g, "grad/").items())
    assert worst < 2e-4, worst


def test_ssrn_full_size_config1_oracle_vs_reference():
    """BASELINE config 1 ("SSRN forward on 1 synthetic mel (80 x 200), CPU reference path only"): the oracle's full-size SSRN
    (models/TTSModel.py:342-362) on the seeded input and the seeded initialisation of G4 against the reference's stored output
    (strided 1/64 slice, sum).  The weights come from the host-side module's construction order + init_weights, which G9 pins
    (tests/test_host_cpu.py); no HIP code runs."""
    from spoofsv_amd.train import init_weights
    from spoofsv_amd.tts import SSRN
    g = load("ssrn_full.npz")
    torch.manual_seed(int(g["w_seed"]))
    m = SSRN(80, 513, 256)
    m.apply(init_weights)
    torch.manual_seed(int(g["x_seed"]))
    x = torch.rand(1, 80, 200)
    with torch.no_grad():
        y = TO.ssrn(x, {k: v.detach() for k, v in m.state_dict().items()})
    assert tuple(y.shape) == (1, 513, 800)
    assert rel_err(y[0, ::8, ::8], t(g["y_slice"])) < TOL, rel_err(y[0, ::8, ::8], t(g["y_slice"]))
    assert abs(float(y.double().sum()) - float(g["y_sum"])) < 1e-6 * float(g["y_abs"])


def test_gaw_probes():
    g = load("gaw.npz")
    W = TO.guided_attention_mat(186, 325)
    assert np.array_equal(W.numpy()[g["n"], g["t"]], g["w"])
    assert np.array_equal(W.numpy()[93], g["row93"])
    assert abs(float(W.double().sum()) - float(g["total"])) < 1e-6


def test_adam_matches_torch():
    g = load("adam.npz")
    p = t(g["p0"]).clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for s in (1, 2, 3):
        TO.adam_step(p, t(g["g%d" % s]), m, v, s)
        assert rel_err(p, t(g["p%d" % s])) < 1e-6


def test_ge2e_embedder():
    g = load("ge2e_embedder.npz")
    sd = sub(g, "sd/")
    x = t(g["x"])
    hs = GO.lstm_stack(x, sd, 3)
    assert rel_err(hs, t(g["h_last_layer"])) < 1e-5
    e = GO.speech_embedder(x, sd)
    assert rel_err(e, t(g["e"])) < 1e-5


def test_ge2e_loss_and_known_answer():
    g = load("ge2e_loss.npz")
    emb = t(g["emb"]).requires_grad_(True)
    w = torch.tensor(10.0, requires_grad=True)
    b = torch.tensor(-5.0, requires_grad=True)
    assert rel_err(GO.ge2e_cossim(emb), t(g["cossim"])) < 1e-5
    loss, _ = GO.ge2e_loss(emb, w, b)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    assert rel_err(emb.grad, t(g["demb"])) < 1e-4
    assert abs(float(w.grad) - float(g["dw"])) < 1e-4 * max(1, abs(float(g["dw"])))
    assert abs(float(b.grad) - float(g["db"])) < 1e-4 * max(1, abs(float(g["db"])))
    # the reference's own known-answer case (GE2E/utils.py:89-96): loss = 5.2501
    kl, kper = GO.ge2e_loss(t(g["kat_emb"]), 1.0, 0.0)
    assert rel_err(GO.ge2e_cossim(t(g["kat_emb"])), t(g["kat_cossim"])) < 1e-6
    assert abs(float(kl) - 5.2501) < 1e-4 and abs(float(kl) - float(g["kat_loss"])) < 1e-5
    assert rel_err(kper, t(g["kat_per"])) < 1e-5

# Biases that feed a LayerNorm have an exactly-zero true gradient; what autograd returns for them is rounding noise.
# They are excluded; every other critic gradient must agree in the relative L2 norm.  (Post-step WEIGHTS are not
# compared for the critic: Adam's first step maps every gradient to +-lr, so near-zero entries flip on rounding noise.)
_ZERO_GRAD = ("conv1.bias", "hc.conv.bias", "conv2.bias", "conv3.bias", "conv4.bias", "conv5.bias")


def _critic_grads_agree(sd, want, tol):
    for k, p in sd.items():
        if k in _ZERO_GRAD:
            continue
        a, b = p.grad.detach().cpu().double(), want[k].double()
        if float(b.norm()) == 0.0:         # exactly cancelling terms (e.g. the last LayerNorm's bias when no leaky-ReLU changes sign)
            assert float(a.norm()) < 1e-6, k
            continue
        assert float((a - b).norm() / b.norm()) < tol, (k, float((a - b).norm() / b.norm()))


def test_adversarial_iteration_golden_cpu():
    """G8: the oracle generator + the oracle critic reproduce the reference's G and D iterations (critic in eval mode)."""
    g = load("adversarial_iter.npz")
    sd = {n: v.clone().requires_grad_(True) for n, v in sub(g, "m0/").items()}
    dsd = {n: v.clone().requires_grad_(True) for n, v in sub(g, "d0/").items()}
    d = lambda x: CO.critic(x, dsd, "mel", masks=False)
    mel, text, spk, gaw = t(g["mel_gt"]), t(g["text"]), t(g["spk"]), t(g["gaw"])
    mel_in = torch.cat((torch.zeros_like(mel[:, :, :1]), mel[:, :, :-1]), dim=-1)
    opt = torch.optim.Adam(list(sd.values()), 2e-4, (0.5, 0.9), 1e-6)
    pred, att = TO.melsyn_train(mel_in, text, spk, sd)
    l1, bd, la = TO.text2mel_losses(pred, att, mel, gaw)
    ld = torch.mean(-d(pred))
    loss = l1 + bd + la + (l1.item() + bd.item() + la.item()) / abs(ld.item()) * ld
    for mine, ref in ((l1, "g_l1"), (bd, "g_bd"), (la, "g_att"), (ld, "g_disc"), (loss, "g_all")):
        assert abs(float(mine) - float(g[ref])) < 2e-6 * max(1.0, abs(float(g[ref]))), ref
    loss.backward()
    opt.step()
    m1 = sub(g, "m1/")
    assert max(float((sd[k].detach() - m1[k]).abs().max()) for k in m1) < 1e-5      # 5 % of one Adam step (lr 2e-4)
    # D iteration with the stored interpolation coefficients
    sd1 = {k: v.clone() for k, v in m1.items()}
    with torch.no_grad():
        pred, _ = TO.melsyn_train(mel_in, text, spk, sd1)
    for v in dsd.values():
        v.grad = None      # the G iteration left gradients on the critic; the reference zeroes both optimizers (:264-265)
    gp, loss_d = CO.critic_losses(pred, mel, t(g["coeff"]), dsd, "mel", 10.0, masks=False)
    gp.backward()
    loss_d.backward()
    assert abs(float(gp) - float(g["d_gp"])) < 1e-5 * max(1.0, abs(float(g["d_gp"])))
    assert abs(float(loss_d) - float(g["d_loss"])) < 1e-6 * max(1.0, abs(float(g["d_loss"])))
    _critic_grads_agree(dsd, sub(g, "dgrad/"), 1e-4)


def test_critic_iteration_with_dropout_active():
    """G11: the reference's melDisc and linDisc in TRAINING mode (the only mode the reference runs them in): with the
    stored seed the oracle draws the same dropout masks at the same sites, so the first call's output, its input gradient,
    both losses and every critic gradient (penalty double-backward included) must reproduce."""
    g = load("critic_dropout.npz")
    for kind in ("mel", "lin"):
        dsd = {n: v.clone().requires_grad_(True) for n, v in sub(g, kind + "/sd/").items()}
        gt, pred, coeff = t(g[kind + "/gt"]), t(g[kind + "/pred"]), t(g[kind + "/coeff"])
        torch.manual_seed(int(g[kind + "/seed"]))
        drawn = []
        gp, loss_d = CO.critic_losses(pred, gt, coeff, dsd, kind, 10.0, masks=None, drawn=drawn)
        assert len(drawn) == 9 and all(0.0 < float((m == 0).float().mean()) < 0.15 for m in drawn[:2])     # p = 0.05 zeros
        gp.backward()
        loss_d.backward()
        assert abs(float(gp) - float(g[kind + "/gp"])) < 1e-5 * max(1.0, abs(float(g[kind + "/gp"])))
        assert abs(float(loss_d) - float(g[kind + "/loss_d"])) < 1e-6 * max(1.0, abs(float(g[kind + "/loss_d"])))
        _critic_grads_agree(dsd, sub(g, kind + "/grad/"), 1e-4)
        # the same masks injected reproduce the first call and its input gradient exactly
        c = coeff.view(-1, 1, 1)
        mid = (c * gt + (1 - c) * pred).requires_grad_(True)
        out = CO.critic(mid, dsd, kind, masks=drawn[:3])
        assert rel_err(out, t(g[kind + "/out_mid"])) < 1e-6
        dmid, = torch.autograd.grad(out, mid, torch.ones_like(out))
        assert rel_err(dmid, t(g[kind + "/dmid"])) < 1e-5


def test_ge2e_training_iteration():
    """G10: loss, every parameter gradient and the clipped SGD step of one GE2E training iteration (reference modules)."""
    g = load("ge2e_train.npz")
    N, M, T, H, P = [int(v) for v in g["dims"]]
    sd = {k[3:]: t(v) for k, v in g.items() if k.startswith("p0/")}
    loss, grads, (dw, db), new_sd, (w1, b1) = GO.ge2e_train_step(t(g["x"]), sd, float(g["w0"]), float(g["b0"]), N, M)
    assert rel_err(loss, t(g["loss"])) < 1e-5
    for k, v in grads.items():
        assert rel_err(v, t(g["g/" + k])) < 2e-4, (k, rel_err(v, t(g["g/" + k])))
    # dw, db are sums with heavy cancellation (|dw| ~ 5e-5 against per-embedding terms of order 1): absolute tolerance
    assert abs(float(dw) - float(g["dw"])) < 1e-5 and abs(float(db) - float(g["db"])) < 1e-5
    for k, v in new_sd.items():
        assert rel_err(v, t(g["p1/" + k])) < 1e-5, k
    assert abs(float(w1) - float(g["w1"])) < 1e-5 and abs(float(b1) - float(g["b1"])) < 1e-5
    # verification-style similarity (enrollment centroids of another set), train_speech_embedder.py:156-159
    assert rel_err(GO.ge2e_cossim(t(g["ev_ver"]), t(g["ev_cent"])), t(g["ev_sim"])) < 1e-5
    assert rel_err(t(g["ev_enr"]).mean(dim=1), t(g["ev_cent"])) < 1e-6
