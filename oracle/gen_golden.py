#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on CPU.  TEST INFRASTRUCTURE.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's modules unmodified, feeds them seeded inputs and stores
inputs + parameters + outputs (+ gradients) as small fixtures.  Nothing from the reference
is copied: fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference]

Fixtures (names follow SURVEY.md section 8c):
  highway.npz        G1  highwayConv fwd+bwd, 5 (k, dilation, causal) configs
  melsyn_train.npz   G2  melSyn train fwd+bwd at reduced dims, with the a12 losses (G5)
  melsyn_eval.npz    G3  30-step free-running loop, int64 pma sequence, top-2 margins
  ssrn_small.npz     G4  SSRN reduced dims fwd+bwd
  ssrn_full.npz      G4  SSRN full size (1,80,200)->(1,513,800): checksums + strided slice
  gaw.npz            G5  guided_attention_mat(186,325) probes
  ge2e_embedder.npz  G6  SpeechEmbedder reduced dims
  ge2e_loss.npz      G7  GE2ELoss random case + the utils.py:89-96 known-answer case
  ge2e_train.npz     G10 one GE2E training iteration (loss, gradients, clipped SGD step)
  init_pin.npz       G9  seed -> construct -> apply(init_weights): per-parameter checksums
  adam.npz           a13 three torch.optim.Adam steps with the config.json hyper-parameters
  adversarial_iter.npz G8 one G and one D iteration with the reference's melSyn + melDisc (critic dropout off)
  critic_dropout.npz G11 melDisc / linDisc critic iteration in TRAINING mode (dropout active, as the reference runs them)
"""
import argparse
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _sd_np(sd, pre="sd/"):
    return {pre + k: _np(v) for k, v in sd.items()}


def _grads_np(mod, pre="grad/"):
    return {pre + k: _np(p.grad) for k, p in mod.named_parameters()}


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def init_weights(layer):
    # the reference's initialiser (train/ordinary.py:16-19), re-stated: He-normal on every
    # weight with more than one dimension.
    if hasattr(layer, "weight") and len(layer.weight.shape) > 1:
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")


def randomize_ln(mod, gen):
    """LayerNorm affine params default to 1/0; perturb them so that gamma/beta gradients
    and their use are actually exercised by the fixtures."""
    for m in mod.modules():
        if isinstance(m, torch.nn.LayerNorm):
            with torch.no_grad():
                m.weight.add_(0.3 * torch.randn(m.weight.shape, generator=gen))
                m.bias.add_(0.3 * torch.randn(m.bias.shape, generator=gen))


def gen_highway(TTS):
    out = {}
    cfgs = [(3, 1, 0), (3, 3, 0), (3, 9, 1), (3, 27, 1), (1, 1, 0)]
    out["configs"] = np.array(cfgs, dtype=np.int64)
    B, C, L = 2, 16, 40
    for i, (k, d, causal) in enumerate(cfgs):
        torch.manual_seed(100 + i)
        m = TTS.highwayConv(C, k, d, causal=bool(causal))
        m.apply(init_weights)
        randomize_ln(m, torch.Generator().manual_seed(7 + i))
        with torch.no_grad():
            m.conv.bias.uniform_(-0.5, 0.5)
        x = torch.randn(B, C, L, requires_grad=True)
        dy = torch.randn(B, C, L)
        y = m(x)
        y.backward(dy)
        pre = "c%d/" % i
        out.update(_sd_np(m.state_dict(), pre + "sd/"))
        out.update(_grads_np(m, pre + "grad/"))
        out[pre + "x"] = _np(x)
        out[pre + "y"] = _np(y)
        out[pre + "dy"] = _np(dy)
        out[pre + "dx"] = _np(x.grad)
    np.savez_compressed(os.path.join(OUT, "highway.npz"), **out)


def _t2m_inputs(B, N, T, F=80, spk_dim=200, vocab=34, seed=0):
    g = torch.Generator().manual_seed(seed)
    mel = torch.rand(B, F, T, generator=g)
    text = torch.randint(2, vocab, (B, 1, N), generator=g)
    text[:, :, -2] = 1   # 'E'
    text[:, :, -1] = 0   # 'P'
    spk = 0.04 + 0.05 * torch.rand(B, spk_dim, 1, generator=g)
    return mel, text, spk


def _guided(n, t, g=0.2):
    import math
    W = torch.zeros((n, t))
    for k1 in range(n):
        for k2 in range(t):
            W[k1, k2] = 1 - math.exp(-(k2 / t - k1 / n) ** 2 / (2 * g * g))
    return W


def gen_melsyn_train(TTS):
    import torch.nn.functional as F
    torch.manual_seed(1234)
    hidden, temb = 16, 8
    m = TTS.melSyn(vocab_len=34, condition=True, spkemb_dim=200, textemb_dim=temb, freq_bins=80, hidden_dim=hidden)
    m.apply(init_weights)
    randomize_ln(m, torch.Generator().manual_seed(5))
    m.train()
    B, N, T = 2, 12, 20
    mel_gt, text, spk = _t2m_inputs(B, N, T, seed=11)
    mel_in = torch.cat((torch.zeros_like(mel_gt[:, :, :1]), mel_gt[:, :, :-1]), dim=-1)
    Y, A = m(mel_in, text, spk)
    # the a12 losses exactly as train/ordinary.py:230-236 spells them, MAX dims (24, 32)
    gaw = _guided(24, 32)
    l1 = torch.mean(torch.abs(mel_gt - Y))
    bd = torch.mean(-mel_gt * torch.log(Y + 1e-8) - (1 - mel_gt) * torch.log(1 - Y + 1e-8))
    aug = F.pad(A, (0, 32 - A.size()[-1], 0, 24 - A.size()[-2]), value=-1)
    att = torch.sum(torch.ne(aug, -1).float() * aug * gaw) / torch.sum(torch.ne(aug, -1).float())
    Y.retain_grad()
    A.retain_grad()
    (l1 + bd + att).backward()
    out = dict(mel_gt=_np(mel_gt), mel_in=_np(mel_in), text=_np(text), spk=_np(spk), Y=_np(Y), A=_np(A),
               gaw=_np(gaw), l1=_np(l1), bd=_np(bd), att=_np(att), dY=_np(Y.grad), dA=_np(A.grad),
               dims=np.array([hidden, temb, B, N, T], dtype=np.int64))
    out.update(_sd_np(m.state_dict()))
    out.update(_grads_np(m))
    np.savez_compressed(os.path.join(OUT, "melsyn_train.npz"), **out)


def gen_melsyn_eval(TTS):
    hidden, temb = 16, 8
    B, N, steps = 2, 12, 30
    best = None
    for seed in range(2000, 2040):
        torch.manual_seed(seed)
        m = TTS.melSyn(vocab_len=34, condition=True, spkemb_dim=200, textemb_dim=temb, freq_bins=80, hidden_dim=hidden)
        m.apply(init_weights)
        randomize_ln(m, torch.Generator().manual_seed(seed))
        m.eval()
        _, text, spk = _t2m_inputs(B, N, 4, seed=seed)
        margins, pmas = [], []
        with torch.no_grad():
            init = torch.zeros(B, 80, 1)
            Y, A, pma, K, V = m(melspec=init, textid=text, spkemb=spk, pma=torch.zeros(B).long())
            inputs = torch.cat((init, Y), dim=-1)
            top = torch.topk(A[:, :, -1], 2, dim=1).values
            margins.append(top[:, 0] - top[:, 1]); pmas.append(pma.clone())
            for _ in range(steps):
                Y, A, pma = m(melspec=inputs, textid=None, spkemb=spk, K=K, V=V, A_last=A, pma=pma)
                inputs = torch.cat((inputs, Y[:, :, -1:]), dim=-1)
                top = torch.topk(A[:, :, -1], 2, dim=1).values
                margins.append(top[:, 0] - top[:, 1]); pmas.append(pma.clone())
        mm = float(torch.stack(margins).min())
        moved = int(torch.stack(pmas).max())
        score = mm if moved >= 3 else -1.0   # prefer runs where the attention actually advances
        if best is None or score > best[0]:
            best = (score, seed, m, text, spk, Y, A, torch.stack(pmas), torch.stack(margins), K, V)
    score, seed, m, text, spk, Y, A, pmas, margins, K, V = best
    print("melsyn_eval: seed %d, min top-2 margin %.4g, max pma %d" % (seed, score, int(pmas.max())))
    out = dict(text=_np(text), spk=_np(spk), Y=_np(Y), A=_np(A), pma=_np(pmas), margins=_np(margins),
               K=_np(K), V=_np(V), seed=np.array(seed), steps=np.array(steps),
               dims=np.array([hidden, temb, B, N], dtype=np.int64))
    out.update(_sd_np(m.state_dict()))
    np.savez_compressed(os.path.join(OUT, "melsyn_eval.npz"), **out)


def gen_ssrn(TTS):
    torch.manual_seed(4321)
    m = TTS.SSRN(freq_bins=80, output_bins=65, ssrn_dim=16)
    m.apply(init_weights)
    randomize_ln(m, torch.Generator().manual_seed(9))
    m.train()
    g = torch.Generator().manual_seed(3)
    mel = torch.rand(2, 80, 12, generator=g, requires_grad=True)
    lin = torch.rand(2, 65, 48, generator=g)
    P = m(mel)
    l1 = torch.mean(torch.abs(lin - P))
    bd = torch.mean(-lin * torch.log(P + 1e-8) - (1 - lin) * torch.log(1 - P + 1e-8))
    P.retain_grad()
    (l1 + bd).backward()
    out = dict(mel=_np(mel), lin=_np(lin), P=_np(P), l1=_np(l1), bd=_np(bd), dP=_np(P.grad), dmel=_np(mel.grad))
    out.update(_sd_np(m.state_dict()))
    out.update(_grads_np(m))
    np.savez_compressed(os.path.join(OUT, "ssrn_small.npz"), **out)

    # config 1: full-size SSRN forward on one synthetic mel (80 x 200)
    torch.manual_seed(1234)
    m = TTS.SSRN(freq_bins=80, output_bins=513, ssrn_dim=256)
    m.apply(init_weights)
    m.eval()
    torch.manual_seed(0)
    x = torch.rand(1, 80, 200)
    with torch.no_grad():
        y = m(x)
    yn = _np(y)
    np.savez_compressed(os.path.join(OUT, "ssrn_full.npz"),
                        x_sha=np.array(_sha(_np(x))), y_sum=np.array(yn.astype(np.float64).sum()),
                        y_abs=np.array(np.abs(yn.astype(np.float64)).sum()),
                        y_slice=yn[0, ::8, ::8].copy(), w_seed=np.array(1234), x_seed=np.array(0))


def gen_gaw():
    W = _guided(186, 325)
    idx_n = np.array([0, 1, 7, 93, 185, 185, 0, 50, 120], dtype=np.int64)
    idx_t = np.array([0, 3, 300, 162, 324, 0, 324, 88, 210], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "gaw.npz"), n=idx_n, t=idx_t, w=_np(W)[idx_n, idx_t],
                        total=np.array(_np(W).astype(np.float64).sum()), row93=_np(W)[93].copy())


def _import_ge2e(ref):
    """GE2E/hparam.py opens 'config/config.yaml' relative to cwd and calls yaml.load_all
    without a Loader (TypeError on PyYAML>=6); GE2E/utils.py imports librosa (absent).
    Give both what they need from the outside; the reference files stay untouched."""
    import yaml
    orig = yaml.load_all
    yaml.load_all = lambda s, Loader=yaml.SafeLoader: orig(s, Loader=Loader)
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    cwd = os.getcwd()
    os.chdir(os.path.join(ref, "GE2E"))
    sys.path.insert(0, os.path.join(ref, "GE2E"))
    try:
        import hparam as hpmod
        import speech_embedder_net as net
        import utils as ge2e_utils
    finally:
        os.chdir(cwd)
        yaml.load_all = orig
    return hpmod.hparam, net, ge2e_utils


def gen_ge2e(ref):
    hp, net, U = _import_ge2e(ref)
    full = (hp.model.hidden, hp.model.proj, hp.model.num_layer, hp.data.nmels)
    hp.model.hidden, hp.model.proj = 32, 16
    torch.manual_seed(77)
    m = net.SpeechEmbedder()
    with torch.no_grad():   # biases are initialised to 0 by the reference; exercise them
        for n, p in m.LSTM_stack.named_parameters():
            if "bias" in n:
                p.uniform_(-0.2, 0.2)
    x = torch.randn(6, 10, hp.data.nmels)
    with torch.no_grad():
        e = m(x)
        hs, _ = m.LSTM_stack(x)
    out = dict(x=_np(x), e=_np(e), h_last_layer=_np(hs), full_dims=np.array(full, dtype=np.int64))
    out.update(_sd_np(m.state_dict()))
    np.savez_compressed(os.path.join(OUT, "ge2e_embedder.npz"), **out)
    hp.model.hidden, hp.model.proj = full[0], full[1]

    # loss: random (N=4, M=3, D=8) unit vectors, w=10, b=-5 (speech_embedder_net.py:39-40)
    torch.manual_seed(5)
    emb = torch.randn(4, 3, 8)
    emb = (emb / emb.norm(dim=2, keepdim=True)).requires_grad_(True)
    L = net.GE2ELoss(torch.device("cpu"))
    loss = L(emb)
    loss.backward()
    cos = U.get_cossim(emb.detach(), U.get_centroids(emb.detach()))
    # known-answer case carried by the reference at utils.py:89-96 (w=1, b=0)
    kat = torch.tensor([[0, 1, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0]]).float().reshape(3, 2, 3)
    kc = U.get_cossim(kat, U.get_centroids(kat))
    kl, kper = U.calc_loss(1.0 * kc + 0.0)
    np.savez_compressed(os.path.join(OUT, "ge2e_loss.npz"), emb=_np(emb), loss=_np(loss), cossim=_np(cos),
                        demb=_np(emb.grad), dw=_np(L.w.grad), db=_np(L.b.grad),
                        kat_emb=_np(kat), kat_cossim=_np(kc), kat_loss=_np(kl), kat_per=_np(kper))


def gen_ge2e_train(ref):
    """G10: one training iteration of GE2E/train_speech_embedder.py:70-86 with the reference's own SpeechEmbedder and
    GE2ELoss at reduced dims (hidden 32, proj 16; N=4 speakers x M=3 utterances x 10 frames): loss, the gradient of every
    parameter before clipping, and the parameters after clip_grad_norm_(3.0 / 1.0) + SGD(lr=0.01)."""
    hp, net, U = _import_ge2e(ref)
    full = (hp.model.hidden, hp.model.proj)
    hp.model.hidden, hp.model.proj = 32, 16
    torch.manual_seed(91)
    m = net.SpeechEmbedder()
    L = net.GE2ELoss(torch.device("cpu"))
    with torch.no_grad():
        for n, p in m.LSTM_stack.named_parameters():
            if "bias" in n:
                p.uniform_(-0.2, 0.2)
    N, M, T = 4, 3, 10
    x = torch.randn(N * M, T, hp.data.nmels)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    opt = torch.optim.SGD([{"params": m.parameters()}, {"params": L.parameters()}], lr=0.01)   # train_speech_embedder.py:54-57
    m.train()
    opt.zero_grad()
    emb = m(x)
    emb3 = torch.reshape(emb, (N, M, emb.size(1)))
    loss = L(emb3)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    gw, gb = L.w.grad.clone(), L.b.grad.clone()
    n1 = torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0)
    n2 = torch.nn.utils.clip_grad_norm_(L.parameters(), 1.0)
    opt.step()
    # verification-against-enrollment similarity as GE2E/train_speech_embedder.py:156-159 computes it (utils.get_cossim with
    # centroids of ANOTHER set: the own-speaker column still uses the leave-one-out centroid of the verification set)
    torch.manual_seed(92)
    ver = torch.randn(4, 5, 8); ver = ver / ver.norm(dim=2, keepdim=True)
    enr = torch.randn(4, 3, 8); enr = enr / enr.norm(dim=2, keepdim=True)
    ev_cent = U.get_centroids(enr)
    ev_sim = U.get_cossim(ver, ev_cent)
    out = dict(x=_np(x), dims=np.array([N, M, T, 32, 16], dtype=np.int64), loss=_np(loss.detach()), emb=_np(emb.detach()),
               ev_ver=_np(ver), ev_enr=_np(enr), ev_cent=_np(ev_cent), ev_sim=_np(ev_sim),
               w0=np.float32(10.0), b0=np.float32(-5.0), dw=_np(gw), db=_np(gb), w1=_np(L.w.detach()), b1=_np(L.b.detach()),
               norm_net=_np(n1), norm_loss=_np(n2))
    out.update({"p0/" + k: _np(v) for k, v in sd0.items()})
    out.update({"g/" + k: _np(v) for k, v in grads.items()})
    out.update({"p1/" + k: _np(v.detach()) for k, v in m.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "ge2e_train.npz"), **out)
    hp.model.hidden, hp.model.proj = full


def gen_adversarial(TTS, ref):
    """G8: one generator iteration and one critic iteration of train/adversarial_wasserstein_gp.py:261-322, executed
    with the reference's own melSyn and melDisc modules and its loss expressions, at reduced dims.  The critic is put in
    eval mode (dropout off) and the gradient-penalty coefficients are stored, so the iteration is reproducible by another
    implementation; Adam as configured in config.json."""
    import torch.nn.functional as F
    import importlib
    D = importlib.import_module("models.discriminator")
    torch.manual_seed(31)
    hidden, temb, B, N, T = 16, 8, 2, 12, 20
    m = TTS.melSyn(vocab_len=34, condition=True, spkemb_dim=200, textemb_dim=temb, freq_bins=80, hidden_dim=hidden)
    d = D.melDisc(freq_bins=80, disc_dim=16)
    m.apply(init_weights); d.apply(init_weights)
    m.train(); d.eval()
    sd_m0, sd_d0 = {k: v.clone() for k, v in m.state_dict().items()}, {k: v.clone() for k, v in d.state_dict().items()}
    opt_syn = torch.optim.Adam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    opt_disc = torch.optim.Adam(d.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    mel_gt, text, spk = _t2m_inputs(B, N, T, seed=17)
    gaw = _guided(24, 32)
    spec_inputs = torch.cat((torch.zeros_like(mel_gt[:, :, :1]), mel_gt[:, :, :-1]), dim=-1)
    # ---- G iteration (:278-297)
    pred, att = m(spec_inputs, text, spk)
    disc_syn = d(pred)
    l1 = torch.mean(torch.abs(mel_gt - pred))
    bd = torch.mean(-mel_gt * torch.log(pred + 1e-8) - (1 - mel_gt) * torch.log(1 - pred + 1e-8))
    aug = F.pad(att, (0, 32 - att.size()[-1], 0, 24 - att.size()[-2]), value=-1)
    la = torch.sum(torch.ne(aug, -1).float() * aug * gaw) / torch.sum(torch.ne(aug, -1).float())
    ld = torch.mean(-disc_syn)
    loss = l1 + bd + la + (l1.item() + bd.item() + la.item()) / (abs(ld.item())) * ld
    loss.backward()
    opt_syn.step()
    out = dict(mel_gt=_np(mel_gt), text=_np(text), spk=_np(spk), gaw=_np(gaw),
               g_l1=_np(l1), g_bd=_np(bd), g_att=_np(la), g_disc=_np(ld), g_all=_np(loss),
               dims=np.array([hidden, temb, B, N, T], dtype=np.int64))
    out.update(_sd_np(sd_m0, "m0/")); out.update(_sd_np(sd_d0, "d0/"))
    out.update(_sd_np(m.state_dict(), "m1/"))
    # ---- D iteration (:299-322) on the updated generator
    opt_syn.zero_grad(); opt_disc.zero_grad()
    pred, att = m(spec_inputs, text, spk)
    coeff_b = torch.rand(B)
    C = mel_gt.shape[1]
    coeff = torch.stack(T * [torch.stack(C * [coeff_b], dim=1)], dim=2)
    mid = coeff * mel_gt.detach() + (1 - coeff) * pred.detach()
    mid.requires_grad = True
    out_mid = d(mid)
    grads = torch.autograd.grad(outputs=out_mid, inputs=mid, grad_outputs=torch.ones(out_mid.size()), retain_graph=True, create_graph=True)[0]
    gp = torch.mean(10 * (torch.norm(grads, p=2, dim=(1, 2)) - 1) ** 2)
    gp.backward()
    loss_D = torch.mean(d(pred.detach()) - d(mel_gt.detach()))
    loss_D.backward()
    out.update(_grads_np(d, "dgrad/"))          # critic gradients (gradient penalty + Wasserstein terms) before the step
    opt_disc.step()
    out.update(dict(coeff=_np(coeff_b), d_gp=_np(gp), d_loss=_np(loss_D)))
    out.update(_sd_np(d.state_dict(), "d1/"))
    np.savez_compressed(os.path.join(OUT, "adversarial_iter.npz"), **out)


def gen_critic_dropout(ref):
    """G11: the critic side of one D iteration (train/adversarial_wasserstein_gp.py:300-316) with the reference's own
    melDisc and linDisc IN TRAINING MODE -- the reference never calls disc.eval(), so dropout (p = 0.05, three sites per
    critic call, three calls per iteration) is always active.  Dropout draws from the CPU generator: the seed set right
    before the iteration is stored, and a restatement that issues its dropouts at the same sites in the same order
    reproduces the masks.  Stored: inputs, parameters, the two losses, every critic gradient, and the first call's output."""
    import importlib
    D = importlib.import_module("models.discriminator")
    out = {}
    for tag, ctor, (B, Fb, T) in (("mel", lambda: D.melDisc(freq_bins=80, disc_dim=16), (3, 80, 24)),
                                  ("lin", lambda: D.linDisc(freq_bins=65, disc_dim=16), (2, 65, 64))):
        torch.manual_seed(41)
        d = ctor()
        d.apply(init_weights)
        randomize_ln(d, torch.Generator().manual_seed(5))
        d.train()
        g = torch.Generator().manual_seed(19)
        gt, pred = torch.rand(B, Fb, T, generator=g), torch.rand(B, Fb, T, generator=g)
        coeff_b = torch.rand(B, generator=g)
        out.update(_sd_np(d.state_dict(), tag + "/sd/"))
        out.update({tag + "/gt": _np(gt), tag + "/pred": _np(pred), tag + "/coeff": _np(coeff_b)})
        seed = 1000 + len(tag)
        torch.manual_seed(seed)
        coeff = torch.stack(T * [torch.stack(Fb * [coeff_b], dim=1)], dim=2)
        mid = coeff * gt.detach() + (1 - coeff) * pred.detach()
        mid.requires_grad = True
        out_mid = d(mid)
        grads = torch.autograd.grad(outputs=out_mid, inputs=mid, grad_outputs=torch.ones(out_mid.size()), retain_graph=True, create_graph=True)[0]
        gp = torch.mean(10 * (torch.norm(grads, p=2, dim=(1, 2)) - 1) ** 2)
        gp.backward()
        disc_gt = d(gt.detach())
        disc_syn = d(pred.detach())
        loss_D = torch.mean(disc_syn - disc_gt)
        loss_D.backward()
        out.update(_grads_np(d, tag + "/grad/"))
        out.update({tag + "/seed": np.array(seed), tag + "/out_mid": _np(out_mid), tag + "/gp": _np(gp), tag + "/loss_d": _np(loss_D),
                    tag + "/dmid": _np(grads)})
    np.savez_compressed(os.path.join(OUT, "critic_dropout.npz"), **out)


def gen_init_pin(TTS):
    out = {}
    for tag, ctor in (("t2m", lambda: TTS.melSyn(vocab_len=34, condition=True, spkemb_dim=200, textemb_dim=128,
                                                   freq_bins=80, hidden_dim=256)),
                      ("ssrn", lambda: TTS.SSRN(freq_bins=80, output_bins=513, ssrn_dim=256))):
        torch.manual_seed(1234)
        m = ctor()
        m.apply(init_weights)
        names, sums, shas = [], [], []
        for k, v in m.state_dict().items():
            a = _np(v)
            names.append(k); sums.append(a.astype(np.float64).sum()); shas.append(_sha(a))
        out[tag + "/names"] = np.array(names)
        out[tag + "/sums"] = np.array(sums)
        out[tag + "/sha"] = np.array(shas)
        out[tag + "/numel"] = np.array(sum(p.numel() for p in m.parameters()))
    np.savez_compressed(os.path.join(OUT, "init_pin.npz"), **out)


def gen_adam():
    torch.manual_seed(3)
    p = torch.nn.Parameter(torch.randn(257))
    opt = torch.optim.Adam([p], 2e-4, (0.5, 0.9), 1e-6)
    out = {"p0": _np(p).copy()}
    for s in range(3):
        g = torch.randn(257)
        p.grad = g.clone()
        opt.step()
        out["g%d" % (s + 1)] = _np(g)
        out["p%d" % (s + 1)] = _np(p).copy()
    np.savez_compressed(os.path.join(OUT, "adam.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None, help="generate a single fixture group (e.g. ge2e_train) and leave the others as they are")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    torch.set_num_threads(1)           # bit-stable reductions for the fixtures
    if args.only == "ge2e_train":
        gen_ge2e_train(args.ref)
        return
    if args.only == "critic_dropout":
        gen_critic_dropout(args.ref)
        return
    import models.TTSModel as TTS
    gen_highway(TTS)
    gen_melsyn_train(TTS)
    gen_melsyn_eval(TTS)
    gen_ssrn(TTS)
    gen_gaw()
    gen_init_pin(TTS)
    gen_adam()
    gen_adversarial(TTS, args.ref)
    gen_critic_dropout(args.ref)
    gen_ge2e(args.ref)
    gen_ge2e_train(args.ref)
    for f in sorted(os.listdir(OUT)):
        print("%-22s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
