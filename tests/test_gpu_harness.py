"""GPU tests of the drop-in harness: the reference's main.py entry points run end to end on the HIP path."""
import json
import os

import pytest
import torch

from _golden import rel_err
from oracle import tts_oracle as TO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(tmp_path, **over):
    cfg = json.load(open(os.path.join(ROOT, "config.json")))
    cfg.update(SRC_ROOT_DIR=str(tmp_path) + os.sep, BATCH_SIZE=2, MAX_TEXT_LEN=24, MAX_FRAME_NUM=40, HIDDEN_DIM=32,
               TEXT_EMB_DIM=16, SSRN_DIM=32, DISC_DIM=16, VAL_EVERY_ITER=2, SYNTHETIC_BATCHES_PER_EPOCH=3,
               TTS_TEXTS=os.path.join(ROOT, "tts_texts.txt"))
    cfg["STFT"] = {"FFT_LENGTH": 128, "HOP_LENGTH": 32}
    cfg.update(over)
    return cfg


def test_ordinary_text2mel_trains_and_checkpoints(tmp_path):
    from spoofsv_amd import harness
    cfg = _cfg(tmp_path, MAX_ITERATIONS=6)
    model, hist = harness.ordinary_train("train_text2mel", "conditional", cfg, current_time="t")
    assert len(hist) == 6 and all(h == h for h in hist)
    ck = os.path.join(str(tmp_path), "checkpoints", "conditional", "not_adversarial", "t", "text2mel_iteration_3.tar.pth")
    payload = torch.load(ck, map_location="cpu")
    assert set(payload) == {"epoch", "iteration", "model_state_dict", "optimizer_state_dict", "loss_val_log"}
    # resume from it (the reference's -R path) and keep going
    cfg2 = _cfg(tmp_path, MAX_ITERATIONS=5)
    harness.ordinary_train("train_text2mel", "conditional", cfg2, resume_checkpoints=ck, current_time="t2")


def test_ordinary_step_matches_cpu_oracle_step(tmp_path):
    """One full optimizer iteration (fwd, losses, bwd, Adam) on the HIP path equals the oracle's."""
    from spoofsv_amd import train
    from spoofsv_amd.tts import melSyn
    torch.manual_seed(7)
    m = melSyn(34, True, 200, textemb_dim=16, freq_bins=80, hidden_dim=32)
    m.apply(train.init_weights)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    mel, text, spk = train.synthetic_text2mel_batch(3, N=24, T=40, seed=2)
    gaw = train.guided_attention_mat(186, 325)
    opt_o = torch.optim.Adam(list(sd.values()), 2e-4, (0.5, 0.9), 1e-6)
    Y, A = TO.melsyn_train(train.shift_right(mel), text, spk, sd)
    l = TO.text2mel_losses(Y, A, mel, gaw)
    sum(l).backward()
    opt_o.step()
    m = m.to("cuda:0").train()
    opt = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    l1, bd, la, _ = train.text2mel_step(m, opt, mel.cuda(), text.cuda(), spk.cuda(), gaw.cuda())
    for mine, ref in zip((l1, bd, la), l):
        assert abs(float(mine) - float(ref)) < 1e-5 * max(1, abs(float(ref)))
    # Adam's first step moves every weight by ~lr*sign(g): compare the UPDATE, which is what the step computed
    worst = 0.0
    for k, p in m.state_dict().items():
        worst = max(worst, float((p.cpu() - sd[k].detach()).abs().max()))
    assert worst < 2e-5, worst          # lr = 2e-4: a sign flip of one update would show as 4e-4


def test_adversarial_ssrn_and_text2mel_iterations(tmp_path):
    from spoofsv_amd import harness
    for step in ("train_text2mel", "train_ssrn"):
        cfg = _cfg(tmp_path, MAX_ITERATIONS=7, RATIO=2)
        model, disc, logs = harness.adversarial_train(step, "conditional", cfg, current_time="adv_" + step)
        assert len(logs["loss_train_log_syn"]) == 3 and len(logs["loss_train_log_disc"]) == 4
        assert all(v == v for v in logs["loss_train_log_syn"] + logs["loss_train_log_disc"])
        ck = os.path.join(str(tmp_path), "checkpoints", "conditional", "adversarial", "adv_" + step,
                          "{}_iteration_3.tar.pth".format(step[6:]))
        keys = set(torch.load(ck, map_location="cpu"))
        assert {"model_state_dict", "disc_state_dict", "opt_state_dict_syn", "opt_state_dict_disc", "wd_log"} <= keys


def test_adversarial_train_captured_graphs(tmp_path):
    from spoofsv_amd import harness
    cfg = _cfg(tmp_path, MAX_ITERATIONS=7, RATIO=2, CAPTURE_GRAPHS=True)
    model, disc, logs = harness.adversarial_train("train_text2mel", "conditional", cfg, current_time="cap")
    assert len(logs["loss_train_log_syn"]) == 3 and len(logs["loss_train_log_disc"]) == 4
    assert all(v == v for v in logs["loss_train_log_syn"] + logs["loss_train_log_disc"])
    ck = os.path.join(str(tmp_path), "checkpoints", "conditional", "adversarial", "cap", "text2mel_iteration_3.tar.pth")
    assert os.path.exists(ck)


def test_generator_accepts_critic_gradient_like_oracle():
    """dL/dY coming from the critic flows through the HIP generator exactly as through the oracle."""
    from spoofsv_amd import train
    from spoofsv_amd.critic import linDisc
    from spoofsv_amd.tts import SSRN
    torch.manual_seed(11)
    g = SSRN(80, 65, 16)
    g.apply(train.init_weights)
    from oracle import critic_oracle as CO
    d = linDisc(65, 16).eval()                     # eval: no dropout, so both arms see the same critic
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in g.state_dict().items()}
    mel = torch.rand(2, 80, 16)
    (-CO.critic(TO.ssrn(mel, sd), d.state_dict(), "lin", masks=False)).mean().backward()
    g = g.to("cuda:0").train()
    d = d.to("cuda:0")
    (-d(g(mel.cuda()))).mean().backward()
    bad = {k: rel_err(p.grad, sd[k].grad) for k, p in g.named_parameters() if rel_err(p.grad, sd[k].grad) > 3e-4}
    assert not bad, bad


def test_validation_free_run_follows_weight_updates():
    """The trainers validate the LIVE model through the cached column-incremental synthesizer (harness.validate).  FusedAdam
    updates weights through raw pointers, so nothing bumps a version counter: the synthesizer's tap-major copies of the k=3
    weights and the resident split planes must still follow.  Validate, take optimizer steps, validate again (same shapes ->
    cached synthesizer, captured graph) and compare with the reference's prefix loop on the current weights."""
    from spoofsv_amd import harness, train
    from spoofsv_amd.tts import melSyn
    torch.manual_seed(3)
    m = melSyn(34, True, 200, textemb_dim=16, freq_bins=80, hidden_dim=32)
    m.apply(train.init_weights)
    m = m.to("cuda:0")
    opt = train.FusedAdam(m.parameters(), 5e-3, (0.5, 0.9), 1e-6)             # a large step: stale weights would show clearly
    mel, text, spk = train.synthetic_text2mel_batch(4, N=24, T=40, seed=2, device="cuda:0")
    gaw = train.guided_attention_mat(24, 40, device="cuda:0")

    def both():
        m.eval()
        with torch.no_grad():
            yi, ai = harness._free_run(m, text, spk, 12, 80, incremental=True)
            yp, ap = harness._free_run(m, text, spk, 12, 80, incremental=False)
        m.train()
        return yi, yp, ai, ap
    y0i, y0p, _, _ = both()
    assert rel_err(y0i, y0p) < 2e-4
    for _ in range(3):
        train.text2mel_step(m, opt, mel, text, spk, gaw)
    y1i, y1p, a1i, a1p = both()
    assert rel_err(y1p, y0p) > 1e-2                                            # the weights really moved
    assert rel_err(y1i, y1p) < 2e-4 and rel_err(a1i, a1p) < 2e-4, (rel_err(y1i, y1p), rel_err(a1i, a1p))


def test_main_cli_synthesize(tmp_path):
    import main as cli_main
    cfg = _cfg(tmp_path)
    path = os.path.join(str(tmp_path), "cfg.json")
    json.dump(cfg, open(path, "w"))

    class A:
        step, pattern, resume, configuration, adversarial, save_spectrogram, current_time = \
            "synthesize", "conditional", None, path, False, False, "syn"
    cfg["MAX_FRAME_NUM"] = 12
    json.dump(cfg, open(path, "w"))
    outs = cli_main.run(A)
    mel, lin, att = outs[0]
    assert mel.shape == (80, 12) and lin.shape == (65, 48) and att.shape == (43, 12)
    assert os.path.exists(os.path.join(str(tmp_path), "samples", "syn", "S1_lin.npy"))
    from scipy.io import wavfile                    # vocoder tail, synthesize.py:138-147
    sr, wav = wavfile.read(os.path.join(str(tmp_path), "samples", "syn", "S1_B1.wav"))
    assert sr == cfg["SAMPLING_RATE"] and wav.dtype.kind == "f" and wav.shape == (32 * 47,)
    assert abs(float(wav.max()) - 0.75) < 1e-5 and (wav == wav).all()


def test_extract_features_writes_the_reference_cache_layout(tmp_path):
    """data/dataset.py:94-123 with the spectrogram front end on the GPU: wav -> <spec_dir>/pXXX/pXXX_NNN_{mel,lin}.npy, values as
    the numpy oracle computes them, and BatchSource picks the cache up for SSRN training."""
    import json as _json
    import numpy as np
    from scipy.io import wavfile
    from oracle import vocoder_oracle as vo
    from spoofsv_amd import harness
    from spoofsv_amd.vocoder import trim_silence
    cfg = _json.load(open(os.path.join(ROOT, "config.json")))
    rng = np.random.RandomState(5)
    sr = cfg["SAMPLING_RATE"]
    tone = (0.3 * np.sin(2 * np.pi * 330 * np.arange(sr) / sr) * np.hanning(sr) + 0.01 * rng.randn(sr)).astype(np.float32)
    wav = np.concatenate([np.zeros(3000, np.float32), tone, np.zeros(2000, np.float32)])
    wdir = os.path.join(str(tmp_path), "wav22", "p225")
    os.makedirs(wdir)
    path = os.path.join(wdir, "p225_001.wav")
    wavfile.write(path, sr, (wav * 32767).astype(np.int16))
    spec_dir = os.path.join(str(tmp_path), "spec") + os.sep
    shapes = harness.extract_features([path], cfg, spec_dir)
    mel = np.load(os.path.join(spec_dir, "p225", "p225_001_mel.npy"))
    lin = np.load(os.path.join(spec_dir, "p225", "p225_001_lin.npy"))
    assert shapes == [(mel.shape, lin.shape)] and mel.shape[0] == 80 and lin.shape == (513, 4 * mel.shape[1])
    y = (wav * 32767).astype(np.int16).astype(np.float32) / 32768.0
    y, _ = trim_silence(y, 22)
    mr, lr = vo.wav2spectrogram(y, sr, cfg)
    assert np.abs(mel - mr).max() <= 5e-5 and np.abs(lin - lr).max() <= 5e-5
    src = harness.BatchSource(dict(cfg, BATCH_SIZE=2), "train_ssrn", 2, spec_dir)
    batch = next(iter(src))
    assert tuple(batch["data_0"].shape) == (2, 80, mel.shape[1]) and tuple(batch["data_1"].shape) == (2, 513, 4 * mel.shape[1])


def test_training_from_a_corpus_in_the_reference_layout(tmp_path):
    """wav files + path lists + transcripts + speaker codes (data/dataset.py:37-51): features are extracted once on the GPU into
    the reference's cache layout, then ordinary_train runs on real, variable-length batches for both models."""
    from test_host_cpu import _make_corpus
    from spoofsv_amd import harness
    cfg, _ = _make_corpus(str(tmp_path), n_items=5, with_cache=False, wav=True)
    cfg.update(BATCH_SIZE=2, HIDDEN_DIM=32, TEXT_EMB_DIM=16, SSRN_DIM=32, VAL_EVERY_ITER=2, MAX_ITERATIONS=4)
    spec = os.path.join(str(tmp_path), "spec") + os.sep
    model, hist = harness.ordinary_train("train_text2mel", "conditional", cfg, spec_dir=spec, current_time="c")
    assert len(hist) == 4 and all(h == h for h in hist)
    assert len([f for _, _, fs in os.walk(spec) for f in fs if f.endswith("_mel.npy")]) == 5      # every file cached once
    ck = os.path.join(cfg["SRC_ROOT_DIR"], "checkpoints", "conditional", "not_adversarial", "c")
    val = torch.load(os.path.join(ck, "text2mel_iteration_3.tar.pth"), map_location="cpu")["loss_val_log"]
    assert len(val) == 1 and val[0] == val[0] and os.path.exists(os.path.join(ck, "text2mel_best_model.tar.pth"))   # free-running validation ran
    model, hist = harness.ordinary_train("train_ssrn", "conditional", cfg, spec_dir=spec, current_time="c")
    assert len(hist) == 4 and all(h == h for h in hist)
    # synthesize.py proper: the 'synthesize' split in batches of 8, losses against the ground truth, every item vocoded
    outs = harness.synthesize("conditional", dict(cfg, GRIFFIN_LIM_ITERS=4), spec, current_time="s")
    assert len(outs) == 5 and outs[0][0].shape[0] == 80 and outs[0][1].shape == (513, 4 * outs[0][0].shape[1])
    assert sorted(f for f in os.listdir(os.path.join(cfg["SRC_ROOT_DIR"], "samples", "s")) if f.endswith(".wav")) == \
        ["S%d_B1.wav" % (k + 1) for k in range(5)]


def test_generate_test_utterances_writes_every_speakers_batch(tmp_path):
    """generate_test_utterances.py:56-139 on the HIP path: one batched free run + SSRN + vocoder per speaker."""
    import numpy as np
    from scipy.io import wavfile
    from spoofsv_amd import harness
    cfg = _cfg(tmp_path, MAX_FRAME_NUM=9, GRIFFIN_LIM_ITERS=4, SYNTH_INCREMENTAL=True)
    spk = {"p225": np.full(200, 0.05, np.float32), "p301": np.linspace(-0.1, 0.1, 200).astype(np.float32)}
    texts = ["The birch canoe slid.", "Glue the sheet.", "It's easy to tell the depth of a well."]
    out = harness.generate_test_utterances(cfg, "gen", eval_utt_num=3, speakers=spk, texts=texts)
    assert sorted(out) == ["p225", "p301"] and all(len(v) == 3 for v in out.values())
    assert out["p301"][2].endswith(os.path.join("test", "gen", "spoof_data", "s301", "s301_003.wav"))
    for paths in out.values():
        for p in paths:
            sr, wav = wavfile.read(p)
            assert sr == cfg["SAMPLING_RATE"] and 0 < len(wav) <= 32 * (4 * 10 - 1) and abs(float(wav.max()) - 0.75) < 1e-5


def test_adversarial_graph_step_matches_eager_generator_iteration():
    """The captured G iteration equals the eager one (same weights, same batch, critic in eval so no dropout RNG)."""
    from spoofsv_amd import ops, train
    from spoofsv_amd.critic import melDisc
    from spoofsv_amd.tts import melSyn
    dev = "cuda:0"

    def build():
        torch.manual_seed(21)
        m = melSyn(34, True, 200, textemb_dim=16, freq_bins=80, hidden_dim=32)
        d = melDisc(80, 16)
        m.apply(train.init_weights); d.apply(train.init_weights)
        return m.to(dev).train(), d.to(dev).eval()
    # B*T = 256 >= 128: the convolutions run on the split-bf16 kernels with RESIDENT weight planes, whose addresses the
    # captured graphs bake in -- restoring weights behind them must be followed by a refresh (harness does the same)
    batch = train.synthetic_text2mel_batch(4, N=24, T=64, seed=4, device=dev)
    gaw = train.guided_attention_mat(186, 325, device=dev)
    # eager, as the reference spells it (host-side adaptive weight)
    m, d = build()
    opt = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    mel, text, spk = batch
    pred, att = m(train.shift_right(mel), text, spk)
    l1, bd = ops.spec_losses(pred, mel)
    la = ops.guided_att_loss(att, gaw)
    ld = torch.mean(-d(pred))
    loss = l1 + bd + la + (float(l1) + float(bd) + float(la)) / abs(float(ld)) * ld
    loss.backward()
    opt.step()
    ref = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # captured
    m2, d2 = build()
    og = train.FusedAdam(m2.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(d2.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    w0 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
    c0 = {k: v.detach().clone() for k, v in d2.state_dict().items()}
    stepper = train.AdversarialGraphStep("text2mel", m2, d2, og, od, batch, gaw)
    # construction ran warm-up iterations of both kinds: restore the initial weights (generator AND critic) and the
    # generator's optimizer moments, then replay ONE G step
    m2.load_state_dict(w0)
    d2.load_state_dict(c0)
    for st in og.state.values():
        st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    og._step_dev.zero_()
    og.refresh_resident_weights(); od.refresh_resident_weights()
    out = stepper.g_step()
    assert abs(float(out[4]) - float(loss)) < 1e-4 * abs(float(loss))
    worst = max(float((m2.state_dict()[k] - ref[k]).abs().max()) for k in ref)
    assert worst < 2e-5, worst
    ld_, gp_ = stepper.d_step()
    assert float(ld_) == float(ld_) and float(gp_) == float(gp_)

# Biases that feed a LayerNorm have an exactly-zero true gradient; what autograd returns for them is rounding noise.
# They are excluded; every other critic gradient must agree in the relative L2 norm.  (Post-step WEIGHTS are not
# compared for the critic: Adam's first step maps every gradient to +-lr, so near-zero entries flip on rounding noise.)
_ZERO_GRAD = ("conv1.bias", "hc.conv.bias", "conv2.bias", "conv3.bias", "conv4.bias", "conv5.bias")


def _critic_grads_agree(module, want, tol):
    for k, p in module.named_parameters():
        if k in _ZERO_GRAD:
            continue
        a, b = p.grad.detach().cpu().double(), want[k].double()
        assert float((a - b).norm() / b.norm()) < tol, (k, float((a - b).norm() / b.norm()))


def test_adversarial_iteration_golden_gpu():
    """G8 on the GPU: HIP generator + HIP-op critic reproduce the reference's G and D iterations (losses, weights)."""
    from _golden import load, sub, t
    from spoofsv_amd import ops, train
    from spoofsv_amd.critic import melDisc
    from spoofsv_amd.tts import melSyn
    g = load("adversarial_iter.npz")
    hidden, temb, B, N, T = [int(v) for v in g["dims"]]
    dev = "cuda:0"
    m = melSyn(34, True, 200, textemb_dim=temb, freq_bins=80, hidden_dim=hidden)
    m.load_state_dict(sub(g, "m0/"))
    d = melDisc(80, 16)
    d.load_state_dict(sub(g, "d0/"))
    m, d = m.to(dev).train(), d.to(dev).eval()
    mel, text, spk, gaw = t(g["mel_gt"], dev), t(g["text"], dev), t(g["spk"], dev), t(g["gaw"], dev)
    opt = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    pred, att = m(train.shift_right(mel), text, spk)
    l1, bd = ops.spec_losses(pred, mel)
    la = ops.guided_att_loss(att, gaw)
    ld = torch.mean(-d(pred))
    loss = l1 + bd + la + (float(l1) + float(bd) + float(la)) / abs(float(ld)) * ld
    for mine, ref in ((l1, "g_l1"), (bd, "g_bd"), (la, "g_att"), (ld, "g_disc"), (loss, "g_all")):
        assert abs(float(mine) - float(g[ref])) < 1e-5 * max(1.0, abs(float(g[ref]))), (ref, float(mine), float(g[ref]))
    loss.backward()
    opt.step()
    m1 = sub(g, "m1/")
    worst = max(float((m.state_dict()[k].cpu() - m1[k]).abs().max()) for k in m1)
    assert worst < 5e-5, worst                       # one Adam step moves weights by ~2e-4
    od = torch.optim.Adam(d.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    m.load_state_dict(m1)
    with torch.no_grad():
        pred, _ = m(train.shift_right(mel), text, spk)
    d.zero_grad()          # the G iteration left gradients on the critic; the reference zeroes both optimizers (:264-265)
    coeff = t(g["coeff"], dev).view(-1, 1, 1)
    mid = (coeff * mel + (1 - coeff) * pred).requires_grad_(True)
    out = d(mid)
    grads = torch.autograd.grad(out, mid, torch.ones_like(out), retain_graph=True, create_graph=True)[0]
    gp = torch.mean(10 * (torch.norm(grads, p=2, dim=(1, 2)) - 1) ** 2)
    gp.backward()
    loss_d = torch.mean(d(pred) - d(mel))
    loss_d.backward()
    assert abs(float(gp) - float(g["d_gp"])) < 1e-3 * max(1.0, abs(float(g["d_gp"])))
    assert abs(float(loss_d) - float(g["d_loss"])) < 1e-4 * max(1.0, abs(float(g["d_loss"])))
    _critic_grads_agree(d, sub(g, "dgrad/"), 2e-3)


@pytest.mark.gpu
def test_resident_weights_match_per_call_split_and_follow_weight_updates():
    """Resident pre-split planes (one refresh launch) give bit-identical results to the per-call split, and a weight
    changed through torch stops using them until the next refresh."""
    from spoofsv_amd import resident, train
    from spoofsv_amd.tts import highwayConv
    torch.manual_seed(5)
    m = highwayConv(64, 3, 3, causal=True).to("cuda")
    x = torch.randn(4, 64, 200, device="cuda")
    dy = torch.randn(4, 64, 200, device="cuda")

    def run():
        xg = x.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(xg)
        y.backward(dy)
        return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    w = m.conv.weight
    assert resident.lookup(w) is None
    base = run()
    opt = train.FusedAdam(m.parameters(), 1e-3)
    opt.refresh_resident_weights()
    assert resident.lookup(w) is not None
    res = run()
    for a, b in zip(base, res):
        assert torch.equal(a, b)
    with torch.no_grad():
        w.mul_(1.5)                                   # in-place torch op: version bump -> planes are stale
    assert resident.lookup(w) is None
    changed = run()
    assert not torch.equal(changed[0], base[0])
    opt.refresh_resident_weights()
    assert resident.lookup(w) is not None
    again = run()
    for a, b in zip(changed, again):
        assert torch.equal(a, b)
    for p in m.parameters():                          # an optimizer step keeps the planes current
        p.grad = torch.randn_like(p)
    opt.step()
    assert resident.lookup(w) is not None
    after = run()
    resident.invalidate([w])
    assert resident.lookup(w) is None
    ref = run()
    for a, b in zip(after, ref):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adam_state_roundtrip_continues_bias_correction(capturable):
    """A FusedAdam restored from its state_dict takes exactly the step the original would have taken next
    (step count, moments), and its state_dict carries torch-Adam style ``step`` entries."""
    from spoofsv_amd import train
    torch.manual_seed(11)
    p0 = torch.randn(1000, device="cuda")
    grads = [torch.randn(1000, device="cuda") for _ in range(5)]

    def make():
        p = torch.nn.Parameter(p0.clone())
        return p, train.FusedAdam([p], 1e-2, (0.5, 0.9), 1e-6, capturable=capturable)

    pa, oa = make()
    for g in grads:
        pa.grad = g.clone()
        oa.step()
    pb, ob = make()
    for g in grads[:3]:
        pb.grad = g.clone()
        ob.step()
    sd = ob.state_dict()
    assert float(sd["state"][0]["step"]) == 3.0
    pc = torch.nn.Parameter(pb.detach().clone())
    oc = train.FusedAdam([pc], 1e-2, (0.5, 0.9), 1e-6, capturable=capturable)
    oc.load_state_dict(sd)
    for g in grads[3:]:
        pc.grad = g.clone()
        oc.step()
    assert torch.equal(pc, pa)
    assert float(oc.state_dict()["state"][0]["step"]) == 5.0


@pytest.mark.gpu
def test_graph_synthesis_matches_step_by_step_loop():
    """The replayed fixed-shape synthesis step (spoofsv_amd/synth.py) against the reference-shaped loop: attention
    arg-max path identical, spectrogram within the split-bf16 tolerance (short prefixes use exact fp32 step by step)."""
    from spoofsv_amd import harness, train
    from spoofsv_amd.tts import melSyn
    torch.manual_seed(2017)
    m = melSyn(34, True, 200, textemb_dim=16, freq_bins=80, hidden_dim=32)
    m.apply(train.init_weights)
    m = m.to("cuda").eval()
    text = torch.randint(2, 33, (2, 1, 12), device="cuda")
    spk = 0.04 + 0.05 * torch.rand(2, 200, 1, device="cuda")
    with torch.no_grad():
        Y0, A0 = harness._free_run(m, text, spk, 24, 80)
        Y1, A1 = harness._free_run(m, text, spk, 24, 80, graph=True)
        Y2, A2 = harness._free_run(m, text, spk, 24, 80, graph=True)      # second run replays the cached graph
    assert torch.equal(Y1, Y2) and torch.equal(A1, A2)
    assert torch.equal(A0.argmax(dim=1), A1.argmax(dim=1))
    assert rel_err(Y1, Y0) < 1e-3 and rel_err(A1, A0) < 1e-3, (rel_err(Y1, Y0), rel_err(A1, A0))


@pytest.mark.gpu
def test_ge2e_harness_trains_checkpoints_and_verifies(tmp_path):
    """GE2E/train_speech_embedder.py end to end on synthetic preprocessed TI-SV data: train (HIP forward + backward),
    checkpoint, then the mixture / no-spoof verification tests and the spoof-rate pass over the saved matrices."""
    import numpy as np
    from spoofsv_amd import ge2e_harness as GH
    # (runs in the default arithmetic; an earlier version switched the process to split-bf16 here and never switched back, so every test
    # that ran after it in the same process -- e.g. the single-process arm of the data-parallel comparisons -- ran in another mode)
    rng = np.random.RandomState(3)
    for split, nspk in (("train", 8), ("test", 4)):
        d = tmp_path / split
        d.mkdir()
        for s in range(nspk):
            base = rng.randn(1, 40, 1).astype(np.float32)           # a speaker-specific spectral offset
            np.save(d / ("spk%02d.npy" % s), base + 0.3 * rng.randn(16, 40, 24).astype(np.float32))
    cfg = GH.default_config()
    cfg["data"].update(train_path=str(tmp_path / "train"), test_path=str(tmp_path / "test"))
    cfg["model"].update(hidden=32, proj=16)
    cfg["train"].update(N=4, M=4, epochs=3, log_interval=1, checkpoint_interval=2, checkpoint_dir=str(tmp_path / "ckpt"))
    cfg["test"].update(N=4, M=12, epochs=1)
    cfg["save_simmat_dir"] = str(tmp_path / "simmat")
    torch.manual_seed(0)
    net, hist = GH.train(cfg)
    assert len(hist) == 6 and all(np.isfinite(hist))
    files = sorted(os.listdir(tmp_path / "ckpt"))
    assert any(f.startswith("ckpt_epoch_2") for f in files) and any(f.endswith(".model") for f in files)
    model_path = str(tmp_path / "ckpt" / [f for f in files if f.endswith(".model")][0])
    eer, spoof = GH.test(cfg, model_path, enroll_num=2)
    thres = GH.test_nospoof(cfg, model_path, enroll_num=2, eval_num=2)
    rate = GH.spoof_rate_at(cfg, thres, eval_num=2)
    assert 0.0 <= eer <= 1.0 and 0.0 <= spoof <= 1.0 and 0.5 <= thres < 1.0 and 0.0 <= rate <= 1.0


def _t2m_grads(m, mel, text, spk, gaw, ddp=None):
    from spoofsv_amd import train
    for p in m.parameters():
        p.grad = None
    pred, att = m(train.shift_right(mel), text, spk)
    l1, bd, la = train.text2mel_losses(pred, att, mel, gaw)
    (l1 + bd + la).backward()
    if ddp is not None:
        ddp.all_reduce_grads()
    torch.cuda.synchronize()
    return [p.grad.detach().cpu().numpy() for p in m.parameters()]


def _ddp_gpu_worker(rank, world, port, q):
    import torch.distributed as dist
    from spoofsv_amd import train
    from spoofsv_amd.tts import melSyn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"                                       # rehearsal: both ranks share the one GPU of the test box
    torch.manual_seed(100 + rank)                        # replicas start different; broadcast_parameters must fix that
    m = melSyn(34, True, 200, textemb_dim=16, freq_bins=80, hidden_dim=32)
    m.apply(train.init_weights)
    m = m.to(dev).train()
    ddp = train.DataParallelRanks(list(m.parameters()))
    ddp.broadcast_parameters(0)
    opt = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    mel, text, spk = train.synthetic_text2mel_batch(4, N=20, T=33, seed=3, device=dev)
    gaw = train.guided_attention_mat(20, 33, device=dev)
    sl = slice(2 * rank, 2 * rank + 2)                  # each rank takes its half of the global batch
    grads = _t2m_grads(m, mel[sl], text[sl], spk[sl], gaw, ddp)
    for _ in range(2):                                   # and two full optimizer steps: replicas must stay identical
        train.text2mel_step(m, opt, mel[sl], text[sl], spk[sl], gaw, ddp=ddp)
    torch.cuda.synchronize()
    q.put((rank, grads, [p.detach().cpu().numpy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_process_on_the_global_batch():
    """One process per rank: utterances sharded by rank, gradients averaged by the flat-bucket all-reduce.  Two ranks
    with half the batch each must produce the gradient one process computes on the whole batch (every loss is a mean over the
    batch), and their replicas must stay bit-identical through optimizer steps.  gloo here because the test box has one GPU; the
    driver's multi-GPU runs use RCCL through the same code path."""
    import numpy as np
    import torch.multiprocessing as mp
    from spoofsv_amd import train
    from spoofsv_amd.tts import melSyn
    dev = "cuda:0"
    torch.manual_seed(100)                               # = rank 0's seed
    m = melSyn(34, True, 200, textemb_dim=16, freq_bins=80, hidden_dim=32)
    m.apply(train.init_weights)
    m = m.to(dev).train()
    mel, text, spk = train.synthetic_text2mel_batch(4, N=20, T=33, seed=3, device=dev)
    gaw = train.guided_attention_mat(20, 33, device=dev)
    want = _t2m_grads(m, mel, text, spk, gaw)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)                      # replicas bit-identical after two steps
    scale = max(float(np.abs(w).max()) for w in want)
    for g0, g1, w in zip(res[0][1], res[1][1], want):
        assert np.array_equal(g0, g1)
        # split-bf16 products over a different batch partition: agreement to a few 1e-4 of the gradient's own size
        assert float(np.abs(g0 - w).max()) <= 2e-5 * scale + 5e-4 * float(np.abs(w).max())


def _ge2e_setup_gpu():
    from spoofsv_amd.ge2e import GE2ELoss, SpeechEmbedder
    torch.manual_seed(55)
    net = SpeechEmbedder(nmels=40, hidden=32, num_layer=3, proj=16).to("cuda:0").train()
    crit = GE2ELoss("cuda:0")
    x = torch.randn(4, 4, 20, 40, device="cuda:0")                       # N = 4 speakers x M = 4 utterances x 20 frames x 40 mels
    return net, crit, x


def _ge2e_gpu_worker(rank, world, port, q):
    import torch.distributed as dist
    from spoofsv_amd import ge2e
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, crit, x = _ge2e_setup_gpu()
    opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], lr=0.01)
    n_local = x.shape[0] // world
    loss = ge2e.sharded_train_iteration(net, crit, opt, x[rank * n_local:(rank + 1) * n_local], n_local, 4)
    torch.cuda.synchronize()
    q.put((rank, float(loss), [p.detach().cpu().numpy() for p in list(net.parameters()) + list(crit.parameters())]))
    dist.barrier()
    dist.destroy_process_group()


def test_ge2e_speaker_sharded_iteration_on_the_hip_path():
    """SURVEY 8e, GE2E row, on the HIP embedder: speakers split over two ranks, one all-gather of the embeddings, summed embedder
    gradients -> the loss and the post-step weights of the single-process iteration on the whole batch."""
    import numpy as np
    import torch.multiprocessing as mp
    from spoofsv_amd import ge2e
    net, crit, x = _ge2e_setup_gpu()
    opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], lr=0.01)
    before = [p.detach().cpu().numpy().copy() for p in list(net.parameters()) + list(crit.parameters())]
    want_loss = float(ge2e.train_iteration(net, crit, opt, x, 4, 4))
    want = [p.detach().cpu().numpy() for p in list(net.parameters()) + list(crit.parameters())]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ge2e_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, loss, weights in res:
        assert abs(loss - want_loss) <= 1e-4 * abs(want_loss)
        for a, b, b0 in zip(weights, want, before):
            step = float(np.abs(b - b0).max())                          # compare the UPDATE (SGD: lr * clipped gradient)
            assert float(np.abs(a - b).max()) <= 2e-3 * step + 1e-7


def test_integration_md_binding_snippet_runs_as_written():
    """INTEGRATION.md section 2 is executable documentation: the ctypes stub, pasted into a reference-style module, must give the
    same output as this repo's own highwayConv (same weights)."""
    import re
    from spoofsv_amd.tts import highwayConv as mine_cls
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("spoofsv_amd/libssv_hip.so")', 'ctypes.CDLL(%r)' % os.path.join(ROOT, "spoofsv_amd", "libssv_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    torch.manual_seed(12)
    mine = mine_cls(32, 3, 9, causal=True).to("cuda:0").eval()
    doc = ns["highwayConv"]()                                   # the documented class: a torch.nn.Module with a re-routed forward
    doc.conv, doc.ln1, doc.ln2, doc.causal = mine.conv, mine.ln1, mine.ln2, True
    x = torch.randn(3, 32, 70, device="cuda:0")
    with torch.no_grad():
        assert torch.equal(doc(x), mine(x))


@pytest.mark.gpu
def test_captured_step_can_be_released_and_captured_again():
    """TrainStep.prepare() is one-shot (a prepared step ignores further calls); release() drops the hipGraphs and frees the batched weight
    gradients' frozen job tables (ops.DeferredWgrad.release_capture), after which the same step object captures and replays again with the
    same results.  Without release(), forcing a second capture through the same DeferredWgrad is refused loudly."""
    from spoofsv_amd import train
    from spoofsv_amd.tts import SSRN
    torch.manual_seed(3)
    m = SSRN(80, 65, 32)
    m.apply(train.init_weights)
    m = m.to("cuda").train()
    opt = train.FusedAdam(m.parameters(), 0.0, (0.5, 0.9), 1e-6, capturable=True)       # lr 0: every iteration sees the same weights
    opt.refresh_resident_weights()
    batch = train.synthetic_ssrn_batch(4, 40, out_bins=65, seed=0, device="cuda")
    st = train.TrainStep("ssrn", m, opt, batch, None, None, graph=True, defer_wgrad=True).prepare()
    first = [float(v) for v in st()]
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert st.prepare() is st and st.stepper.plan is not None                           # a second prepare() is a no-op
    st.stepper.plan = None                                                              # forcing a re-capture without release(): refused
    with pytest.raises(RuntimeError, match="frozen job tables"):
        st.prepare()
    torch.cuda.synchronize()
    st.release().prepare()
    again = [float(v) for v in st()]
    torch.cuda.synchronize()
    assert again == first
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, g1[k]), k


@pytest.mark.gpu
def test_input_grads_only_skips_the_critics_parameter_gradients():
    """ops.input_grads_only(disc): inside the block the critic's twice-differentiable operators return no parameter gradients (the gradient
    penalty's first pass and the generator iterations ask for the input gradient only) -- pinned here so that a future weight transform in
    front of those operators (a contiguous copy, a cast: the match is by data_ptr) cannot silently bring the discarded GEMMs back."""
    from spoofsv_amd import ops
    from spoofsv_amd.critic import melDisc
    torch.manual_seed(2)
    d = melDisc(80, 32).to("cuda").eval()
    x = torch.rand(3, 80, 40, device="cuda", requires_grad=True)
    out = d(x)
    with ops.input_grads_only(d):
        out.sum().backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0
    inside = {k: p.grad for k, p in d.named_parameters()}
    assert all(g is None for g in inside.values()), [k for k, g in inside.items() if g is not None]
    x2 = x.detach().clone().requires_grad_(True)
    d(x2).sum().backward()                                                              # outside the block: every parameter gets its gradient
    assert all(p.grad is not None for p in d.parameters())
    assert torch.allclose(x2.grad, x.grad, rtol=1e-5, atol=1e-7)
