"""GPU parity of the vocoder / spectrogram front end against oracle/vocoder_oracle.py.

Tolerances: one transform carries the split-bf16 basis error (16 mantissa bits): <= 2e-5 of the peak.  Griffin-Lim is an
iterated projection that amplifies rounding differences, so waveform parity is asserted at few iterations (<= 1e-4 of
the peak after 4) and at 64 iterations through what the algorithm is for: the spectral inconsistency it reaches.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = json.load(open(os.path.join(ROOT, "config.json")))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _pack(c):
    return _dev(np.concatenate([c.real, c.imag], -2))


@pytest.fixture(scope="module")
def voc():
    from spoofsv_amd.vocoder import Vocoder
    return Vocoder(1024, 256)


def _wave(rng, B, n):
    return (rng.randn(B, n) * np.hanning(n)[None] * 0.3).astype(np.float32)


@pytest.mark.parametrize("T", [4, 25, 67])
def test_stft_istft_match_oracle(voc, T):
    rng = np.random.RandomState(T)
    B, n = 2, 256 * (T - 1)
    y = _wave(rng, B, n)
    ref = np.stack([vo.stft(y[b]) for b in range(B)])
    S = voc.stft(_dev(y))
    assert tuple(S.shape) == (B, 1026, T)
    peak = np.abs(ref).max()
    assert np.abs(S.cpu().numpy() - np.concatenate([ref.real, ref.imag], 1)).max() <= 2e-5 * peak
    assert np.abs(voc.magnitude(S).cpu().numpy() - np.abs(ref)).max() <= 2e-5 * peak
    spec = rng.randn(B, 513, T) + 1j * rng.randn(B, 513, T)       # not a consistent STFT: exercises the full overlap-add
    yi = np.stack([vo.istft(spec[b]) for b in range(B)])
    got = voc.istft(_pack(spec)).cpu().numpy()
    assert got.shape == yi.shape and np.abs(got - yi).max() <= 2e-5 * np.abs(yi).max()
    assert np.abs(voc.istft(S).cpu().numpy() - y).max() <= 2e-5 * np.abs(y).max()    # exact reconstruction


def test_full_size_batch_round_trip_and_consistency(voc):
    """Synthesis size (T = 1300 frames, 8 utterances: the DFT convs take the wide-workgroup tiles here, unlike the small cases):
    ISTFT(STFT(y)) = y, one item against the oracle, and a Griffin-Lim run from the TRUE phases stays at the fixed point."""
    rng = np.random.RandomState(23)
    B, T = 8, 1300
    n = 256 * (T - 1)
    y = (rng.randn(B, n) * 0.1).astype(np.float32)
    S = voc.stft(_dev(y))
    assert tuple(S.shape) == (B, 1026, T)
    back = voc.istft(S).cpu().numpy()
    assert np.abs(back - y).max() <= 2e-5 * np.abs(y).max()
    ref = vo.stft(y[3])
    assert np.abs(S[3].cpu().numpy() - np.concatenate([ref.real, ref.imag], 0)).max() <= 2e-5 * np.abs(ref).max()
    mag = voc.magnitude(S)
    ang = S / torch.cat([mag, mag], 1).clamp_min(1e-12)               # the consistent phases: Griffin-Lim must not move off them
    w = voc.griffinlim(mag, ang.contiguous(), 3).cpu().numpy()
    assert np.abs(w - y).max() <= 1e-3 * np.abs(y).max()


def test_small_transform_sizes(voc):
    from spoofsv_amd.vocoder import Vocoder
    v = Vocoder(128, 32)
    rng = np.random.RandomState(5)
    y = _wave(rng, 3, 32 * 21 + 7)                # length not a multiple of hop: T = 1 + n // hop
    ref = np.stack([vo.stft(y[b], 128, 32) for b in range(3)])
    S = v.stft(_dev(y)).cpu().numpy()
    assert S.shape == (3, 130, ref.shape[2])
    assert np.abs(S - np.concatenate([ref.real, ref.imag], 1)).max() <= 2e-5 * np.abs(ref).max()


def test_griffinlim_matches_oracle(voc):
    rng = np.random.RandomState(7)
    B, T = 2, 25
    mag = np.abs(np.stack([vo.stft(y) for y in _wave(rng, B, 256 * (T - 1))]))
    a0 = vo.random_angles((B, 513, T), rng)
    for it, tol in ((1, 2e-5), (4, 1e-4)):
        w = np.stack([vo.griffinlim(mag[b], a0[b], it) for b in range(B)])
        g = voc.griffinlim(_dev(mag), _pack(a0), it).cpu().numpy()
        assert g.shape == w.shape and np.abs(g - w).max() <= tol * np.abs(w).max()
    tr_o, tr_g = [], []
    w = vo.griffinlim(mag[0], a0[0], 64, trace=tr_o)
    g = voc.griffinlim(_dev(mag[:1]), _pack(a0[:1]), 64, trace=tr_g).cpu().numpy()[0]
    assert len(tr_g) == 64 and abs(tr_g[3] - tr_o[3]) <= 1e-3 * tr_o[3]
    assert tr_g[-1] <= 1.1 * tr_o[-1] and tr_g[-1] < 0.25 * tr_g[0]          # converges as far as the oracle does
    assert np.abs(g - w).max() <= 2e-2 * np.abs(w).max()                     # measured 8e-4; rounding differences grow with iterations


def test_griffinlim_graph_replay_equals_eager(voc):
    rng = np.random.RandomState(9)
    B, T = 1, 30
    S = _dev(rng.rand(B, 513, T))
    for seed in (1, 2):                         # second call replays the cached graph with new inputs
        a0 = _pack(vo.random_angles((B, 513, T), np.random.RandomState(seed)))
        assert torch.equal(voc.griffinlim_graph(S, a0, 6), voc.griffinlim(S, a0, 6))
        S = S * 0.5 + 0.1


def test_griffinlim_draws_phases_like_librosa_when_none_given(voc):
    S = torch.rand(1, 513, 12, device="cuda")
    torch.manual_seed(3)
    a = voc.griffinlim(S, None, 2)
    torch.manual_seed(3)
    b = voc.griffinlim(S, None, 2)
    assert torch.equal(a, b) and torch.isfinite(a).all() and tuple(a.shape) == (1, 256 * 11)
    ang = voc.random_angles(2, 9)
    assert torch.allclose(ang[:, :513] ** 2 + ang[:, 513:] ** 2, torch.ones(2, 513, 9, device="cuda"), atol=1e-5)


def test_spectrogram2wav_matches_reference_call_sequence(voc):
    rng = np.random.RandomState(11)
    B, T = 2, 20
    lin = rng.rand(B, 513, T).astype(np.float32)
    a0 = vo.random_angles((B, 513, T), rng)

    def ref(l, a):          # oracle.spectrogram2wav with 8 iterations (waveform parity holds at few iterations)
        spec = (l.astype(np.float64) / l.max()) ** (CFG["NORM_POWER"]["RECONSTRUCTION"] / CFG["NORM_POWER"]["ANALYSIS"])
        yy = vo.deemphasis(vo.griffinlim(spec, a, 8), CFG["PREEMPH"])
        return yy / yy.max() * 0.75
    want = np.stack([ref(lin[b], a0[b]) for b in range(B)])
    got = voc.spectrogram2wav(_dev(lin), CFG, _pack(a0), n_iter=8).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4
    assert np.allclose(got.max(1), 0.75, atol=1e-6)


def test_log_feature_spectrograms_both_directions(voc):
    """LOG_FEATURE = true (config.json:26-28): synthesize.py:133-135,147 and data/dataset.py:101-105."""
    cfg = dict(CFG, LOG_FEATURE=True)
    rng = np.random.RandomState(19)
    B, T = 2, 20
    lin = (0.5 + 0.5 * rng.rand(B, 513, T)).astype(np.float32)
    a0 = vo.random_angles((B, 513, T), rng)
    want = np.stack([vo.spectrogram2wav(lin[b], a0[b], cfg, n_iter=4) for b in range(B)])
    got = voc.spectrogram2wav(_dev(lin), cfg, _pack(a0), n_iter=4).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    y = _wave(rng, 1, 256 * 30 + 100)[0]
    mel, ln = voc.wav2spectrogram(_dev(y), cfg["SAMPLING_RATE"], cfg)
    mr, lr = vo.wav2spectrogram(y, cfg["SAMPLING_RATE"], cfg)
    # the log of magnitudes near the 1e-5 floor magnifies the transform's 1e-6 error: absolute tolerance on the [0, 1] scale
    assert np.abs(mel.cpu().numpy() - mr).max() <= 2e-4 and np.abs(ln.cpu().numpy() - lr).max() <= 2e-3


def test_deemphasis_long_rows_match_scipy(voc):
    from spoofsv_amd import _lib, ops
    from spoofsv_amd.vocoder import _p
    rng = np.random.RandomState(13)
    for n in (1, 255, 256, 1000, 332544):                      # 332544 = 256 * 1299, a full synthesis utterance
        x = rng.randn(2, n).astype(np.float32)
        xd = _dev(x)
        out = torch.empty_like(xd)
        _lib.call("ssv_deemphasis", _p(xd), _p(out), 0.97, 2, n, ops._stream())
        want = np.stack([vo.deemphasis(x[b], 0.97) for b in range(2)])
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


def test_wav2spectrogram_matches_dataset_front_end(voc):
    rng = np.random.RandomState(17)
    y = _wave(rng, 1, 256 * 30 + 100)[0]
    mel, lin = voc.wav2spectrogram(_dev(y), CFG["SAMPLING_RATE"], CFG)
    mr, lr = vo.wav2spectrogram(y, CFG["SAMPLING_RATE"], CFG)
    assert tuple(mel.shape) == mr.shape == (80, 7) and tuple(lin.shape) == lr.shape == (513, 28)
    # fp32 pow(x, 0.6) near zero magnifies relative error; normalised spectrograms live in [0, 1] -> absolute tolerance
    assert np.abs(mel.cpu().numpy() - mr).max() <= 2e-5 and np.abs(lin.cpu().numpy() - lr).max() <= 2e-5


def test_bad_arguments_fail_loudly(voc):
    with pytest.raises(RuntimeError):
        voc.griffinlim(torch.rand(1, 513, 8), None, 1)                         # CPU tensor
    with pytest.raises(RuntimeError):
        voc.griffinlim(torch.rand(1, 512, 8, device="cuda"), None, 1)          # wrong bin count
    with pytest.raises(RuntimeError, match="ola"):
        voc.griffinlim(torch.rand(1, 513, 2, device="cuda"), None, 1)          # hop*(T-1) <= N/2: no reflect padding possible
