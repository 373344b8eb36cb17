"""CPU tests of the vocoder oracle (oracle/vocoder_oracle.py) and of the host-side constants of spoofsv_amd.vocoder.

librosa 0.7.0 (requirements.txt:1 of the reference) is absent from the image, so the oracle is anchored on what can be
checked independently: numpy.fft, exact reconstruction, the closed-form Hann envelope and the filterbank value librosa's
own docstring prints.
"""
import numpy as np
import pytest

from oracle import vocoder_oracle as vo


def test_stft_matches_direct_dft_definition():
    rng = np.random.RandomState(1)
    y = rng.randn(64 * 9)
    S = vo.stft(y, n_fft=128, hop=32)
    assert S.shape == (65, 1 + len(y) // 32)
    yp = np.pad(y, 64, mode="reflect")
    n = np.arange(128)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / 128)
    for t in (0, 3, S.shape[1] - 1):
        frame = w * yp[t * 32:t * 32 + 128]
        direct = np.array([np.sum(frame * np.exp(-2j * np.pi * k * n / 128)) for k in range(65)])
        assert np.abs(S[:, t] - direct).max() < 1e-10


def test_istft_inverts_stft_and_envelope_is_closed_form():
    rng = np.random.RandomState(2)
    y = rng.randn(256 * 10)
    assert np.abs(vo.istft(vo.stft(y)) - y).max() < 1e-12
    env = vo.window_sumsquare(11)
    assert np.allclose(env[1024:-1024], 1.5)          # sum of 4 shifted hann^2 at hop = N/4
    assert env[0] == 0.0


def test_mel_filterbank_known_answer_and_shape():
    m = vo.mel_filterbank(22050, 2048, 128)
    assert m.shape == (128, 1025) and m.dtype == np.float32
    assert round(float(m[0, 1]), 3) == 0.016          # value printed in librosa 0.7.0's filters.mel docstring example
    assert m[0, 0] == 0 and m[-1, -1] == 0 and (m >= 0).all()
    df = 22050 / 2048
    area = m.astype(np.float64).sum(1) * df           # Slaney normalisation: unit-area triangles (in Hz)
    assert np.allclose(area, 1.0, atol=0.05) and np.allclose(area[100:], 1.0, atol=2e-3)
    m80 = vo.mel_filterbank(22050, 1024, 80)          # the reference's configuration (config.json:15-24)
    assert m80.shape == (80, 513) and (m80.sum(1) > 0).all()


def test_griffinlim_reduces_inconsistency_and_preemphasis_round_trip():
    rng = np.random.RandomState(3)
    y = rng.randn(32 * 40) * np.hanning(32 * 40)
    S = np.abs(vo.stft(y, 128, 32))
    tr = []
    w = vo.griffinlim(S, vo.random_angles(S.shape, rng), n_iter=32, hop=32, trace=tr)
    assert len(w) == len(y) and tr[-1] < 0.5 * tr[0]
    assert np.abs(vo.deemphasis(vo.preemphasis(y)) - y).max() < 1e-10


def test_host_constants_match_oracle():
    from spoofsv_amd import vocoder as V
    for sr, n_fft, n_mels in ((22050, 1024, 80), (16000, 512, 40)):
        assert np.array_equal(V._slaney_mel(sr, n_fft, n_mels), vo.mel_filterbank(sr, n_fft, n_mels))
    fwd, inv = V._bases(128)
    assert fwd.shape == (130, 128, 1) and inv.shape == (128, 130, 1) and fwd.flags.c_contiguous and inv.flags.c_contiguous
    rng = np.random.RandomState(4)
    x = rng.randn(128)
    w = vo.hann_periodic(128)
    sp = fwd[:, :, 0].astype(np.float64) @ x
    ref = np.fft.rfft(w * x)
    assert np.abs(sp[:65] - ref.real).max() < 1e-5 and np.abs(sp[65:] - ref.imag).max() < 1e-5
    S = rng.randn(65) + 1j * rng.randn(65)
    fr = inv[:, :, 0].astype(np.float64) @ np.concatenate([S.real, S.imag])
    assert np.abs(fr - w * np.fft.irfft(S, 128)).max() < 1e-6
    e = V._inv_envelope(128, 32, 9)
    env = vo.window_sumsquare(9, 128, 32)
    assert e.shape == env.shape and np.allclose(e[32:-32] * env[32:-32], 1.0, atol=1e-6) and e[0] == 1.0


def test_vocoder_refuses_cpu():
    from spoofsv_amd import vocoder as V
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        V.Vocoder(128, 32, device="cpu")


def test_trim_silence_host_logic():
    """librosa.effects.trim restated (generate_test_utterances.py:136): frame-RMS threshold relative to the loudest frame."""
    from spoofsv_amd.vocoder import trim_silence
    sr = 22050
    tone = 0.5 * np.sin(2 * np.pi * 220 * np.arange(20000) / sr)
    y = np.concatenate([np.zeros(5000), tone, 1e-4 * np.random.RandomState(0).randn(8000)])
    t, (a, b) = trim_silence(y, 30)
    assert a % 512 == 0 and 5000 - 1024 - 512 <= a <= 5000 and 25000 <= b <= 25000 + 1024 + 512 and len(t) == b - a
    assert trim_silence(y, 120)[1] == (0, len(y))                  # nothing is 120 dB below the peak frame
    assert trim_silence(tone, 30)[1] == (0, len(tone))
    assert trim_silence(np.zeros(0), 30)[1] == (0, 0)
