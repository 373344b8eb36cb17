"""CPU oracle for the vocoder end of synthesis and the spectrogram front end.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Restates, in numpy (float64 unless a function says otherwise):

* ``synthesize.py:138-147`` / ``generate_test_utterances.py:128-139`` -- per-utterance max-normalise, power
  RECONSTRUCTION/ANALYSIS, Griffin-Lim with 64 iterations, de-emphasis ``lfilter([1], [1, -PREEMPH])``, peak scale 0.75;
* ``data/dataset.py:96-118`` -- pre-emphasis, |STFT(1024, 256)|, 80-band mel projection, max-normalise, power 0.6,
  time reduction by 4.

PARITY UNPINNED against the third-party code: the STFT / ISTFT / Griffin-Lim / mel-filterbank arithmetic lives in
``librosa`` (pinned by the reference at ``requirements.txt:1`` to **librosa==0.7.0**), which is not installed in this
image and is not vendored under /root/reference.  The functions below restate librosa 0.7.0's published algorithms
(``librosa.core.stft`` / ``istft`` / ``griffinlim``, ``librosa.filters.mel`` / ``window_sumsquare``; Perraudin et al.
2013 "fast Griffin-Lim", momentum 0.99) with that version's defaults -- periodic Hann window, ``center=True``,
``pad_mode='reflect'``, Slaney mel scale with area normalisation.  They are anchored on what can be checked here:
the DFT against ``numpy.fft``, ISTFT(STFT(y)) == y, the window-sum-square envelope in closed form, and the one
filterbank value librosa 0.7.0's docstring prints (``librosa.filters.mel(22050, 2048)[0, 1] == 0.016``).  librosa draws
Griffin-Lim's initial phases from an unseeded RNG (``random_state=None`` at the reference's call sites), so the
reference's waveform is not reproducible either; here the initial phases are an explicit argument.
"""
import numpy as np
from scipy import signal as _signal


def hann_periodic(n):
    """``scipy.signal.get_window('hann', n, fftbins=True)`` -- librosa's default STFT window."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft(y, n_fft=1024, hop=256):
    """librosa 0.7.0 ``core.stft(y, n_fft, hop_length)`` with win_length = n_fft: (1 + n_fft/2, n_frames) complex."""
    y = np.asarray(y, dtype=np.float64)
    w = hann_periodic(n_fft)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    out = np.empty((1 + n_fft // 2, n_frames), dtype=np.complex128)
    for t in range(n_frames):
        out[:, t] = np.fft.rfft(w * yp[t * hop:t * hop + n_fft])
    return out


def window_sumsquare(n_frames, n_fft=1024, hop=256):
    """librosa 0.7.0 ``filters.window_sumsquare`` (norm=None): overlap-added squared window."""
    n = n_fft + hop * (n_frames - 1)
    x = np.zeros(n)
    wsq = hann_periodic(n_fft) ** 2
    for t in range(n_frames):
        s = t * hop
        x[s:min(n, s + n_fft)] += wsq[:max(0, min(n_fft, n - s))]
    return x


def istft(S, hop=256):
    """librosa 0.7.0 ``core.istft(S, hop_length)`` with win_length = n_fft, center=True, length=None."""
    n_fft = 2 * (S.shape[0] - 1)
    n_frames = S.shape[1]
    w = hann_periodic(n_fft)
    y = np.zeros(n_fft + hop * (n_frames - 1))
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += w * np.fft.irfft(S[:, t], n_fft)
    env = window_sumsquare(n_frames, n_fft, hop)
    nz = env > np.finfo(np.float32).tiny
    y[nz] /= env[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def random_angles(shape, rng):
    """librosa 0.7.0 griffinlim ``init='random'``: exp(2j*pi*U[0,1))."""
    return np.exp(2j * np.pi * rng.rand(*shape))


def griffinlim(S, angles0, n_iter=64, hop=256, momentum=0.99, trace=None):
    """librosa 0.7.0 ``core.griffinlim(S, n_iter, hop_length, win_length=n_fft)`` from the given initial phases.

    ``trace``: optional list that receives the spectral inconsistency ||abs(STFT(ISTFT(S*angles))) - S|| / ||S|| per iteration.
    """
    n_fft = 2 * (S.shape[0] - 1)
    angles = np.array(angles0, dtype=np.complex128)
    rebuilt = 0.0
    for _ in range(n_iter):
        tprev = rebuilt
        inverse = istft(S * angles, hop)
        rebuilt = stft(inverse, n_fft, hop)
        if trace is not None:
            trace.append(float(np.linalg.norm(np.abs(rebuilt) - S) / np.linalg.norm(S)))
        angles = rebuilt - (momentum / (1 + momentum)) * tprev
        angles = angles / (np.abs(angles) + 1e-16)
    return istft(S * angles, hop)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels):
    """librosa 0.7.0 ``filters.mel(sr, n_fft, n_mels)`` (fmin=0, fmax=sr/2, htk=False, norm=1): (n_mels, 1+n_fft/2)."""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(float(sr) / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)      # librosa returns float32 weights


def preemphasis(speech, a=0.97):
    """data/dataset.py:96."""
    speech = np.asarray(speech)
    return np.append(speech[0], speech[1:] - a * speech[:-1])


def deemphasis(x, a=0.97):
    """synthesize.py:145 -- ``signal.lfilter([1], [1, -PREEMPH], time_signal)``."""
    return _signal.lfilter([1], [1, -a], x)


def spectrogram2wav(lin, angles0, cfg, n_iter=64):
    """synthesize.py:129-147 for one utterance: lin (F, T) SSRN output -> waveform."""
    lin = np.asarray(lin, dtype=np.float64)
    if cfg.get("LOG_FEATURE", False):                      # :133-135
        lin = np.power(10, 0.05 * (lin * cfg["MAX_DB"] - cfg["MAX_DB"] + cfg["REF_DB"]))
    else:                                                  # :140-141
        lin = lin / np.max(lin)
    spec = lin ** (cfg["NORM_POWER"]["RECONSTRUCTION"] / cfg["NORM_POWER"]["ANALYSIS"])
    y = griffinlim(spec, angles0, n_iter=n_iter, hop=cfg["STFT"]["HOP_LENGTH"])
    y = deemphasis(y, cfg["PREEMPH"])
    return y if cfg.get("LOG_FEATURE", False) else y / np.max(y) * 0.75      # :147


def wav2spectrogram(speech, sr, cfg):
    """data/dataset.py:96-118 for one (already loaded and trimmed) utterance.
    Returns (reduced mel (80, T//4), linear (513, 4*(T//4)))."""
    n_fft, hop = cfg["STFT"]["FFT_LENGTH"], cfg["STFT"]["HOP_LENGTH"]
    r = cfg["COARSE_MELSPEC"]["REDUCTION"]
    speech = preemphasis(speech, cfg["PREEMPH"])
    lin = np.abs(stft(speech, n_fft, hop))
    mel = np.dot(mel_filterbank(sr, n_fft, cfg["COARSE_MELSPEC"]["FREQ_BINS"]).astype(np.float64), lin)
    if cfg.get("LOG_FEATURE", False):                      # :101-105
        mel_db, lin_db = 20 * np.log10(np.maximum(1e-5, mel)), 20 * np.log10(np.maximum(1e-5, lin))
        mel_n = np.clip((mel_db - cfg["REF_DB"] + cfg["MAX_DB"]) / cfg["MAX_DB"], 1e-8, 1)
        lin_n = np.clip((lin_db - cfg["REF_DB"] + cfg["MAX_DB"]) / cfg["MAX_DB"], 1e-8, 1)
    else:                                                  # :107-111
        p = cfg["NORM_POWER"]["ANALYSIS"]
        lin_n = (lin / np.max(lin)) ** p
        mel_n = (mel / np.max(mel)) ** p
    rt = mel.shape[1] // r
    return mel_n[:, [r * k for k in range(rt)]], lin_n[:, :r * rt]
