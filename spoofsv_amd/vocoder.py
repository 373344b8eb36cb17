"""GPU vocoder and spectrogram front end.

Replaces the CPU tail of the reference's synthesis scripts -- ``synthesize.py:138-147`` and
``generate_test_utterances.py:128-139``: per-utterance max-normalise, power RECONSTRUCTION/ANALYSIS,
``librosa.core.griffinlim(S, n_iter=64, hop_length, win_length)``, ``scipy.signal.lfilter([1], [1, -PREEMPH], y)``, peak
scale 0.75 -- and the STFT / mel front end of ``data/dataset.py:96-118``, for batches of equal-length utterances (the
synthesis scripts always produce MAX_FRAME_NUM frames, so a whole speaker's sentences go through together).

Design: a 1024-point real DFT over T frames is a (2F x N) by (N x T) matrix product, i.e. a 1x1 convolution of the frame
matrix (B, N, T) with a windowed Fourier basis -- it runs on the same split-bf16 MFMA conv kernel as the models
(``ssv_conv1d_fwd``; bases resident as pre-split planes).  One Griffin-Lim iteration is four launches and never
materialises the waveform: inverse-basis conv -> ``ssv_ola_frames`` (overlap-add, envelope, trim, reflect-pad, re-frame)
-> forward-basis conv -> ``ssv_gl_project`` (momentum phase update fused with ``S * angles``).  At 2*2F*N = 2.1 MFLOP per
frame and transform the DFT-as-GEMM costs ~350 GFLOP per 1300-frame utterance for 64 iterations, ~1.5 ms of MFMA time,
against seconds of ``numpy.fft`` on the host; an FFT kernel would do fewer flops but could not be batched across the
frame axis on the matrix cores.

The basis entries carry 16 mantissa bits (bf16 hi + lo), so one transform is accurate to ~1e-5 of the signal norm
(librosa's complex64 FFT: ~1e-7); Griffin-Lim's own spectral inconsistency is 1e-1..1e-2, four orders above that.
There is no CPU fallback: tensors must be on a ROCm device.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops, resident

_F32 = torch.float32


def _bases(n_fft):
    """Windowed Fourier bases as conv weights: forward (2F, N, 1) and inverse (N, 2F, 1), float32 from float64."""
    N, F = n_fft, n_fft // 2 + 1
    n = np.arange(N)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N)                      # periodic Hann (librosa default window)
    ang = 2.0 * np.pi * ((np.arange(F)[:, None] * n[None, :]) % N) / N   # (F, N), argument reduced exactly
    fwd = np.concatenate([np.cos(ang) * w[None, :], -np.sin(ang) * w[None, :]], 0)           # rfft of the windowed frame
    ck = np.full(F, 2.0)
    ck[0] = ck[-1] = 1.0                                             # irfft: DC and Nyquist count once
    inv = np.concatenate([np.cos(ang) * ck[:, None], -np.sin(ang) * ck[:, None]], 0).T * (w[:, None] / N)   # (N, 2F)
    return (np.ascontiguousarray(fwd, dtype=np.float32)[:, :, None], np.ascontiguousarray(inv, dtype=np.float32)[:, :, None])


def _inv_envelope(n_fft, hop, T):
    """1 / librosa.filters.window_sumsquare (norm=None) over N + hop*(T-1) samples; 1 where the envelope is ~0
    (librosa's istft divides only where it exceeds ``tiny``)."""
    n = n_fft + hop * (T - 1)
    wsq = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)) ** 2
    env = np.zeros(n)
    for t in range(T):
        env[t * hop:t * hop + n_fft] += wsq
    out = np.ones(n)
    nz = env > np.finfo(np.float32).tiny
    out[nz] = 1.0 / env[nz]
    return out.astype(np.float32)


def _slaney_mel(sr, n_fft, n_mels):
    """librosa 0.7.0 ``filters.mel(sr, n_fft, n_mels)`` defaults (Slaney scale, area-normalised triangles)."""
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    top = float(sr) / 2
    mel_top = top / f_sp if top < min_log_hz else min_log_mel + np.log(top / min_log_hz) / logstep
    mels = np.linspace(0.0, mel_top, n_mels + 2)
    hz = np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    freqs = np.linspace(0.0, top, 1 + n_fft // 2)
    ramps = hz[:, None] - freqs[None, :]
    d = np.diff(hz)
    fb = np.maximum(0.0, np.minimum(-ramps[:-2] / d[:-1, None], ramps[2:] / d[1:, None]))
    fb *= (2.0 / (hz[2:] - hz[:-2]))[:, None]
    return fb.astype(np.float32)


def trim_silence(y, top_db=60.0, frame_length=2048, hop_length=512):
    """Host logic: ``librosa.effects.trim(y, top_db)`` of librosa 0.7.0 for a mono numpy signal -- frame RMS (centred,
    reflect-padded frames), dB relative to the loudest frame, keep from the first to one past the last frame above -top_db.
    Returns (y[start:end], (start, end)).  Runs once per written utterance on ~10^5 samples; not a GPU candidate."""
    y = np.asarray(y)
    pad = frame_length // 2
    yp = np.pad(y.astype(np.float64), pad, mode="reflect") if len(y) > pad else np.pad(y.astype(np.float64), pad, mode="constant")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    if n_frames <= 0:
        return y[0:0], (0, 0)
    csum = np.concatenate([[0.0], np.cumsum(yp * yp)])
    idx = np.arange(n_frames) * hop_length
    mse = (csum[idx + frame_length] - csum[idx]) / frame_length
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(max(1e-10, float(mse.max())))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[0:0], (0, 0)
    start, end = int(nz[0]) * hop_length, min(len(y), (int(nz[-1]) + 1) * hop_length)
    return y[start:end], (start, end)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class Vocoder:
    """STFT / ISTFT / Griffin-Lim on one ROCm device for a fixed (n_fft, hop)."""

    def __init__(self, n_fft=1024, hop=256, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("spoofsv_amd.vocoder: needs a ROCm device (no CPU fallback exists), got %s" % dev)
        if n_fft % 2 or hop <= 0 or hop > n_fft:
            raise ValueError("n_fft must be even and 0 < hop <= n_fft")
        self.n_fft, self.hop, self.F, self.device = n_fft, hop, n_fft // 2 + 1, dev
        fwd, inv = _bases(n_fft)
        self.w_fwd = torch.from_numpy(fwd).to(dev)
        self.w_inv = torch.from_numpy(inv).to(dev)
        self._planes = resident.ResidentWeights([self.w_fwd, self.w_inv])   # constants: split once
        self._planes.refresh(ops._stream())
        self._env = {}
        self._mel = {}
        self._graphs = {}

    # ------------------------------------------------------------------ pieces
    def _inv_env(self, T):
        e = self._env.get(T)
        if e is None:
            e = self._env[T] = torch.from_numpy(_inv_envelope(self.n_fft, self.hop, T)).to(self.device)
        return e

    def _dft(self, w, x, out=None):
        B, _, T = x.shape
        if not (w.is_contiguous() and x.is_contiguous() and x.shape[1] == w.shape[1]):
            raise RuntimeError("spoofsv_amd.vocoder: transform input %s does not match the basis %s" % (tuple(x.shape), tuple(w.shape)))
        y = out if out is not None else torch.empty((B, w.shape[0], T), dtype=_F32, device=x.device)
        ops._conv_fwd(x, x.shape[1] * T, w, None, None, y, w.shape[0] * T, 1, 1, 0)
        return y

    def _check_spec(self, S, rows):
        if not (S.is_cuda and S.dtype == _F32 and S.dim() == 3 and S.shape[1] == rows and S.is_contiguous()):
            raise RuntimeError("spoofsv_amd.vocoder: expected a contiguous float32 ROCm tensor (B, %d, T), got %s %s on %s"
                               % (rows, tuple(S.shape), S.dtype, S.device))

    def frames(self, y):
        """(B, n) waveform -> (B, N, T) centred, reflect-padded frames (librosa.stft framing)."""
        if not (y.is_cuda and y.dtype == _F32 and y.dim() == 2 and y.is_contiguous()):
            raise RuntimeError("spoofsv_amd.vocoder: waveform must be a contiguous float32 ROCm tensor (B, n)")
        B, n = y.shape
        T = 1 + n // self.hop
        fr = torch.empty((B, self.n_fft, T), dtype=_F32, device=y.device)
        _lib.call("ssv_frame_signal", _p(y), _p(fr), B, n, self.n_fft, T, self.hop, ops._stream())
        return fr

    def stft(self, y):
        """librosa.stft(y, n_fft, hop_length): (B, n) -> (B, 2F, T), real rows then imaginary rows."""
        return self._dft(self.w_fwd, self.frames(y))

    def magnitude(self, spec):
        self._check_spec(spec, 2 * self.F)
        B, _, T = spec.shape
        mag = torch.empty((B, self.F, T), dtype=_F32, device=spec.device)
        _lib.call("ssv_complex_abs", _p(spec), _p(mag), B, self.F, T, ops._stream())
        return mag

    def istft(self, spec):
        """librosa.istft(S, hop_length): (B, 2F, T) -> (B, hop*(T-1))."""
        self._check_spec(spec, 2 * self.F)
        B, _, T = spec.shape
        fr = self._dft(self.w_inv, spec)
        y = torch.empty((B, self.hop * (T - 1)), dtype=_F32, device=spec.device)
        _lib.call("ssv_ola_signal", _p(fr), _p(self._inv_env(T)), _p(y), B, self.n_fft, T, self.hop, ops._stream())
        return y

    # ------------------------------------------------------------------ Griffin-Lim
    def random_angles(self, B, T, generator=None):
        """librosa's ``init='random'``: exp(2j*pi*U[0,1)) as (B, 2F, T) (cos rows, sin rows)."""
        ph = torch.rand((B, self.F, T), dtype=_F32, device=self.device, generator=generator) * (2.0 * np.pi)
        return torch.cat([torch.cos(ph), torch.sin(ph)], 1)

    def griffinlim(self, S, angles0=None, n_iter=64, momentum=0.99, trace=None):
        """librosa.core.griffinlim(S, n_iter, hop_length, win_length=n_fft) for a batch: S (B, F, T) magnitudes ->
        (B, hop*(T-1)) waveforms.  ``angles0`` (B, 2F, T): initial phases (cos rows, sin rows); None draws them like
        librosa does.  ``trace``: optional list that receives the rebuilt spectra's inconsistency per iteration (syncs)."""
        self._check_spec(S, self.F)
        B, F, T = S.shape
        if angles0 is None:
            angles0 = self.random_angles(B, T)
        self._check_spec(angles0, 2 * F)
        st, N, hop = ops._stream(), self.n_fft, self.hop
        env = self._inv_env(T)
        alpha = momentum / (1.0 + momentum)
        proj = torch.empty((B, 2 * F, T), dtype=_F32, device=S.device)
        reb = [torch.empty_like(proj), torch.empty_like(proj)]
        fa = torch.empty((B, N, T), dtype=_F32, device=S.device)
        fb = torch.empty_like(fa)
        _lib.call("ssv_gl_project", _p(S), _p(angles0), None, 0.0, _p(proj), B, F, T, st)
        for it in range(n_iter):
            cur = reb[it & 1]
            self._dft(self.w_inv, proj, fa)
            _lib.call("ssv_ola_frames", _p(fa), _p(env), _p(fb), B, N, T, hop, st)
            self._dft(self.w_fwd, fb, cur)
            if trace is not None:
                m = self.magnitude(cur)
                trace.append(float((m - S).norm() / S.norm()))
            _lib.call("ssv_gl_project", _p(S), _p(cur), _p(reb[1 - (it & 1)]) if it else None, alpha, _p(proj), B, F, T, st)
        self._dft(self.w_inv, proj, fa)
        y = torch.empty((B, hop * (T - 1)), dtype=_F32, device=S.device)
        _lib.call("ssv_ola_signal", _p(fa), _p(env), _p(y), B, N, T, hop, st)
        return y

    def griffinlim_graph(self, S, angles0=None, n_iter=64, momentum=0.99):
        """``griffinlim`` replayed from a captured hipGraph (one per (B, T, n_iter, momentum)): the 4*n_iter + 3 launches
        of a small batch are launch-bound when issued one by one (B = 1, T = 1300: 9.3 ms eager)."""
        self._check_spec(S, self.F)
        B, F, T = S.shape
        key = (B, T, int(n_iter), float(momentum))
        g = self._graphs.get(key)
        if g is None:
            sS = torch.empty_like(S)
            sA = torch.empty((B, 2 * F, T), dtype=_F32, device=S.device)
            sS.copy_(S)
            sA.copy_(self.random_angles(B, T))
            self._inv_env(T)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.griffinlim(sS, sA, 1, momentum)                # warm the allocator outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.griffinlim(sS, sA, n_iter, momentum)
            g = self._graphs[key] = (graph, sS, sA, out)
        graph, sS, sA, out = g
        sS.copy_(S)
        sA.copy_(angles0 if angles0 is not None else self.random_angles(B, T))
        graph.replay()
        return out.clone()

    # ------------------------------------------------------------------ reference call sequences
    def _norm_pow(self, x2d, p, s):
        B, n = x2d.shape
        mx = torch.empty((B,), dtype=_F32, device=x2d.device)
        out = torch.empty_like(x2d)
        _lib.call("ssv_rowmax", _p(x2d), _p(mx), B, n, ops._stream())
        _lib.call("ssv_scale_pow", _p(x2d), _p(mx), _p(out), float(p), float(s), B, n, ops._stream())
        return out

    def spectrogram2wav(self, lin, cfg, angles0=None, n_iter=64, graph=False, peak=0.75):
        """synthesize.py:129-147 for a batch (both settings of LOG_FEATURE): SSRN output (B, F, T) -> (B, hop*(T-1)) waveforms,
        each max-normalised to ``peak`` = 0.75 as the reference writes them (``peak=None``: the de-emphasised signal as it
        is, for callers that trim before normalising, generate_test_utterances.py:135-139)."""
        self._check_spec(lin, self.F)
        B, F, T = lin.shape
        p = cfg["NORM_POWER"]["RECONSTRUCTION"] / cfg["NORM_POWER"]["ANALYSIS"]
        log = bool(cfg.get("LOG_FEATURE", False))
        if log:      # synthesize.py:133-135: dB = x*MAX_DB - MAX_DB + REF_DB, amplitude 10^(dB/20); then the power of :142
            k = np.log(10.0) * 0.05 * p
            spec = torch.empty_like(lin)
            _lib.call("ssv_exp_affine", _p(lin), _p(spec), float(k * cfg["MAX_DB"]), float(k * (cfg["REF_DB"] - cfg["MAX_DB"])),
                      B * F * T, ops._stream())
        else:
            spec = self._norm_pow(lin.view(B, F * T), p, 1.0).view(B, F, T)
        y = (self.griffinlim_graph if graph else self.griffinlim)(spec, angles0, n_iter=n_iter)
        out = torch.empty_like(y)
        _lib.call("ssv_deemphasis", _p(y), _p(out), float(cfg["PREEMPH"]), B, y.shape[1], ops._stream())
        if log or peak is None:      # synthesize.py:147 writes the LOG_FEATURE signal as it is
            return out
        return self._norm_pow(out, 1.0, peak)

    def mel_basis(self, sr, n_mels):
        key = (int(sr), int(n_mels))
        w = self._mel.get(key)
        if w is None:
            w = self._mel[key] = torch.from_numpy(_slaney_mel(sr, self.n_fft, n_mels)[:, :, None].copy()).to(self.device)
        return w

    def wav2spectrogram(self, speech, sr, cfg):
        """data/dataset.py:96-118 (both settings of LOG_FEATURE) for one loaded, trimmed utterance (n,) on the device:
        returns (reduced mel (n_mels, T//r), linear (F, r*(T//r))) normalised spectrograms."""
        if not (speech.is_cuda and speech.dim() == 1):
            raise RuntimeError("spoofsv_amd.vocoder: speech must be a 1-D ROCm tensor")
        x = speech.to(_F32).contiguous().view(1, -1)
        n = x.shape[1]
        pre = torch.empty_like(x)
        _lib.call("ssv_preemphasis", _p(x), _p(pre), float(cfg["PREEMPH"]), 1, n, ops._stream())
        lin = self.magnitude(self.stft(pre))                               # (1, F, T)
        T = lin.shape[2]
        mel = self._dft(self.mel_basis(sr, cfg["COARSE_MELSPEC"]["FREQ_BINS"]), lin)
        if cfg.get("LOG_FEATURE", False):      # data/dataset.py:101-105
            lin_n, mel_n = torch.empty_like(lin), torch.empty_like(mel)
            for src, dst in ((lin, lin_n), (mel, mel_n)):
                _lib.call("ssv_log_norm", _p(src), _p(dst), float(cfg["REF_DB"]), float(cfg["MAX_DB"]), src.numel(), ops._stream())
            lin_n, mel_n = lin_n.view(self.F, T), mel_n.view(-1, T)
        else:
            p = cfg["NORM_POWER"]["ANALYSIS"]
            lin_n = self._norm_pow(lin.view(1, -1), p, 1.0).view(self.F, T)
            mel_n = self._norm_pow(mel.view(1, -1), p, 1.0).view(-1, T)
        r = cfg["COARSE_MELSPEC"]["REDUCTION"]
        rt = T // r
        return mel_n[:, 0:r * rt:r].contiguous(), lin_n[:, :r * rt].contiguous()
