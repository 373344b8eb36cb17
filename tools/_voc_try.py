import sys, time, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
from spoofsv_amd.vocoder import Vocoder
from oracle import vocoder_oracle as vo
cfg = json.load(open("/root/repo/tests/golden/../../gpurun_out/_cfg.json")) if False else {"STFT": {"FFT_LENGTH": 1024, "HOP_LENGTH": 256}, "PREEMPH": 0.97, "NORM_POWER": {"ANALYSIS": 0.6, "RECONSTRUCTION": 1.3}, "COARSE_MELSPEC": {"REDUCTION": 4, "FREQ_BINS": 80}}
v = Vocoder(1024, 256)
rng = np.random.RandomState(0)
B, T = 2, 25
n = 256 * (T - 1)
y = (rng.randn(B, n) * np.hanning(n)[None] * 0.3).astype(np.float32)
yd = torch.from_numpy(y).cuda()
S = v.stft(yd).cpu().numpy()
ref = np.stack([vo.stft(y[b]) for b in range(B)])
print("stft", np.abs(S[:, :513] - ref.real).max() / np.abs(ref).max(), np.abs(S[:, 513:] - ref.imag).max() / np.abs(ref).max())
yr = v.istft(torch.from_numpy(S).cuda()).cpu().numpy()
print("roundtrip", np.abs(yr - y).max() / np.abs(y).max())
refi = np.stack([vo.istft(ref[b]) for b in range(B)])
print("istft", np.abs(yr - refi).max() / np.abs(refi).max())
mag = np.abs(ref).astype(np.float32)
a0 = vo.random_angles((B, 513, T), rng)
a0d = torch.from_numpy(np.concatenate([a0.real, a0.imag], 1).astype(np.float32)).cuda()
for it in (1, 4, 64):
    tr, trg = [], []
    w = np.stack([vo.griffinlim(mag[b].astype(np.float64), a0[b], it, trace=tr if b == 0 else None) for b in range(B)])
    g = v.griffinlim(torch.from_numpy(mag).cuda(), a0d, it, trace=trg).cpu().numpy()
    print("gl", it, np.abs(g - w).max() / np.abs(w).max(), tr[-1], trg[-1])
lin = rng.rand(B, 513, T).astype(np.float32)
wv = v.spectrogram2wav(torch.from_numpy(lin).cuda(), cfg, a0d, n_iter=8).cpu().numpy()
def s2w(l, a):
    spec = (l.astype(np.float64) / l.max()) ** (1.3 / 0.6)
    yy = vo.deemphasis(vo.griffinlim(spec, a, 8), 0.97)
    return yy / yy.max() * 0.75
wr = np.stack([s2w(lin[b], a0[b]) for b in range(B)])
print("s2w", np.abs(wv - wr).max())
mel, ln = v.wav2spectrogram(yd[0], 22050, cfg)
mr, lr = vo.wav2spectrogram(y[0], 22050, cfg)
print("w2s", mel.shape, ln.shape, mr.shape, lr.shape, np.abs(mel.cpu().numpy() - mr).max(), np.abs(ln.cpu().numpy() - lr).max())
# timing at synthesis size
B, T = 16, 1300
S = torch.rand(B, 513, T, device="cuda")
a = v.random_angles(B, T)
for _ in range(2):
    torch.cuda.synchronize(); t = time.time(); w = v.griffinlim(S, a, 64); torch.cuda.synchronize(); print("gl B16 T1300 64it: %.1f ms" % ((time.time() - t) * 1e3))
S1 = S[:1].contiguous(); a1 = a[:1].contiguous()
for _ in range(2):
    torch.cuda.synchronize(); t = time.time(); w = v.griffinlim(S1, a1, 64); torch.cuda.synchronize(); print("gl B1 T1300 64it: %.1f ms" % ((time.time() - t) * 1e3))
t = time.time(); vo.griffinlim(S1[0].cpu().numpy().astype(np.float64), vo.random_angles((513, T), rng), 4); print("oracle 4 it B1: %.2f s" % (time.time() - t))
for _ in range(3):
    torch.cuda.synchronize(); t = time.time(); w = v.griffinlim_graph(S1, a1, 64); torch.cuda.synchronize(); print("gl graph B1 T1300 64it: %.1f ms" % ((time.time() - t) * 1e3))
for _ in range(3):
    torch.cuda.synchronize(); t = time.time(); w = v.griffinlim_graph(S, a, 64); torch.cuda.synchronize(); print("gl graph B16 T1300 64it: %.1f ms" % ((time.time() - t) * 1e3))
