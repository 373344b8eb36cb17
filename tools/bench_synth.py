#!/usr/bin/env python3
"""Time the free-running synthesis loop (reference: synthesize.py:103-109; config 2 of SURVEY.md 8d): 1 + 325 melSyn eval
steps, each re-encoding the whole prefix as the reference does, then SSRN.  Reports mel frames per second."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import harness, ops, resident, train
from spoofsv_amd.tts import melSyn, SSRN
dev = "cuda:0"
torch.manual_seed(1234)
m = melSyn(34, True, 200); m.apply(train.init_weights); m = m.to(dev).eval()
s = SSRN(80, 513, 256); s.apply(train.init_weights); s = s.to(dev).eval()
for B, N in ((1, 43), (8, 43), (32, 186)):
    text = torch.randint(2, 33, (B, 1, N), device=dev); text[:, :, -1] = 1
    spk = 0.04 + 0.05 * torch.rand(B, 200, 1, device=dev)
    frames = 326
    with torch.no_grad():
        harness._free_run(m, text, spk, 8, 80)              # warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Y, A = harness._free_run(m, text, spk, frames, 80)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        resident.ensure(s, ops._stream())
        lin = s(Y)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        harness._free_run(m, text, spk, frames, 80, graph=True)   # capture
        torch.cuda.synchronize(); t3 = time.perf_counter()
        Yg, Ag = harness._free_run(m, text, spk, frames, 80, graph=True)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        harness._free_run(m, text, spk, frames, 80, incremental=True)   # capture
        torch.cuda.synchronize(); t5 = time.perf_counter()
        for _ in range(3):
            Yi, Ai = harness._free_run(m, text, spk, frames, 80, incremental=True)
        torch.cuda.synchronize(); t6 = (time.perf_counter() - t5) / 3
    same = bool(torch.equal(Yg, Y) and torch.equal(Ag, A))
    inc = "incremental %.3f s (%.3f ms/frame), max |dY| %.1e, arg-max path equal %s" % (
        t6, t6 / frames * 1e3, float((Yi - Y).abs().max()), bool(torch.equal(Ai.argmax(1), A.argmax(1))))
    print("B=%d N=%d: step-by-step %.3f s (%.2f ms/frame) | graph replay %.3f s (%.3f ms/frame), identical=%s | %s | ssrn %.1f ms -> %.0f / %.0f / %.0f mel frames/s" %
          (B, N, t1 - t0, (t1 - t0) / frames * 1e3, t4 - t3, (t4 - t3) / frames * 1e3, same, inc, (t2 - t1) * 1e3,
           B * frames / (t2 - t0), B * frames / (t4 - t3 + t2 - t1), B * frames / (t6 + t2 - t1)), flush=True)

# One speaker of generate_test_utterances.py: 20 sentences as a batch, free run + SSRN + vocoder (64 Griffin-Lim iterations)
from spoofsv_amd.vocoder import Vocoder
import json
cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config.json")))
voc = Vocoder(1024, 256)
B, N, frames = 20, 80, 326
text = torch.randint(2, 33, (B, 1, N), device=dev); text[:, :, -1] = 1
spk = (0.04 + 0.05 * torch.rand(1, 200, 1, device=dev)).expand(B, -1, -1).contiguous()
with torch.no_grad():
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Y, A = harness._free_run(m, text, spk, frames, 80, incremental=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        resident.ensure(s, ops._stream())
        lin = s(Y).contiguous()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        wav = voc.spectrogram2wav(lin, cfg, peak=None)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print("speaker batch B=20 N=80: incremental free run %.1f ms, SSRN %.1f ms, vocoder (64 it, T=%d) %.1f ms -> %.1f utterances/s end to end" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, lin.shape[2], (t3 - t2) * 1e3, B / (t3 - t0)), flush=True)
