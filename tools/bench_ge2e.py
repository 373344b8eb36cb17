#!/usr/bin/env python3
"""BASELINE config 5: GE2E d-vector extraction, 88 speakers x 10 utterances x 120 frames x 40 mels: LSTM forward +
projection + GE2E loss on one MI355X, utterances/s, next to the CPU oracle on a bounded sample."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd.ge2e import SpeechEmbedder, GE2ELoss
from oracle import ge2e_oracle as GO

def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    m = SpeechEmbedder()
    x = torch.randn(880, 120, 40)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).eval()
    L = GE2ELoss(dev)
    xg = x.to(dev)
    def step():
        with torch.no_grad():
            e = m(xg)
            return e, L(e.view(88, 10, 256))
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): e, loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # one training iteration (GE2E/train_speech_embedder.py:70-86): forward keeping every frame, loss, backward, clip, SGD
    m.train()
    opt = torch.optim.SGD([{"params": m.parameters()}, {"params": L.parameters()}], lr=0.01)
    def train_step():
        opt.zero_grad(set_to_none=True)
        ls = L(m(xg).reshape(88, 10, 256))
        ls.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0)
        torch.nn.utils.clip_grad_norm_(L.parameters(), 1.0)
        opt.step()
        return ls
    train_step(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(3): tl = train_step()
    torch.cuda.synchronize()
    dtt = (time.perf_counter() - t2) / 3
    m.eval()
    # CPU oracle on 44 utterances (bounded), all cores of the box share
    cores = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception: pass
    torch.set_num_threads(cores)
    xs = x[:44]
    with torch.no_grad():
        t1 = time.perf_counter(); eo = GO.speech_embedder(xs, sd); tc = time.perf_counter() - t1
        lo, _ = GO.ge2e_loss(GO.speech_embedder(x[:40], sd).view(4, 10, 256), torch.tensor(10.0), torch.tensor(-5.0))
    err = float((e[:44].cpu() - eo).abs().max() / eo.abs().max())
    flops = 880 * 120 * 2 * (4 * 768 * (40 + 768) + 2 * 4 * 768 * (768 + 768)) + 880 * 2 * 768 * 256
    print(json.dumps({"metric": "GE2E utterances/s (LSTM fwd + projection + loss)", "value": round(880 / dt, 1), "ms": round(dt * 1e3, 2),
                      "tflops": round(flops / dt / 1e12, 1), "loss": round(float(loss), 4), "rel_err_vs_cpu_oracle": err,
                      "train_iteration": {"ms": round(dtt * 1e3, 2), "utterances_per_s": round(880 / dtt, 1), "tflops": round(3 * flops / dtt / 1e12, 1),
                                          "loss_after": round(float(tl.detach()), 4)},
                      "cpu_baseline": {"value": round(44 / tc, 1), "unit": "utterances/s", "cores": cores, "sample": "44 utterances x 120 frames"}}))
if __name__ == "__main__":
    main()
