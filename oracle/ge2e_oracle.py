"""CPU oracle for the GE2E speaker-embedder forward and loss.  TEST INFRASTRUCTURE ONLY.

Restates ``GE2E/speech_embedder_net.py`` and ``GE2E/utils.py:16-55`` of the reference
(cited as speech_embedder_net.py:line / utils.py:line).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

The LSTM is written out gate by gate (torch.nn.LSTM is the third-party arithmetic the
reference calls at speech_embedder_net.py:19,28; gate order i, f, g, o and the update
equations are torch's documented ones); it is pinned against the reference's own
``nn.LSTM`` output through ``tests/golden/ge2e_embedder.npz``.  The loss is a vectorised
restatement of the reference's triple Python loop, pinned by ``tests/golden/ge2e_loss.npz``
and by the known-answer case the reference carries at utils.py:89-96 (loss = 5.2501).
"""
import torch
import torch.nn.functional as F


def lstm_stack(x, sd, num_layers, prefix="LSTM_stack"):
    """nn.LSTM(batch_first=True), speech_embedder_net.py:19,28.  x: (B, T, F) -> (B, T, H).

    Per layer l and step t:  gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh, split as
    (i, f, g, o); c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g); h_t = sigmoid(o) tanh(c_t).
    """
    B, T, _ = x.shape
    inp = x.float()
    for l in range(num_layers):
        w_ih = sd["%s.weight_ih_l%d" % (prefix, l)]
        w_hh = sd["%s.weight_hh_l%d" % (prefix, l)]
        b = sd["%s.bias_ih_l%d" % (prefix, l)] + sd["%s.bias_hh_l%d" % (prefix, l)]
        H = w_hh.shape[1]
        h = torch.zeros(B, H)
        c = torch.zeros(B, H)
        xp = F.linear(inp, w_ih)  # (B, T, 4H): input projection for all steps at once
        outs = []
        for t in range(T):
            gates = xp[:, t] + F.linear(h, w_hh) + b
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, dim=1)
    return inp


def speech_embedder(x, sd, num_layers=3):
    """SpeechEmbedder.forward, speech_embedder_net.py:27-33: last frame -> Linear -> x/||x||."""
    h = lstm_stack(x, sd, num_layers)[:, -1]
    e = F.linear(h, sd["projection.weight"], sd["projection.bias"])
    return e / torch.norm(e, dim=1).unsqueeze(1)


def ge2e_cossim(emb, centroids=None):
    """get_centroids + get_cossim, utils.py:16-46.  emb: (N, M, D) -> (N, M, K).

    cos[j,i,k] = cosine(e_ji, c_k) + 1e-6 with c_k the speaker mean (utils.py:16-25) -- or the given ``centroids`` (K, D),
    as the verification test passes enrollment centroids (train_speech_embedder.py:156-159) -- except k == j, where
    the centroid is the mean of speaker j's OTHER embeddings in ``emb`` (utils.py:27-34, :42-43), also in that case.
    F.cosine_similarity clamps the *product* of norms at eps=1e-8 in torch>=1.12 (each
    norm separately in older versions); the embeddings are unit vectors so it is inert.
    """
    N, M, _ = emb.shape
    cent = emb.mean(dim=1) if centroids is None else centroids   # (K, D)
    K = cent.shape[0]
    loo = (emb.sum(dim=1, keepdim=True) - emb) / (M - 1)     # (N, M, D)
    cos = F.cosine_similarity(emb.unsqueeze(2), cent.view(1, 1, K, -1), dim=3)
    own = F.cosine_similarity(emb, loo, dim=2)               # (N, M)
    idx = torch.arange(min(N, K))
    cos = cos.clone()
    cos[idx, :, idx] = own[idx]
    return cos + 1e-6


def ge2e_loss(emb, w, b):
    """GE2ELoss.forward + calc_loss, speech_embedder_net.py:43-49, utils.py:48-55.

    S = w*cos + b;  L = sum_ji -(S_jij - log(sum_k exp(S_jik) + 1e-6)).
    (The torch.clamp(self.w, 1e-6) at :44 discards its result: a no-op.)
    Returns (loss, per_embedding_loss (N, M)).
    """
    S = w * ge2e_cossim(emb) + b
    N = S.shape[0]
    idx = torch.arange(N)
    pos = S[idx, :, idx]                                      # (N, M)
    per = -(pos - torch.log(torch.exp(S).sum(dim=2) + 1e-6))
    return per.sum(), per


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (train_speech_embedder.py:84-85): scale every gradient by
    min(1, max_norm / (||g||_2 + 1e-6)) with the norm taken over all of them.  Returns (scaled grads, norm)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def ge2e_train_step(x, sd, w, b, N, M, num_layers=3, lr=0.01, clip_net=3.0, clip_loss=1.0):
    """One iteration of GE2E/train_speech_embedder.py:70-86 (the batch permutation at :67-73 is undone at :78 and does not
    enter the arithmetic): loss, gradients by autograd over this restatement, clip_grad_norm_(3.0) on the embedder and
    (1.0) on (w, b), plain SGD.  Returns (loss, grads dict, (dw, db), new sd, (new w, new b))."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    wv = torch.tensor(float(w), requires_grad=True)
    bv = torch.tensor(float(b), requires_grad=True)
    emb = speech_embedder(x, p, num_layers)
    loss, _ = ge2e_loss(emb.reshape(N, M, -1), wv, bv)
    loss.backward()
    keys = list(p.keys())
    grads = {k: p[k].grad.clone() for k in keys}
    gnet, _ = clip_grad_norm([grads[k] for k in keys], clip_net)
    gloss, _ = clip_grad_norm([wv.grad, bv.grad], clip_loss)
    new_sd = {k: (p[k].detach() - lr * g) for k, g in zip(keys, gnet)}
    return loss.detach(), grads, (wv.grad.clone(), bv.grad.clone()), new_sd, (wv.detach() - lr * gloss[0], bv.detach() - lr * gloss[1])


def eer_sweep(sim_matrix, size_1, es1, spoof=True):
    """The threshold sweep of train_speech_embedder.py:168-191 (spoof=True: ``test``) and :262-282 (spoof=False:
    ``test_nospoof``), written as the reference writes it: thresholds 0.50, 0.51, ..., 0.99; the first threshold with
    the smallest |FAR - FRR| wins.  sim_matrix: (N, V, N) verification x enrollment-centroid similarities; size_1 = M
    utterances per speaker in the batch, es1 = 2 * enroll_num of them used for enrollment.
    Returns dict(EER, thres, FAR, FRR[, gt_FRR, spoof_rate])."""
    N = sim_matrix.shape[0]
    diff, out = 1, dict(EER=0, thres=0, FAR=0, FRR=0)
    for thres in [0.01 * i + 0.5 for i in range(50)]:
        th = sim_matrix > thres
        if spoof:
            FAR = sum(th[i].float().sum() - th[i, :, i].float().sum() for i in range(N)) / (N - 1.0) / float(size_1 - es1) / N
            FRR = sum(size_1 - es1 - th[i, :, i].float().sum() for i in range(N)) / float(size_1 - es1) / N
            gtfrr = sum(size_1 // 2 - es1 // 2 - th[i, :(size_1 - es1) // 2, i].float().sum() for i in range(N)) / float(size_1 / 2 - es1 / 2) / N
            spoof_rate = sum(th[i, -(size_1 - es1) // 2:, i].float().sum() for i in range(N)) / float(size_1 / 2 - es1 / 2) / N
        else:
            FAR = sum(th[i].float().sum() - th[i, :, i].float().sum() for i in range(N)) / (N - 1.0) / float(size_1 / 2 - es1 / 2) / N
            FRR = sum(size_1 // 2 - es1 // 2 - th[i, :, i].float().sum() for i in range(N)) / float(size_1 / 2 - es1 / 2) / N
        if diff > abs(FAR - FRR):
            diff = abs(FAR - FRR)
            out = dict(EER=float((FAR + FRR) / 2), thres=thres, FAR=float(FAR), FRR=float(FRR))
            if spoof:
                out.update(gt_FRR=float(gtfrr), spoof_rate=float(spoof_rate))
    return out
