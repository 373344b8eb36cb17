"""Host-side mirror of the reference's GE2E speaker-verification scripts around the HIP embedder:
``GE2E/data_load.py`` (preprocessed TI-SV data), ``GE2E/train_speech_embedder.py`` (``train``, ``test``,
``test_nospoof``: EER and spoof rate).  The embedder and the loss run in libssv_hip.so (``spoofsv_amd.ge2e``); data
loading, the enrollment/verification bookkeeping and the threshold sweep are host logic, kept as the reference has them.

Configuration is a plain dict with the fields of ``GE2E/config/config.yaml`` (``default_config()``), instead of the
reference's module-global ``hparam`` object.
"""
import os
import random
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .ge2e import GE2ELoss, SpeechEmbedder, train_iteration


def default_config():
    """GE2E/config/config.yaml as shipped."""
    return {
        "training": False, "device": "cuda", "save_simmat_dir": "./simmat",
        "data": {"train_path": "./train_tisv", "test_path": "./test_tisv", "nmels": 40, "tisv_frame": 120},
        "model": {"hidden": 768, "num_layer": 3, "proj": 256, "model_path": None},
        "train": {"N": 6, "M": 50, "num_workers": 0, "lr": 0.01, "epochs": 950, "log_interval": 5, "log_file": None,
                  "checkpoint_interval": 120, "checkpoint_dir": "./speech_id_checkpoint", "restore": False},
        "test": {"N": 20, "M": 86, "num_workers": 0, "epochs": 10},
    }


class SpeakerDatasetPreprocessed(Dataset):
    """SpeakerDatasetTIMITPreprocessed, GE2E/data_load.py:48-86: one ``.npy`` per speaker holding
    (utterances, n_mels, frames); an item is M utterances of one speaker as (M, frames, n_mels)."""

    def __init__(self, path, utter_num, shuffle=False, utter_start=0):
        self.path, self.utter_num, self.shuffle, self.utter_start = path, utter_num, shuffle, utter_start
        self.file_list = sorted(os.listdir(path))

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        files = self.file_list
        selected = files[idx] if self.shuffle else random.sample(files, 1)[0]          # data_load.py:70-73
        utters = np.load(os.path.join(self.path, selected))
        if self.shuffle:
            utterance = utters[np.random.randint(0, utters.shape[0], self.utter_num)]
        else:
            utterance = utters[self.utter_start:self.utter_start + self.utter_num]
        return torch.tensor(np.transpose(utterance, axes=(0, 2, 1)))


def _embedder(cfg, device):
    m = cfg["model"]
    return SpeechEmbedder(cfg["data"]["nmels"], m["hidden"], m["num_layer"], m["proj"]).to(device)


def train(cfg, model_path=None):
    """train_speech_embedder.py:40-108.  Returns (embedder, list of per-iteration losses)."""
    device = torch.device(cfg["device"])
    tr = cfg["train"]
    loader = DataLoader(SpeakerDatasetPreprocessed(cfg["data"]["train_path"], tr["M"], shuffle=True), batch_size=tr["N"], shuffle=True,
                        num_workers=tr["num_workers"], drop_last=True)
    net = _embedder(cfg, device)
    if tr["restore"]:
        net.load_state_dict(torch.load(model_path, map_location="cpu"))
    ge2e_loss = GE2ELoss(device)
    optimizer = torch.optim.SGD([{"params": net.parameters()}, {"params": ge2e_loss.parameters()}], lr=tr["lr"])
    if tr["checkpoint_dir"]:
        os.makedirs(tr["checkpoint_dir"], exist_ok=True)
    net.train()
    iteration, history = 0, []
    e = batch_id = 0
    for e in range(tr["epochs"]):
        total = 0.0
        for batch_id, mel_db_batch in enumerate(loader):
            loss = train_iteration(net, ge2e_loss, optimizer, mel_db_batch.to(device).float(), tr["N"], tr["M"])
            history.append(float(loss))
            total += history[-1]
            iteration += 1
            if (batch_id + 1) % tr["log_interval"] == 0:
                mesg = "{0}\tEpoch:{1}[{2}/{3}],Iteration:{4}\tLoss:{5:.4f}\tTLoss:{6:.4f}\t\n".format(
                    time.ctime(), e + 1, batch_id + 1, len(loader.dataset) // tr["N"], iteration, history[-1], total / (batch_id + 1))
                print(mesg)
                if tr["log_file"]:
                    with open(tr["log_file"], "a") as f:
                        f.write(mesg)
        if tr["checkpoint_dir"] and (e + 1) % tr["checkpoint_interval"] == 0:
            torch.save({k: v.cpu() for k, v in net.state_dict().items()},
                       os.path.join(tr["checkpoint_dir"], "ckpt_epoch_%d_batch_id_%d.pth" % (e + 1, batch_id + 1)))
    if tr["checkpoint_dir"]:
        torch.save({k: v.cpu() for k, v in net.state_dict().items()},
                   os.path.join(tr["checkpoint_dir"], "final_epoch_%d_batch_id_%d.model" % (e + 1, batch_id + 1)))
    return net, history


def cossim_eval(verification_embeddings, enrollment_centroids):
    """utils.get_cossim as the tests call it (train_speech_embedder.py:156-159): cosine of every verification embedding
    with every enrollment centroid, + 1e-6; the own-speaker column uses the mean of that speaker's OTHER verification
    embeddings (utils.py:42-43).  (N, V, D), (N, D) -> (N, V, N).  Vectorised; pinned against the reference by
    tests/golden/ge2e_train.npz through the oracle."""
    ver, cent = verification_embeddings, enrollment_centroids
    N, V, _ = ver.shape
    cos = torch.nn.functional.cosine_similarity(ver.unsqueeze(2), cent.view(1, 1, cent.shape[0], -1), dim=3)
    loo = (ver.sum(dim=1, keepdim=True) - ver) / (V - 1)
    own = torch.nn.functional.cosine_similarity(ver, loo, dim=2)
    idx = torch.arange(N, device=ver.device)
    cos = cos.clone()
    cos[idx, :, idx] = own
    return cos + 1e-6


def eer_sweep(sim_matrix, size_1, es1, spoof=True):
    """The threshold sweep of train_speech_embedder.py:168-191 (``test``) / :262-282 (``test_nospoof``): thresholds
    0.50 ... 0.99, first minimum of |FAR - FRR|.  Same counts as the reference's Python loops, computed in one shot."""
    N = sim_matrix.shape[0]
    thr = torch.tensor([0.01 * i + 0.5 for i in range(50)], device=sim_matrix.device, dtype=sim_matrix.dtype)
    th = (sim_matrix.unsqueeze(0) > thr.view(-1, 1, 1, 1)).double()                  # (50, N, V, N)
    idx = torch.arange(N, device=sim_matrix.device)
    own = th[:, idx, :, idx].permute(1, 0, 2)                                        # (50, N, V): own-speaker column
    tot, own_s = th.sum(dim=(1, 2, 3)), own.sum(dim=(1, 2))
    den = float(size_1 - es1) if spoof else float(size_1 / 2 - es1 / 2)
    pos = (size_1 - es1) if spoof else (size_1 // 2 - es1 // 2)
    FAR = (tot - own_s) / (N - 1.0) / den / N
    FRR = (N * pos - own_s) / den / N
    d = (FAR - FRR).abs()
    best, cur = 0, 1.0
    for i in range(50):                                                              # strict "<", as `if diff > abs(FAR-FRR)`
        if cur > float(d[i]):
            cur, best = float(d[i]), i
    out = dict(EER=float((FAR[best] + FRR[best]) / 2), thres=0.01 * best + 0.5, FAR=float(FAR[best]), FRR=float(FRR[best]))
    if spoof:
        half = (size_1 - es1) // 2
        hden = float(size_1 / 2 - es1 / 2)
        out["gt_FRR"] = float((N * (size_1 // 2 - es1 // 2) - own[best, :, :half].sum()) / hden / N)
        out["spoof_rate"] = float(own[best, :, -half:].sum() / hden / N)
    return out


def _verification_batches(cfg, net, enroll_num, device):
    te = cfg["test"]
    loader = DataLoader(SpeakerDatasetPreprocessed(cfg["data"]["test_path"], te["M"]), batch_size=te["N"], shuffle=True,
                        num_workers=te["num_workers"], drop_last=True)
    for e in range(te["epochs"]):
        for batch_id, mel_db_batch in enumerate(loader):
            assert te["M"] % 2 == 0
            size_1, es1 = mel_db_batch.shape[1], 2 * enroll_num
            enr = mel_db_batch[:, :es1].reshape(te["N"] * es1, mel_db_batch.size(2), mel_db_batch.size(3))
            ver = mel_db_batch[:, es1:].reshape(te["N"] * (size_1 - es1), mel_db_batch.size(2), mel_db_batch.size(3))
            e_enr = net(enr.to(device).float()).reshape(te["N"], es1, -1)
            e_ver = net(ver.to(device).float()).reshape(te["N"], size_1 - es1, -1)
            yield e, batch_id, size_1, es1, e_ver, e_enr.mean(dim=1)                  # get_centroids = speaker means


def _load(cfg, model_path, device):
    net = _embedder(cfg, device)
    net.load_state_dict(torch.load(model_path, map_location="cpu"))
    return net.eval()


@torch.no_grad()
def test(cfg, model_path, enroll_num):
    """train_speech_embedder.py:112-203: EER and spoof rate of the mixture test; the similarity matrices are saved for
    the later spoof-rate pass exactly as the reference saves them.  Returns (avg_EER, avg_spoof_rate)."""
    device = torch.device(cfg["device"])
    net = _load(cfg, model_path, device)
    os.makedirs(cfg["save_simmat_dir"], exist_ok=True)
    per_epoch_eer, per_epoch_spoof = {}, {}
    for e, batch_id, size_1, es1, e_ver, cent in _verification_batches(cfg, net, enroll_num, device):
        sim = cossim_eval(e_ver, cent)
        torch.save(sim.cpu(), os.path.join(cfg["save_simmat_dir"], "simmat_e{}_b{}".format(e + 1, batch_id + 1)))
        r = eer_sweep(sim, size_1, es1, spoof=True)
        print("\nEER : %0.4f (thres:%0.4f)" % (r["EER"], r["thres"]))
        per_epoch_eer.setdefault(e, []).append(r["EER"])
        per_epoch_spoof.setdefault(e, []).append(r["spoof_rate"])
    n = cfg["test"]["epochs"]
    avg_eer = sum(sum(v) / len(v) for v in per_epoch_eer.values()) / n
    avg_spoof = sum(sum(v) / len(v) for v in per_epoch_spoof.values()) / n
    print("\n EER across {0} epochs: {1:.4f}".format(n, avg_eer))
    print("\n Spoof rate across {0} epochs: {1:.4f}".format(n, avg_spoof))
    return avg_eer, avg_spoof


@torch.no_grad()
def test_nospoof(cfg, model_path, enroll_num, eval_num):
    """train_speech_embedder.py:205-297: threshold at the EER of the genuine-only test.  Returns the average threshold."""
    device = torch.device(cfg["device"])
    net = _load(cfg, model_path, device)
    per_epoch = {}
    for e, batch_id, size_1, es1, e_ver, cent in _verification_batches(cfg, net, enroll_num, device):
        sim = cossim_eval(e_ver[:, :2 * eval_num], cent)
        r = eer_sweep(sim, size_1, es1, spoof=False)
        print("\nEER : %0.4f (thres:%0.4f, FAR:%0.4f, FRR:%0.4f)" % (r["EER"], r["thres"], r["FAR"], r["FRR"]))
        per_epoch.setdefault(e, []).append(r["thres"])
    avg = sum(sum(v) / len(v) for v in per_epoch.values()) / cfg["test"]["epochs"]
    print("\n Average threshold: ", avg)
    return avg


def spoof_rate_at(cfg, thres, eval_num):
    """train_speech_embedder.py:311-321: fraction of the last 2*eval_num verification trials of each speaker accepted at
    the no-spoof EER threshold, over the saved similarity matrices."""
    N, rates = cfg["test"]["N"], []
    for k in sorted(os.listdir(cfg["save_simmat_dir"])):
        mat = torch.load(os.path.join(cfg["save_simmat_dir"], k)) > thres
        idx = torch.arange(N)
        rates.append(float(mat[idx, -2 * eval_num:, idx].float().sum() / float(2 * eval_num) / N))
    return sum(rates) / len(rates)
