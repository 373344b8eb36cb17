"""BASELINE.json's configurations at their OWN sizes through the C ABI, in the arithmetic bench.py times (split-fp16), against the pinned CPU
oracles: config 3's adversarial generator and critic iterations at B = 32 (dropout masks injected on both sides) and config 5's
880 x 120 x 40 d-vector extraction + the 88 x 10 GE2E loss.  (Configs 1, 2 and the non-adversarial step of config 3 are in
test_gpu_parity.py.)  Run with `-m gpu` on an MI355X."""
import pytest
import torch

from _golden import rel_err, rel_l2
from oracle import critic_oracle as CO
from oracle import ge2e_oracle as GO
from oracle import tts_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _threads():
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))


# ------------------------------------------------------------------------------------------------------------------ config 5
def test_config5_dvector_extraction_880x120x40_and_loss_vs_oracle():
    """GE2E/speech_embedder_net.py:27-33 at config 5's exact workload: randn(880, 120, 40), seed 0, the default SpeechEmbedder
    (3 x LSTM(768) -> Linear(768, 256) -> L2 norm).  ALL 880 embeddings against oracle/ge2e_oracle.py on the host cores: a 120-frame recurrence
    is where LSTM rounding accumulates (the other full-width tests run 5-12 frames).  Then GE2ELoss at 88 speakers x 10 utterances
    (speech_embedder_net.py:43-49, GE2E/utils.py:16-55) on the HIP embeddings against the oracle's loss on the oracle's embeddings."""
    from spoofsv_amd.ge2e import GE2ELoss, SpeechEmbedder
    _threads()
    torch.manual_seed(0)
    m = SpeechEmbedder()
    x = torch.randn(880, 120, 40)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        eo = GO.speech_embedder(x, sd)
        lo, per_o = GO.ge2e_loss(eo.view(88, 10, 256), torch.tensor(10.0), torch.tensor(-5.0))
    m = m.to(DEV).eval()
    L = GE2ELoss(DEV)
    with torch.no_grad():
        eg = m(x.to(DEV))
        lg, per_g = L(eg.view(88, 10, 256), return_per_embedding=True)
    assert tuple(eg.shape) == (880, 256)
    e_max, e_l2 = rel_err(eg, eo), rel_l2(eg, eo)
    worst_row = float((eg.cpu() - eo).norm(dim=1).max())               # embeddings are unit vectors: an absolute distance per utterance
    print("config 5, 880 x 120 x 40: embeddings max-norm %.2e, rel L2 %.2e, worst utterance distance %.2e; loss %.4f vs %.4f"
          % (e_max, e_l2, worst_row, float(lg), float(lo)))
    assert e_max < 2e-5 and e_l2 < 2e-5 and worst_row < 2e-5, (e_max, e_l2, worst_row)
    assert abs(float(lg) - float(lo)) < 1e-4 * abs(float(lo)), (float(lg), float(lo))
    assert rel_err(per_g.reshape(-1), per_o.reshape(-1)) < 1e-4


# ------------------------------------------------------------------------------------------------------------------ config 3 (--adversarial)
@pytest.mark.parametrize("kind", ["text2mel", "ssrn"])
def test_config3_adversarial_generator_iterations_at_batch_32_vs_oracle(kind):
    """train/adversarial_wasserstein_gp.py:278-297 (:329-343 for the SSRN) at B = 32: generator iterations 0 and 1 of the WGAN-GP trainer on
    the HIP path (train.AdversarialGraphStep) with the critic's dropout masks injected, against oracle/tts_oracle.py + oracle/critic_oracle.py
    with the same masks and torch's Adam.  Iteration 0 checks generator + critic forward; iteration 1 (after one optimizer step) the gradient
    THROUGH the critic, the adaptive weight (l1 + bd + att) / |disc| and Adam.  This is the check bench.py runs on its line, as a test."""
    import bench
    _threads()
    got = bench.adversarial_first_g_iterations(kind, 32, DEV)
    want = bench.adversarial_first_g_iterations_oracle(kind, 32)
    names = ["l1", "bd"] + (["att"] if kind == "text2mel" else []) + ["disc"]
    for it in range(2):
        for n, a, b in zip(names, got[it], want[it]):
            assert abs(a - b) < 1e-4 * max(abs(b), 1e-12), (kind, it, n, a, b)


_CRITIC_ZERO_GRAD = ("conv1.bias", "conv2.bias", "conv3.bias", "conv4.bias", "hc.conv.bias")      # a bias feeding a LayerNorm: exactly-zero true gradient


@pytest.mark.parametrize("kind", ["text2mel", "ssrn"])
def test_config3_adversarial_critic_iteration_at_batch_32_vs_oracle(kind):
    """train/adversarial_wasserstein_gp.py:299-322 at B = 32, through the trainer's own critic iteration (AdversarialGraphStep._d_compute,
    eager): generator forward (no grad), interpolate with the trainer's coefficients, gradient penalty by double backward, Wasserstein term,
    every critic parameter gradient -- against the oracle's generator output fed to oracle/critic_oracle.critic_losses with the same
    coefficients and the same nine dropout masks.  The trainer evaluates disc(pred) and disc(gt) as ONE call on the concatenated batch
    (pred first); the masks of that call are the oracle's pred and gt masks concatenated, so the sums are the same."""
    import bench
    from spoofsv_amd import critic, train
    _threads()
    B = 32
    model, disc = bench._adv_build(kind)
    gsd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dsd = {k: v.detach().clone().requires_grad_(True) for k, v in disc.state_dict().items()}
    gaw = None
    if kind == "text2mel":
        data = train.synthetic_text2mel_batch(B, 186, 325, seed=0)
        gaw = train.guided_attention_mat(186, 325)
    else:
        data = train.synthetic_ssrn_batch(B, 325, seed=0)
    masks = bench._adv_masks(kind, B, 3)                                # oracle order: interpolate, ground truth, prediction
    coeff = torch.rand(B, generator=torch.Generator().manual_seed(0))   # AdversarialGraphStep(coeff_seed=0): its first draw
    # ---- oracle
    with torch.no_grad():
        if kind == "text2mel":
            pred_o, _ = TO.melsyn_train(train.shift_right(data[0]), data[1], data[2], gsd)
            gt = data[0]
        else:
            pred_o = TO.ssrn(data[0], gsd)
            gt = data[1]
    gp_o, ld_o = CO.critic_losses(pred_o, gt, coeff, dsd, "mel" if kind == "text2mel" else "lin", 10.0, masks=list(masks))
    gp_o.backward()
    ld_o.backward()
    gp_o, ld_o = gp_o.detach(), ld_o.detach()
    # ---- HIP trainer
    model.to(DEV).train(); disc.to(DEV).train()
    og = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(disc.parameters(), 0.0, (0.5, 0.9), 1e-6, capturable=True)      # lr 0: the gradients stay to be read
    og.refresh_resident_weights()
    stepper = train.AdversarialGraphStep(kind, model, disc, og, od, [d.to(DEV) for d in data], gaw.to(DEV) if gaw is not None else None,
                                         10.0, None, None, graph=False, coeff_seed=0)
    hip_masks = [m.to(DEV) for m in masks[0:3]] + [torch.cat((p, g), dim=0).to(DEV) for p, g in zip(masks[6:9], masks[3:6])]
    with critic.injected_dropout_masks(hip_masks):
        ld_g, gp_g = stepper.d_step()
    torch.cuda.synchronize()
    assert torch.equal(stepper.coeff.view(-1).cpu(), coeff)
    print("config 3 critic iteration (%s, B = 32): penalty %.6f vs %.6f, loss_D %.6f vs %.6f" % (kind, float(gp_g), float(gp_o), float(ld_g), float(ld_o)))
    assert abs(float(gp_g) - float(gp_o)) < 2e-4 * max(1.0, abs(float(gp_o))), (float(gp_g), float(gp_o))
    assert abs(float(ld_g) - float(ld_o)) < 2e-4 * max(1.0, abs(float(ld_o))), (float(ld_g), float(ld_o))
    for n, p in disc.named_parameters():
        if n in _CRITIC_ZERO_GRAD:
            continue
        want = dsd[n].grad
        e = float((p.grad.detach().cpu() - want).norm() / max(1e-6, float(want.norm())))
        assert e < 3e-3, (kind, n, e)
