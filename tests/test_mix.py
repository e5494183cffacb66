"""MixUp / CutMix of the training step (extras.py:14-109, classifier.py:86-92).

CPU: `trainer.sample_mix` draws (mode, lambda, box) with the reference's RNG calls in the reference's
order -- checked by replaying fixed seeds and comparing the mixed batch / soft targets the oracle
builds from those draws with what the UNMODIFIED reference produced (tools/gen_golden_mix.py).
GPU: one fused train step with the device-side mixing against the oracle's step on the mixed batch."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import filler
from oracle import torch_ref as R

GOLD = Path(__file__).parent / "golden" / "mix.npz"


def test_sampler_and_oracle_mixing_reproduce_the_reference():
    from vision_toolbox.trainer import sample_mix

    g = np.load(GOLD)
    B, ncls, H, W = (int(v) for v in g["meta"])
    x, y = filler.tensor("mix.x", (B, 3, H, W)), filler.labels(B, ncls, seed=77)
    modes = set()
    for seed in range(12):
        torch.manual_seed(seed)
        mode, lam, box = sample_mix(1.0, 0.2, W, H)
        modes.add(mode)
        xb, tb = R.mix_batch(x, y, ncls, mode, lam, box)
        np.testing.assert_allclose(xb.numpy(), g[f"s{seed}/images"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tb.numpy(), g[f"s{seed}/target"], rtol=1e-6, atol=1e-6)
    assert modes == {"mixup", "cutmix"}  # both branches were exercised by these seeds


@pytest.mark.gpu
@pytest.mark.parametrize("mode,lam,box", [("mixup", 0.3, (0, 0, 0, 0)), ("cutmix", 1 - 20 * 12 / (64 * 64), (10, 30, 30, 42)),
                                          ("none", 1.0, (0, 0, 0, 0))])
def test_fused_train_step_with_device_side_mixing_matches_oracle(mode, lam, box):
    from vision_toolbox import _native as N
    from vision_toolbox import backbones
    from vision_toolbox.trainer import TrainStep

    name, ncls, B, S, lr, wd = "vovnet19_slim_ese", 16, 8, 64, 2e-3, 1e-3
    x, y = filler.images(B, S), filler.labels(B, ncls)
    sd = {}
    for k, shape in R.classifier_spec(name, ncls).items():
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        sd[k] = filler.fill_tensor("mix." + k, torch.zeros(shape, dtype=dt))
    params = {k: v.requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    xb, tb = R.mix_batch(x, y, ncls, mode, lam, box)
    ref_loss, _ = R.classifier_loss_soft(name, sd, xb, tb, 0.1, training=True)
    ref_loss.backward()
    init = {k: v.detach().clone() for k, v in params.items()}
    R.sgd_step(params, {k: v.grad for k, v in params.items()}, {}, lr, 0.9,
               lambda k: R.weight_decay_group(k, wd, 0.0, 0.0))

    ts = TrainStep(getattr(backbones, name)(), ncls, B, S, torch.float32, lr=lr, momentum=0.9, weight_decay=wd,
                   label_smoothing=0.1, device="cuda", use_graphs=False, mix=True)
    filler.fill_module(ts.model, "mix.")
    ts.weights_changed()
    ts.set_mix(mode, lam, box)
    before = N.launch_count()
    ts.step(x.cuda(), y.cuda())
    assert N.launch_count() > before
    assert abs(ts.loss() - ref_loss.item()) < 2e-3 * abs(ref_loss.item())
    got = ts.model.state_dict()
    for k in ("0.stem.0.conv.weight", "3.weight", "3.bias"):
        d_got, d_ref = got[k].cpu() - init[k], params[k].detach() - init[k]
        err = ((d_got - d_ref).norm() / d_ref.norm()).item()
        assert err < 0.03, (k, err)
