"""The train-mode oracle (oracle/train_oracle.py) against the committed reference training-step data (tests/golden/train_tiny.npz,
made by tests/golden/make_train_golden.py from the IMPORTED reference Model in train() mode + the reference's own criteria)."""
import dataclasses

import numpy as np
import torch

from oracle import sepreformer_oracle as orc
from oracle import train_oracle as tor
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.synth import _gen, synth_state_dict


def test_train_oracle_reproduces_reference_step(golden):
    g = golden("train_tiny")
    cfg = dataclasses.replace(VARIANTS["tiny"], dropout=0.0)
    x = torch.from_numpy(g["x"])
    src = [torch.from_numpy(g["src"][:, s].copy()) for s in range(cfg.num_spks)]
    sd = tor.leaf_state(synth_state_dict(cfg, 0))
    audio, aux = tor.model_forward_train(sd, cfg, x)
    loss, l_time, l_mag = tor.train_loss(audio, aux, src)
    loss.backward()
    assert orc.BN_TRAINING is False                                    # the switch is restored
    assert abs(float(loss) - float(g["loss"])) < 1e-4 and abs(float(l_time) - float(g["loss_time"])) < 1e-4
    assert np.abs(np.asarray([float(v) for v in l_mag]) - g["loss_mag"]).max() < 1e-4
    assert orc.agreement_db(torch.stack([a.detach() for a in audio], 0), torch.from_numpy(g["main"])) > 100
    names = [str(n) for n in g["grad_names"]]
    assert len(names) == 710 and set(names) == {k for k, v in sd.items() if v.requires_grad}
    for i, k in enumerate(names):                                      # reference gradient summaries, every tensor
        gr = sd[k].grad.double()
        pv = torch.from_numpy(_gen(4242, k).normal(0.0, 1.0, size=tuple(gr.shape)).astype(np.float32)).double()
        got = np.array([float(gr.norm()), float(gr.sum()), float((gr * pv).sum())])
        if k.endswith(("linear_k.bias", "dw_conv_1d.bias", "cla.linear2.bias", "down_conv.bias")):
            continue                                                   # identically zero in exact arithmetic: rounding noise only
        assert np.abs(got - g["grad_summary"][i]).max() <= 1e-4 * (g["grad_summary"][i][0] + 1e-12), k
    off = 0
    for k in [str(n) for n in g["small_names"]]:                       # small tensors are stored whole
        n = sd[k].numel()
        assert np.allclose(sd[k].grad.reshape(-1).numpy(), g["small_grads"][off:off + n], rtol=1e-4, atol=1e-7), k
        off += n
    off = 0
    for k in [str(n) for n in g["bn_names"]]:                          # running statistics after the step (momentum 0.1)
        n = sd[k].numel()
        assert np.allclose(sd[k].reshape(-1).numpy(), g["bn_after"][off:off + n], atol=1e-6), k
        off += n


def test_eval_oracle_unchanged_by_train_switch(golden):
    """The eval forward (BatchNorm on running statistics) is what the inference goldens pin; it must not see the switch."""
    g = golden("e2e_tiny_b1")
    cfg = VARIANTS["tiny"]
    audio, _ = orc.model_forward(synth_state_dict(cfg, 0), cfg, torch.from_numpy(g["x"]))
    assert orc.agreement_db(torch.stack(list(audio), 0), torch.from_numpy(g["main"])) > 100
