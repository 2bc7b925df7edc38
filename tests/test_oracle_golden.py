"""The oracle (CPU restatement) against the committed golden vectors.

The vectors were produced by running the IMPORTED REFERENCE in the build container
(tests/golden/make_golden.py); this re-checks the oracle against them wherever the suite runs, so an
oracle edit (or a torch upgrade that changes an aten op's semantics) cannot go unnoticed.
Tolerance: the oracle executes the same aten ops as the reference, so agreement is at thread-count
noise level; we require >= 100 dB (SURVEY.md section 7 measured 122 dB between 1 and 8 threads)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sepreformer_oracle as orc
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.synth import synth_state_dict

MIN_DB = 100.0
E2E = [("tiny", "tiny"), ("tiny", "tiny_b1"), ("tiny3", "tiny_s3"), ("SepReformer_Base_WSJ0", "base_0p5s"),
       ("SepReformer_Base_WSJ0", "base_4s"), ("SepReformer_Large_DM_WHAMR", "large_whamr_0p5s"),
       ("SepReformer_Large_DM_WHAM", "large_wham_0p5s")]


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("variant,tag", E2E)
def test_e2e_matches_reference_vectors(golden, variant, tag):
    g = golden("e2e_" + tag)
    cfg = VARIANTS[variant]
    sd = synth_state_dict(cfg, 0)
    audio, aux = orc.model_forward(sd, cfg, T(g["x"]))
    main = torch.stack(list(audio), 0)
    assert tuple(main.shape) == g["main"].shape
    assert orc.agreement_db(main, T(g["main"])) >= MIN_DB
    auxt = torch.stack([torch.stack(list(a), 0) for a in aux], 0)
    assert orc.agreement_db(auxt[..., ::16], T(g["aux_dec16"])) >= MIN_DB
    if "aux" in g:
        assert orc.agreement_db(auxt, T(g["aux"])) >= MIN_DB
    np.testing.assert_allclose(auxt.double().abs().mean(-1).numpy(), g["aux_abs_mean"], rtol=1e-5)


def test_sample_wav_case(golden):
    """BASELINE config 1: the reference's own sample_wav/sample_WSJ.wav through Engine._inference_sample's
    I/O rules (reference engine.py:154-163): PCM16/32768, right-pad to a multiple of 4."""
    from scipy.io import wavfile
    g = golden("e2e_base_sample_wav")
    sr, pcm = wavfile.read(os.path.join(os.path.dirname(__file__), "golden", "sample_WSJ.wav"))
    assert sr == 8000 and pcm.shape == (73593,)
    x = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None]
    x = torch.nn.functional.pad(x, (0, (-x.shape[-1]) % 4))
    assert np.array_equal(x.numpy(), g["x"])
    cfg = VARIANTS["SepReformer_Base_WSJ0"]
    audio, _ = orc.model_forward(synth_state_dict(cfg, 0), cfg, x)
    assert orc.agreement_db(torch.stack(list(audio), 0), T(g["main"])) >= MIN_DB


def test_blocks_match_reference_vectors(golden):
    g = golden("blocks_tiny")
    cfg = VARIANTS["tiny"]
    sd = synth_state_dict(cfg, 0)
    H, S = cfg.heads, cfg.num_spks
    e0 = "separator.enc_stages.0"

    def ok(a, key):
        assert orc.agreement_db(a, T(g[key])) >= MIN_DB, key

    ok(orc.gcfn(sd, e0 + ".g_block_1.block.gcfn", T(g["gcfn.x"])), "gcfn.y")
    for t in (24, 150):
        ok(orc.cla(sd, e0 + ".l_block_1.block.cla", T(g[f"cla.x{t}"])), f"cla.y{t}")
    for t in (20, 100):
        ok(orc.mha(sd, e0 + ".g_block_1.block.ega.block.self_attn", T(g[f"mha.x{t}"]),
                   orc.rel_pos_k(sd, t, cfg.maxlen), H), f"mha.y{t}")
    for fac in (1, 2, 4):
        ok(orc.ega(sd, e0 + ".g_block_1.block.ega", T(g[f"ega.x{fac}"]), orc.rel_pos_k(sd, 25, cfg.maxlen), H), f"ega.y{fac}")
    ok(orc.spk_attention(sd, "separator.dec_stages.0.spk_attn_1", T(g["spk.x"]), S, H), "spk.y")
    for t in (40, 41):
        ok(orc.down_conv(sd, e0 + ".downconv", T(g[f"down.x{t}"])), f"down.y{t}")
    ok(orc.spk_split(sd, "separator.spk_split_block", T(g["split.x"]), S), "split.y")
    enc = orc.audio_encoder(sd, T(g["enc.x"]), cfg.enc_stride)
    ok(enc, "enc.y")
    ok(orc.feature_projector(sd, enc), "proj.y")
    ok(orc.output_layer(sd, "out_layer", T(g["out.x"]), enc, S, False), "out.y0")
    ok(orc.output_layer(sd, "out_layer_bn.1", T(g["out.x"]), enc, S, True), "out.y1")
    ok(orc.audio_decoder(sd["audio_decoder.weight"], T(g["dec.x"]), cfg.enc_stride), "dec.y")


def test_pinning_report_is_clean():
    with open(os.path.join(os.path.dirname(__file__), "golden", "PINNING.json")) as f:
        rep = json.load(f)
    assert rep["worst"] < 5e-6
    # the oracle is timed as the CPU baseline: it must cost about what the reference costs
    r = rep["timing"]["base_4s"]["oracle_over_ref"]
    assert 0.8 < r < 1.2, r


def test_metrics():
    a = torch.randn(2, 4000)
    assert orc.agreement_db(a, a) > 200
    n = a + 1e-3 * torch.randn_like(a)
    assert 55 < orc.agreement_db(n, a) < 65
    assert torch.all(orc.si_snr_db(3.0 * a + 0.5, a) > 60)
    b = torch.randn(2, 4000)
    p = orc.pit_si_snr_db([b + 0.01 * a, a + 0.01 * b], [a, b])
    assert torch.all(p > 35)
