"""Inference harness: host logic on CPU, the file -> separator -> files path on the GPU against the
golden produced by the reference on its own ``sample_WSJ.wav``."""
import os

import numpy as np
import pytest
import torch

from sepreformer_amd import infer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = os.path.join(ROOT, "tests", "golden", "sample_WSJ.wav")


def test_load_pad_normalise_write(tmp_path, golden):
    x = infer.load_wav(SAMPLE, 8000)
    assert x.dtype == np.float32 and x.ndim == 1 and x.shape[0] == 73593
    g = golden("e2e_base_sample_wav")
    padded = infer.pad_to_stride(torch.from_numpy(x)[None], 4)
    assert torch.equal(padded, torch.from_numpy(g["x"]))                       # the reference's own network input
    assert infer.pad_to_stride(padded, 4) is padded
    with pytest.raises(RuntimeError, match="sampling rate"):
        infer.load_wav(SAMPLE, 16000)
    y = infer.peak_normalise(x, 0.9)
    assert abs(np.abs(y).max() - 0.9) < 1e-6
    p = str(tmp_path / "o.wav")
    infer.write_wav(p, y, 8000)
    back = infer.load_wav(p, 8000)
    # libsndfile convention: written as rint(y * 32767), read back as pcm / 32768
    assert back.shape == x.shape and np.array_equal(back, (np.rint(y.astype(np.float64) * 32767.0) / 32768.0).astype(np.float32))
    stereo = str(tmp_path / "st.wav")
    from scipy.io import wavfile
    wavfile.write(stereo, 8000, np.stack([np.full(8, 1000, np.int16), np.full(8, 3000, np.int16)], 1))
    assert np.allclose(infer.load_wav(stereo, 8000), 2000 / 32768.0)


@pytest.mark.gpu
def test_separate_file_matches_reference_golden(tmp_path, golden):
    from oracle import sepreformer_oracle as orc
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    model = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
    raw, written = infer.separate_file(model, SAMPLE, out_prefix=str(tmp_path / "sample"))
    g = golden("e2e_base_sample_wav")
    want = torch.from_numpy(g["main"][:, 0, :73593])
    assert raw.shape == (2, 73593)
    assert orc.agreement_db(torch.from_numpy(raw), want) >= 80.0
    assert [os.path.basename(w) for w in written] == ["sample_in.wav", "sample_out_0.wav", "sample_out_1.wav"]
    for i in range(2):
        w = infer.load_wav(written[1 + i], 8000)
        assert w.shape[0] == 73593 and abs(np.abs(w).max() - 0.9) < 1e-3
        ref = 0.9 * want[i].numpy() / np.abs(want[i].numpy()).max()
        ref_file = np.rint(ref.astype(np.float64) * 32767.0) / 32768.0               # what sf.write + a PCM16 read-back give
        assert np.abs(w - ref_file).max() <= 1.0 / 32768 + 1e-9                    # same file up to one LSB of rounding


@pytest.mark.gpu
def test_test_loop_reports_si_snri(tmp_path):
    """Engine._test's SI-SNRi bookkeeping on synthetic utterances: csv rows, mean over utterances / num_spks, and
    agreement with the oracle criterion applied to the oracle-free device outputs."""
    from oracle import criterion_oracle as co
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    from sepreformer_amd.synth import synth_sources
    model = Model.from_config(VARIANTS["tiny"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
    srcs = torch.from_numpy(synth_sources(3, 2001, seed=5))                         # [3, 2, 2001]
    utts = [(srcs[b].sum(0, keepdim=True), [srcs[b, 0:1], srcs[b, 1:2]], f"utt{b}.wav") for b in range(3)]
    mean, n = infer.test_utterances(model, utts, csv_path=str(tmp_path / "s.csv"), wav_dir=str(tmp_path / "wav"))
    assert n == 3
    rows = [r for r in open(tmp_path / "s.csv").read().strip().split("\n")]
    assert len(rows) == 3 and rows[0].startswith("utt0,")
    want = []
    for mix, src, _ in utts:
        est = [e.cpu() for e in infer.separate(model, mix)]
        m, per, _ = co.pit_sisnri(est, src, mix, dtype=torch.float64)
        want.append(float(m) / 2)
    assert abs(mean - float(np.mean(want))) < 1e-3
    assert sorted(os.listdir(tmp_path / "wav"))[0] == "utt00_mixture.wav"
    with pytest.raises(RuntimeError, match="batch size"):
        infer.test_utterances(model, [(torch.zeros(2, 64), [torch.zeros(2, 64)] * 2, "k")])
