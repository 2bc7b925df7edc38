"""Inference harness with the I/O conventions of the reference's engines.

Mirrors ``models/SepReformer_Base_WSJ0/engine.py``:

* ``separate_file``  - ``Engine._inference_sample`` (:151-172): load a wav at the model's sampling rate, zero-pad
  to a multiple of the encoder stride, run the separator, crop to the input length, write
  ``<name>_in.wav`` and ``<name>_out_<i>.wav`` peak-normalised to 0.9;
* ``test_utterances`` - the SI-SNRi part of ``Engine._test`` (:113-149): one utterance per step, device-side
  ``PIT_SISNRi`` (eps 1e-15), one csv row per utterance, running mean divided by ``num_spks``, optional
  ``0.5 / max|.|`` wav dumps.  (``PIT_SDRi`` is mir_eval's BSS-eval on the CPU in the reference and stays out of scope.)

Host-side logic only; every waveform sample is computed by the HIP separator (``Model.forward``) and the HIP
criterion kernels.  File I/O uses scipy (the reference uses librosa / soundfile, absent here): PCM16/PCM32/float
wavs, multi-channel input averaged to mono as ``librosa.load`` does; a sampling-rate mismatch raises instead of
resampling silently.
"""
from __future__ import annotations

import csv
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def load_wav(path: str, fs: int) -> np.ndarray:
    """-> float32 mono in [-1, 1) (``librosa.load(path, sr=fs)`` for a file already at ``fs``)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if sr != fs:
        raise RuntimeError(f"{path}: sampling rate {sr} != model rate {fs} (resample the file first)")
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    return np.ascontiguousarray(x)


def pad_to_stride(x: torch.Tensor, stride: int) -> torch.Tensor:
    """Right zero-padding to a multiple of the encoder stride (engine.py:157-163)."""
    remains = x.shape[-1] % stride
    return x if remains == 0 else torch.nn.functional.pad(x, (0, stride - remains), "constant", 0)


def peak_normalise(x: np.ndarray, peak: float) -> np.ndarray:
    """``peak * x / max|x|`` (engine.py:168,171; 0.9 for infer_sample, 0.5 for test_save)."""
    return peak * x / np.max(np.abs(x))


def write_wav(path: str, x: np.ndarray, fs: int) -> None:
    """float waveform -> PCM16 file (soundfile's default subtype for .wav, which the reference relies on).
    libsndfile's float -> short conversion is ``lrint(x * 0x7FFF)`` (normalised floats, round-half-even); the
    same scale is used here so the written files are sample-identical to the reference's ``sf.write`` output
    (the harness peak-normalises to 0.9, reference engine.py:168-172, so nothing clips; the clip is a guard)."""
    from scipy.io import wavfile
    pcm = np.clip(np.rint(np.asarray(x, dtype=np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    wavfile.write(path, fs, pcm)


@torch.no_grad()
def separate(model, mixture: torch.Tensor, stride: Optional[int] = None) -> List[torch.Tensor]:
    """``mixture`` ``[B,T]`` (any T) -> ``num_spks`` tensors ``[B,T]`` on the model's device."""
    stride = stride or model.cfg.enc_stride
    dev = next(model.parameters()).device
    T = mixture.shape[-1]
    x = pad_to_stride(mixture.to(torch.float32), stride).to(dev)
    model.eval()
    audio, _ = model(x)
    return [a[..., :T] for a in audio]


def separate_file(model, path: str, fs: int = 8000, out_prefix: Optional[str] = None) -> Tuple[np.ndarray, List[str]]:
    """``Engine._inference_sample``.  Returns the raw (un-normalised) estimates ``[S,T]`` and the files written."""
    mix = load_wav(path, fs)
    est = separate(model, torch.from_numpy(mix)[None])
    prefix = out_prefix if out_prefix is not None else path[:-4]
    written = [prefix + "_in.wav"]
    write_wav(written[0], peak_normalise(mix, 0.9), fs)
    raw = []
    for i, e in enumerate(est):
        src = e[0].detach().cpu().numpy()
        raw.append(src)
        written.append(f"{prefix}_out_{i}.wav")
        write_wav(written[-1], peak_normalise(src, 0.9), fs)
    return np.stack(raw), written


def test_utterances(model, utterances: Iterable[Tuple[torch.Tensor, Sequence[torch.Tensor], str]],
                    csv_path: Optional[str] = None, wav_dir: Optional[str] = None, fs: int = 8000) -> Tuple[float, int]:
    """SI-SNRi loop of ``Engine._test``: ``utterances`` yields ``(mixture [1,T], [source_s [1,T]], key)``.
    Returns (mean SI-SNRi per speaker in dB, number of utterances); writes one csv row per utterance."""
    from .criterion import PIT_SISNRi
    dev = next(model.parameters()).device
    crit = PIT_SISNRi(dev, model.num_spks, True)
    total, n = 0.0, 0
    fh = open(csv_path, "w", newline="") if csv_path else None
    writer = csv.writer(fh, quotechar="|", quoting=csv.QUOTE_MINIMAL) if fh else None
    try:
        for mixture, sources, key in utterances:
            if mixture.shape[0] != 1:
                raise RuntimeError("batch size is not one!!")              # engine.py:126-127
            est = separate(model, mixture)
            m, per = crit(estims=est, mixture=mixture.to(dev), input_sizes=torch.tensor([mixture.shape[-1]]),
                          target_attr=[s.to(dev) for s in sources], eps=1.0e-15)
            total += float(m) / model.num_spks
            n += 1
            name = key[:-4] if key.lower().endswith(".wav") else key
            if writer:
                writer.writerow([name] + [float(per[i]) for i in range(model.num_spks)])
            if wav_dir:
                os.makedirs(wav_dir, exist_ok=True)
                write_wav(os.path.join(wav_dir, f"{name}{n - 1}_mixture.wav"), peak_normalise(mixture[0].cpu().numpy(), 0.5), fs)
                for i, e in enumerate(est):
                    write_wav(os.path.join(wav_dir, f"{name}{n - 1}_out_{i}.wav"), peak_normalise(e[0].cpu().numpy(), 0.5), fs)
    finally:
        if fh:
            fh.close()
    return (total / n if n else 0.0), n


test_utterances.__test__ = False      # not a pytest test


def _main() -> None:
    import argparse
    from .config import VARIANTS
    from .model import Model
    ap = argparse.ArgumentParser(description="separate one wav file (reference: run.py --engine-mode infer_sample)")
    ap.add_argument("wav")
    ap.add_argument("--model", default="SepReformer_Base_WSJ0", choices=sorted(VARIANTS))
    ap.add_argument("--checkpoint", default=None, help="reference checkpoint (.pth with model_state_dict); default: synthetic weights")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    model = Model.from_config(VARIANTS[args.model], init_seed=0)
    if args.checkpoint:
        ck = torch.load(args.checkpoint, map_location="cpu")
        model.load_state_dict(ck.get("model_state_dict", ck), strict=False)      # utils/util_engine.py:43
    else:
        model.load_synthetic_(0)
    model = model.eval().to(args.device)
    _, written = separate_file(model, args.wav)
    print("\n".join(written))


if __name__ == "__main__":
    _main()
