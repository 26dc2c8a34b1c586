"""Oracle (test infrastructure): CPU restatement of the ISTFT+ / iSTFT vocoder tail.

decode_stft / depad_STFT follow tools.py:334-345 and :185-191 (fp32 polar form,
widened to complex128 by the zero DC row).  The inverse STFT itself is
``librosa.istft(D, hop_length=256, win_length=1024)`` (call site
webUI/natural_language_guided_4/utils.py:241).  librosa is a third-party
dependency, unpinned in requirements.txt:4, absent from /root/reference and not
installed: this stage is PARITY UNPINNED.  The restatement below follows
librosa's published algorithm and defaults — n_fft = 2*(rows-1), periodic Hann
window of win_length padded to n_fft, center=True, per-frame irFFT * window,
overlap-add, division by the window sum-of-squares where it exceeds
tiny(float), trim n_fft//2 on both sides — and is cross-checked against
torch.istft and scipy.signal.istft in tests/test_oracle_golden.py.
"""
import numpy as np


def decode_stft(enc):
    """tools.py:334-345 — (3,F,T) [log1p|D|, cos, sin] -> complex (F,T), dtype follows input."""
    mag = np.expm1(enc[0])
    ph = np.arctan2(enc[2], enc[1])
    return mag * (np.cos(ph) + 1j * np.sin(ph))


def depad_stft(d):
    """tools.py:185-191 — prepend the zero DC row (float64 zeros => complex128 result)."""
    return np.concatenate([np.zeros((1, d.shape[1])), d], axis=0)


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def istft(D, hop_length=256, win_length=1024):
    """librosa.istft semantics (see module docstring).  D: (1+n_fft/2, frames) complex."""
    n_fft = 2 * (D.shape[0] - 1)
    frames = D.shape[1]
    win = hann_periodic(win_length)
    if win_length < n_fft:
        lp = (n_fft - win_length) // 2
        win = np.pad(win, (lp, n_fft - win_length - lp))
    sig = np.fft.irfft(D, n=n_fft, axis=0) * win[:, None]            # (n_fft, frames)
    total = n_fft + hop_length * (frames - 1)
    y = np.zeros(total, dtype=sig.dtype)
    wss = np.zeros(total, dtype=sig.dtype)
    wsq = win ** 2
    for t in range(frames):
        y[t * hop_length:t * hop_length + n_fft] += sig[:, t]
        wss[t * hop_length:t * hop_length + n_fft] += wsq
    ok = wss > np.finfo(sig.dtype).tiny
    y[ok] /= wss[ok]
    return y[n_fft // 2: total - n_fft // 2]


def latents_to_audio(decoder_out):
    """utils.py:219-245 audio branch — decoder output (B,3,F,T) fp32 numpy -> list of float64 signals."""
    return [istft(depad_stft(decode_stft(s)), 256, 1024) for s in decoder_out]


# ---------------------------------------------------------------------------------------- audio -> STFT representation
def pad_stft(d, time_resolution=256):
    """tools.py:170-182 — drop the DC row, zero-pad the time axis up to time_resolution (longer inputs are kept)."""
    d = d[1:, :]
    if time_resolution is None:
        return d
    padn = time_resolution - d.shape[1]
    return np.pad(d, ((0, 0), (0, padn)), "constant") if padn > 0 else d


def encode_stft(d):
    """tools.py:320-331 — [log1p|D|, cos(angle D), sin(angle D)]."""
    mag, ph = np.abs(d), np.angle(d)
    return np.stack([np.log1p(mag), np.cos(ph), np.sin(ph)], axis=0)


def stft(y, n_fft=1024, hop_length=256, win_length=1024, pad_mode="constant"):
    """librosa.stft semantics (PARITY UNPINNED, see module docstring): center=True, periodic Hann, frames = 1 + len//hop,
    complex64 output for float32 input.  pad_mode: librosa >= 0.10 defaults to "constant" (zeros), older to "reflect"."""
    y = np.asarray(y, dtype=np.float32)
    ypad = np.pad(y, n_fft // 2, mode=pad_mode)
    win = hann_periodic(win_length).astype(np.float32)
    n_frames = 1 + (len(ypad) - n_fft) // hop_length
    frames = np.stack([ypad[t * hop_length:t * hop_length + n_fft] * win for t in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0).astype(np.complex64)
