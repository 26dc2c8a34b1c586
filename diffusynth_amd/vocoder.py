"""Latent -> audio tail: VQGAN decoder output -> ISTFT+ -> inverse STFT, on the GPU.

Replaces the audio branch of encodeBatch2GradioOutput_STFT
(webUI/natural_language_guided_4/utils.py:219-245: decoder -> .cpu().numpy() -> per-sample
tools.decode_stft / tools.depad_STFT -> librosa.istft(D, hop_length=256, win_length=1024)) with one
batched kernel pair (ds_istft_plus): the (B,3,512,T) decoder output never leaves HBM and the whole
batch is inverted at once.  The UI images of the reference function are out of scope.

librosa is absent offline, so the inverse-STFT stage is parity-unpinned against the reference; it is
checked against the CPU oracle (librosa's documented algorithm, cross-checked with torch.istft /
scipy.signal.istft) in tests/.
"""
import numpy as np
import torch

from . import _lib as L


@torch.no_grad()
def stft_representation_to_audio(enc, hop_length=256):
    """enc: (B, 3, F, T) fp32 CUDA tensor [log1p|D|, cos, sin] (F = n_fft/2 rows, DC row implied zero)
    -> (B, hop*(T-1)) fp32 audio."""
    if not enc.is_cuda:
        raise RuntimeError("diffusynth_amd vocoder runs on MI355X only (ds_istft_plus); no CPU fallback")
    enc = enc.float().contiguous()
    B, three, F, T = enc.shape
    assert three == 3, "expected (B, 3, F, T)"
    ws = torch.empty(L.load().ds_istft_ws_floats(B, F, T), dtype=torch.float32, device=enc.device)
    audio = torch.empty((B, hop_length * (T - 1)), dtype=torch.float32, device=enc.device)
    L.call("ds_istft_plus", enc.data_ptr(), B, F, T, hop_length, ws.data_ptr(), audio.data_ptr(), L.current_stream())
    return audio


@torch.no_grad()
def latents_to_audio(decoder, quantized_latents):
    """(B, 4, H, W) quantised latents -> (B, 256*(4W-1)) audio: decoder + ISTFT+ + iSTFT, all on device."""
    return stft_representation_to_audio(decoder(quantized_latents))


def encodeBatch2GradioOutput_STFT(decoder, latent_vector_batch, resolution=(512, 256), original_STFT_batch=None):
    """Signature of utils.py:194.  Returns the reference's 6-tuple with the image slots set to None (UI rendering is
    out of scope) and the signals as float64 numpy arrays like librosa returns them."""
    dev = next(decoder.parameters()).device
    if isinstance(latent_vector_batch, np.ndarray):
        latent_vector_batch = torch.from_numpy(latent_vector_batch)
    rec = decoder(latent_vector_batch.to(dev))
    signals = [s.astype(np.float64) for s in stft_representation_to_audio(rec).cpu().numpy()]
    with_amp = []
    if original_STFT_batch is not None:
        mixed = rec.clone()
        mixed[:, 0] = torch.as_tensor(original_STFT_batch)[:, 0].to(dev)
        with_amp = [s.astype(np.float64) for s in stft_representation_to_audio(mixed).cpu().numpy()]
    return None, None, signals, None, None, with_amp


@torch.no_grad()
def audio_to_stft_representation(audio, time_resolution=256, hop_length=256, pad_mode="constant"):
    """(B, L) or (L,) fp32 audio -> (B, 3, 512, T') STFT+ representation, T' = max(time_resolution, 1 + L//hop):
    librosa.stft(n_fft=1024, hop, win 1024) + tools.pad_STFT + tools.encode_stft in one kernel (ds_stft_plus).
    pad_mode "constant" (librosa >= 0.10 default) or "reflect" (older librosa)."""
    if not audio.is_cuda:
        raise RuntimeError("diffusynth_amd STFT runs on MI355X only (ds_stft_plus); no CPU fallback")
    assert pad_mode in ("constant", "reflect")
    a = audio.float().contiguous()
    if a.dim() == 1:
        a = a.unsqueeze(0)
    B, Ln = a.shape
    T = 1 + Ln // hop_length
    T_out = max(T, time_resolution or T)
    enc = torch.empty((B, 3, 512, T_out), dtype=torch.float32, device=a.device)
    L.call("ds_stft_plus", a.data_ptr(), B, Ln, hop_length, 1 if pad_mode == "reflect" else 0, T_out, enc.data_ptr(), L.current_stream())
    return enc


@torch.no_grad()
def InputBatch2Encode_STFT(encoder, STFT_batch, resolution=(512, 256), quantizer=None, squared=True):
    """Latent branch of utils.py:131-191: (latents, quantised latents); the images / reconstructed signals of the reference
    tuple are UI products and returned as None."""
    dev = next(encoder.parameters()).device
    z = encoder(STFT_batch.to(dev))
    q = quantizer(z)[0] if quantizer is not None else None
    return None, None, None, z, q
