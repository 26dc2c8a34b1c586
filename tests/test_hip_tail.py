"""GPU parity of the latent -> audio tail (VQ, VQGAN decoder, ISTFT+/iSTFT) against reference goldens / the oracle."""
import numpy as np
import pytest
import torch

from conftest import golden_keys, load_golden, rel_err
from diffusynth_amd.synth import synth_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vae(vqgan_sd):
    from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN
    m = VQGAN(**PRODUCTION_CONFIG)
    m.load_state_dict(vqgan_sd)
    return m.to("cuda")


def test_quantiser_matches_reference(vae):
    g = load_golden("tail")
    z = torch.from_numpy(g["vq_z"]).cuda()
    q, loss, (perp, a, b) = vae._vq_vae(z)
    assert a is None and b is None
    idx = vae._vq_vae.last_indices.flatten().cpu()
    assert (idx == torch.from_numpy(g["vq_idx"])).float().mean().item() > 0.999
    same = (idx == torch.from_numpy(g["vq_idx"])).view(z.shape[0], z.shape[2], z.shape[3])[:, None].expand_as(z)
    assert torch.equal(q.cpu()[same], torch.from_numpy(g["vq_q"])[same])
    assert abs(loss.item() - g["vq_loss"].item()) < 1e-3 * abs(g["vq_loss"].item())
    assert abs(perp.item() - g["vq_perplexity"].item()) < 1e-2 * g["vq_perplexity"].item()


@pytest.mark.parametrize("name", ["dec", "dec2"])
def test_decoder_fp32_matches_reference(vae, name):
    g = load_golden("tail")
    vae._decoder.set_compute_dtype("fp32")
    y = vae._decoder(torch.from_numpy(g[name + "_q"]).cuda())
    err = rel_err(y.cpu(), g[name + "_y"])
    print(f"decoder fp32 {name}: rel err {err:.2e}")
    assert y.shape == g[name + "_y"].shape and err < 1e-3


def test_decoder_bf16_error_is_bounded(vae):
    g = load_golden("tail")
    vae._decoder.set_compute_dtype("bf16")
    y = vae._decoder(torch.from_numpy(g["dec_q"]).cuda())
    vae._decoder.set_compute_dtype("fp32")
    err = rel_err(y.cpu(), g["dec_y"])
    print(f"decoder bf16: rel err {err:.2e}")
    assert err < 5e-2


def test_latents_to_audio_matches_oracle(vae, vqgan_sd):
    """Config-5 style end-to-end tail from IDENTICAL quantised latents: decoder + ISTFT+ + iSTFT vs the CPU oracle."""
    from diffusynth_amd.vocoder import encodeBatch2GradioOutput_STFT, latents_to_audio
    from oracle import vocoder_ref as V
    from oracle import vqgan_ref as Q
    q = synth_input("tail_e2e_q", (2, 4, 128, 3))          # decoder output (2,3,512,12): F=512 like production
    vae._decoder.set_compute_dtype("fp32")
    audio = latents_to_audio(vae._decoder, q.cuda())
    dec_ref = Q.decoder_forward(vqgan_sd, Q.PRODUCTION_CONFIG, q)
    ref = np.stack(V.latents_to_audio(dec_ref.numpy()))
    assert audio.shape == ref.shape == (2, 256 * 11)
    err = rel_err(audio.cpu(), ref)
    print(f"latents->audio fp32: rel err {err:.2e}")
    assert err < 1e-3
    out = encodeBatch2GradioOutput_STFT(vae._decoder, q.numpy())
    assert len(out) == 6 and out[2][0].dtype == np.float64 and rel_err(np.stack(out[2]), ref) < 1e-3
