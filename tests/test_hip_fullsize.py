"""GPU tests at BASELINE.json's FULL sizes (configs[2]: batch 64 + CFG => U-Net batch 128; configs[4]: 64 latents ->
VQ -> decoder -> (64, 65280) audio) through size-independent properties, plus the cases ADVICE r01 found missing:
per-channel inpaint masks, the shard contract with max_batchsize > batch / Philox noise, and a 2-rank rehearsal of
the multi-GPU path on the one card a test box has."""
import socket

import numpy as np
import pytest
import torch

from conftest import rel_err
from diffusynth_amd.synth import synth_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def unet(unet_sd):
    from diffusynth_amd.unet import ConditionedUnet, PRODUCTION_CONFIG
    m = ConditionedUnet(**PRODUCTION_CONFIG)
    m.load_state_dict(unet_sd)
    return m.to("cuda")


@pytest.fixture(scope="module")
def vae(vqgan_sd):
    from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN
    m = VQGAN(**PRODUCTION_CONFIG)
    m.load_state_dict(vqgan_sd)
    return m.to("cuda")


def _sampler(K, H, mb, **kw):
    from diffusynth_amd.sampler import DiffSynthSampler
    kw.setdefault("noise_device", "cpu")
    s = DiffSynthSampler(1000, mute=True, device="cuda", height=H, max_batchsize=mb, **kw)
    s.respace(list(np.linspace(0, 999, K, dtype=np.int32)))
    return s


# ----------------------------------------------------------------------------------------------- ADVICE r01 (high)
@pytest.mark.parametrize("kind", ["repeat_channels", "channel_varying", "broadcast_hw"])
def test_inpaint_accepts_every_mask_the_reference_broadcasts(unet, unet_sd, kind):
    """The reference blends `mask * noisy + (1 - mask) * img` by broadcasting (DSS:506) and its inpaint UI passes a
    (B, C, H, W) mask (inpaint_with_text.py:229-231): per-channel and lower-rank masks vs the CPU oracle."""
    from oracle.sampler_ref import RefSampler
    from oracle.unet_ref import RefUnet
    B, H, W, K = 2, 32, 64, 3
    cond = synth_input("inp_c", (B, 512))
    guide = synth_input("inp_g", (B, 4, H, W))
    base = torch.zeros(B, 1, H, W)
    base[..., 12:40] = 1.0
    if kind == "repeat_channels":
        mask = base.repeat(1, 4, 1, 1)
    elif kind == "channel_varying":
        mask = base.repeat(1, 4, 1, 1)
        mask[:, 1] = 0.0
        mask[:, 2, :, :20] = 1.0
        mask[:, 3] = 0.25                     # fractional masks blend too
    else:
        mask = base[0, 0]                     # (H, W): broadcast over batch and channels
    unet.set_compute_dtype("fp32")
    s = _sampler(K, H, B)
    got, _ = s.inpaint_sample(unet, (B, 4, H, W), 1.0, guide.cuda(), mask.cuda(), return_tensor=True, condition=cond.cuda(),
                              sampler="ddpm", seed=17)
    r = RefSampler(1000, height=H, max_batchsize=B)
    r.respace(list(np.linspace(0, 999, K, dtype=np.int32)))
    want, _ = r.inpaint_sample(RefUnet(unet_sd), (B, 4, H, W), 1.0, guide, mask, condition=cond, sampler="ddpm", seed=17)
    errs = [rel_err(a.cpu(), b) for a, b in zip(got, want)]
    print(f"inpaint mask {kind}: per-step rel err {['%.1e' % e for e in errs]}")
    assert len(got) == len(want) and max(errs) < 1e-3


# ----------------------------------------------------------------------------------------------- ADVICE r01 (medium)
def test_shard_contract_philox_and_large_max_batchsize():
    """shard=(rank, world): rank r's noise == rows [r*B, (r+1)*B) of what ONE process with max_batchsize*world draws,
    for the Philox generator too (ranks must differ) and with max_batchsize > batch."""
    from diffusynth_amd.sampler import DiffSynthSampler
    H, B, world = 16, 2, 2
    for nd, mb in (("philox", 2), ("philox", 3), ("cpu", 3)):
        g = DiffSynthSampler(1000, mute=True, device="cuda", height=H, max_batchsize=mb * world, noise_device=nd)
        g._seed(77)
        full = [g.get_deterministic_noise_tensor(B * world, 100)[0] for _ in range(2)]      # two consecutive draws
        locs = []
        for rank in range(world):
            s = DiffSynthSampler(1000, mute=True, device="cuda", height=H, max_batchsize=mb, noise_device=nd, shard=(rank, world))
            s._seed(77)
            loc = [s.get_deterministic_noise_tensor(B, 100)[0] for _ in range(2)]
            for a, b in zip(loc, full):
                assert torch.equal(a, b[rank * B:(rank + 1) * B]), (nd, mb, rank)
            locs.append(loc[0])
        assert not torch.equal(locs[0], locs[1])
        assert abs(float(full[0].mean())) < 0.05 and abs(float(full[0].std()) - 1.0) < 0.05


def test_sharded_sampling_with_default_style_max_batchsize(unet):
    """Two shards of a batch of 4 with max_batchsize 3 > local batch 2 == the unsharded run (max_batchsize 6), bit for bit."""
    unet.set_compute_dtype("fp32")
    cond = synth_input("shard2_c", (4, 512)).cuda()
    ref, _ = _sampler(3, 32, 6).sample(unet, (4, 4, 32, 64), return_tensor=True, condition=cond, sampler="ddpm", seed=5)
    for rank in (0, 1):
        got, _ = _sampler(3, 32, 3, shard=(rank, 2)).sample(unet, (2, 4, 32, 64), return_tensor=True,
                                                           condition=cond[2 * rank:2 * rank + 2], sampler="ddpm", seed=5)
        assert torch.equal(got[-1], ref[-1][2 * rank:2 * rank + 2])


# ----------------------------------------------------------------------------------------------- BASELINE configs[2]
def test_headline_batch64_cfg_properties_bf16(unet):
    """configs[2] at full size (batch 64, CFG => U-Net batch 128, 256x64, bf16): finite; guidance is the identity when
    uncond == cond; a sample's result inside the CFG-doubled batch of 128 equals its result in a batch of 64 (no
    tiling decision may depend on B once the chip is full), and both batch halves of cat([x, x]) agree bit for bit."""
    unet.set_compute_dtype("bf16")
    try:
        B, H, W = 64, 256, 64
        cond1 = synth_input("h64_c", (512,)).cuda()
        cond = cond1.unsqueeze(0).repeat(B, 1)
        # (3 steps: a 2-step DDPM schedule is degenerate — sqrt(1 - a_prev - sigma^2) of its first step is sqrt(~ -1e-12))
        s = _sampler(3, H, B, noise_device="philox")
        a, _ = s.sample(unet, (B, 4, H, W), return_tensor=True, condition=cond, sampler="ddpm", seed=3)
        s = _sampler(3, H, B, noise_device="philox")
        s.activate_classifier_free_guidance(6.0, cond1)           # uncond == cond  =>  eps_u + 6 (eps_c - eps_u) == eps_u
        b, _ = s.sample(unet, (B, 4, H, W), return_tensor=True, condition=cond, sampler="ddpm", seed=3)
        assert torch.isfinite(a[-1]).all() and torch.isfinite(b[-1]).all()
        e_cfg = rel_err(b[-1], a[-1])
        x = a[1]
        t = torch.full((B,), 999, device="cuda", dtype=torch.long)
        y64 = unet(x, t, cond)
        y128 = unet(torch.cat([x, x]), torch.cat([t, t]), torch.cat([cond, cond]))
        e_b = rel_err(y128[:B], y64)
        print(f"B=64 CFG identity rel err {e_cfg:.2e}; sample in batch 128 vs 64 rel err {e_b:.2e}")
        assert e_cfg < 2e-2 and e_b < 2e-2
        assert torch.equal(y128[:B], y128[B:])
        assert not torch.equal(a[-1][0], a[-1][1])               # same condition, different noise
    finally:
        unet.set_compute_dtype("fp32")


def test_headline_batch64_cfg_properties_bf16x3(unet):
    """The bench headline's tier at its full size (configs[2]: batch 64, CFG => U-Net batch 128, 256x64, split-precision bf16x3): finite;
    guidance is the identity when uncond == cond; both halves of cat([x, x]) agree bit for bit; a sample's result inside the batch of 128
    equals its result in a batch of 64; and samples 0-1 of that batch agree with the all-fp32 tier to < 1e-4 (the tier's tiling / split
    decisions at U-Net batch 128 were exercised by the bench alone before)."""
    B, H, W = 64, 256, 64
    cond1 = synth_input("h64_c", (512,)).cuda()
    cond = cond1.unsqueeze(0).repeat(B, 1)
    unet.set_compute_dtype("bf16x3")
    try:
        s = _sampler(3, H, B, noise_device="philox")
        a, _ = s.sample(unet, (B, 4, H, W), return_tensor=True, condition=cond, sampler="ddpm", seed=3)
        s = _sampler(3, H, B, noise_device="philox")
        s.activate_classifier_free_guidance(6.0, cond1)           # uncond == cond  =>  eps_u + 6 (eps_c - eps_u) == eps_u
        b, _ = s.sample(unet, (B, 4, H, W), return_tensor=True, condition=cond, sampler="ddpm", seed=3)
        assert torch.isfinite(a[-1]).all() and torch.isfinite(b[-1]).all()
        e_cfg = rel_err(b[-1], a[-1])
        x = a[1].clone()
        t = torch.full((B,), 999, device="cuda", dtype=torch.long)
        y64 = unet(x, t, cond).clone()
        y128 = unet(torch.cat([x, x]), torch.cat([t, t]), torch.cat([cond, cond])).clone()
        e_b = rel_err(y128[:B], y64)
        assert torch.isfinite(y128).all()
        assert torch.equal(y128[:B], y128[B:])
        assert not torch.equal(a[-1][0], a[-1][1])               # same condition, different noise
    finally:
        unet.set_compute_dtype("fp32")
    ref = unet(x[:2], t[:2], cond[:2])
    e_ref = rel_err(y128[:2], ref)
    print(f"bf16x3 B=64 CFG identity rel err {e_cfg:.2e}; sample in batch 128 vs 64 rel err {e_b:.2e} "
          f"(bit-equal: {torch.equal(y128[:B], y64)}); samples 0-1 of the batch of 128 vs the fp32 tier {e_ref:.2e}")
    assert e_cfg < 1e-4 and e_b < 1e-5 and e_ref < 1e-4


def test_reference_latent_size_batch64_cfg_bf16x3(unet):
    """BASELINE configs[4]'s sampling half at full size in the headline tier: batch 64, CFG => U-Net batch 128 at the reference's own 128 x 64
    latents (text2sound.py:84).  At this size the deepest level's images are 16 x 8 and its 3x3 convolutions take TWO samples per 8 x 32 tile
    (conv3x3_halo3<3, HP, PAIR>: r05).  Finite; the halves of cat([x, x]) agree bit for bit; samples 0-3 (two blocks of sample pairs) and
    the LAST two of the batch agree with the all-fp32 tier (one sample per tile, no split-K) at the tier's error level."""
    B, H, W = 64, 128, 64
    cond = synth_input("r64_c", (512,)).cuda().unsqueeze(0).repeat(B, 1)
    x = synth_input("r64_x", (B, 4, H, W)).cuda()
    t = (torch.arange(B, device="cuda") * 13) % 1000
    unet.set_compute_dtype("bf16x3")
    try:
        y128 = unet(torch.cat([x, x]), torch.cat([t, t]), torch.cat([cond, cond]), paired_halves=True).clone()
        assert torch.isfinite(y128).all()
        assert torch.equal(y128[:B], y128[B:])
        y3 = unet(x[:3], t[:3], cond[:3]).clone()                 # an odd batch: the last block of every paired launch is half empty
    finally:
        unet.set_compute_dtype("fp32")
    idx = [0, 1, 2, 3, B - 2, B - 1]
    ref = unet(x[idx], t[idx], cond[idx])
    e = rel_err(y128[idx], ref)
    e3 = rel_err(y3, ref[:3])
    print(f"bf16x3 at 128x64, U-Net batch 128 (two samples per tile at 16x8): samples {idx} vs the fp32 tier {e:.2e}; batch of 3: {e3:.2e}")
    assert e < 1e-4 and e3 < 1e-4


# ----------------------------------------------------------------------------------------------- BASELINE configs[4]
def test_config5_chain_batch64_latents_to_audio(unet, vae, vqgan_sd):
    """text2sound.py:112-134 at full size: sample() (batch 64, CFG, bf16, reference-native 128x64 latents) -> VQ ->
    decoder -> ISTFT+ / iSTFT -> (64, 65280) audio.  Finite; the first two samples' audio equals the CPU oracle's from
    the IDENTICAL quantised latents at 1e-3 (fp32 tail); analysing and re-synthesising the audio is idempotent."""
    from diffusynth_amd.vocoder import audio_to_stft_representation, latents_to_audio, stft_representation_to_audio
    from oracle import vocoder_ref as V
    from oracle import vqgan_ref as Q
    B, H, W = 64, 128, 64
    unet.set_compute_dtype("bf16")
    try:
        cond1 = synth_input("c5_c", (512,)).cuda()
        s = _sampler(2, H, B, noise_device="philox")
        s.activate_classifier_free_guidance(6.0, synth_input("c5_u", (512,)).cuda())
        lat, _ = s.sample(unet, (B, 4, H, W), return_tensor=True, condition=cond1.unsqueeze(0).repeat(B, 1), sampler="ddim", seed=9)
    finally:
        unet.set_compute_dtype("fp32")
    z = lat[-1]
    z = z / z.std() * 1.2                       # random-init U-Net output has no trained scale: bring it into the codebook's range
    q, _, _ = vae._vq_vae(z)
    assert q.shape == (B, 4, H, W) and len(torch.unique(vae._vq_vae.last_indices)) > 1000
    vae._decoder.set_compute_dtype("fp32")
    audio = latents_to_audio(vae._decoder, q)
    assert audio.shape == (B, 256 * (4 * W - 1)) == (64, 65280) and torch.isfinite(audio).all()
    dec_ref = Q.decoder_forward(vqgan_sd, Q.PRODUCTION_CONFIG, q[:2].cpu())
    ref = np.stack(V.latents_to_audio(dec_ref.numpy()))
    err = rel_err(audio[:2].cpu(), ref)
    print(f"config-5 chain: audio of samples 0-1 vs oracle rel err {err:.2e}")
    assert err < 1e-3
    r1 = stft_representation_to_audio(audio_to_stft_representation(audio, time_resolution=4 * W)[:, :, :, :4 * W].contiguous())
    r2 = stft_representation_to_audio(audio_to_stft_representation(r1, time_resolution=4 * W)[:, :, :, :4 * W].contiguous())
    assert r1.shape == audio.shape
    # (not exactly idempotent: the representation drops each FRAME's DC bin, and overlap-adding DC-free frames does not
    # give frames that are DC-free again; measured 1.8e-3 in the max norm, 6.9e-3 rms-relative — a property of the representation,
    # not a parity bound: the parity bound of this test is the 1e-3 against the oracle above, 2.6e-6 measured)
    assert rel_err(r2[:, 1024:-1024], r1[:, 1024:-1024]) < 1.5e-2
    # the bf16 decoder (throughput tier of the tail) stays within its reported tolerance at this size
    vae._decoder.set_compute_dtype("bf16")
    ab = latents_to_audio(vae._decoder, q[:8])
    vae._decoder.set_compute_dtype("fp32")
    assert torch.isfinite(ab).all() and rel_err(ab, audio[:8]) < 1e-1


# ----------------------------------------------------------------------------------------------- SURVEY §8e rehearsal
def test_two_rank_rehearsal_on_one_gpu():
    """Two fresh ranks on cuda:0 over gloo: broadcast of the embeddings, sharded sampling, gather == the unsharded
    run bit for bit (fp32 parity tier, CPU-generator noise), then the bench's own sharded Philox path for two steps."""
    import torch.multiprocessing as mp

    from dist_gpu_worker import worker
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    print(res)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res
    assert all(p.exitcode == 0 for p in procs)


# ----------------------------------------------------------------------------------------------- SURVEY §8f row 4 (serving)
def test_plan_cache_is_bounded_and_shares_one_arena(unet):
    """Variable-width serving (text2sound.py:84): every new (B, H, W, cond) shape used to keep its own arena forever.  The
    engine now keeps at most DS_MAX_PLANS plans in ONE arena; evicted / rebuilt plans give bit-identical results."""
    unet.set_compute_dtype("bf16")
    try:
        t = torch.tensor([321]).cuda()
        c = synth_input("pc_c", (1, 512)).cuda()
        first = {}
        widths = [20, 24, 27, 32, 40, 48, 56, 64, 72, 80, 100]
        for w in widths:
            x = synth_input("pc_x%d" % w, (1, 4, 32, w)).cuda()
            first[w] = unet(x, t, c)
        eng = unet._engine
        assert len(eng.plans) <= eng._max_plans == 8
        arenas = {id(p.ws) for p in eng.plans.values()}
        assert arenas == {id(eng._arena)}                       # every cached plan lives in the one arena
        big = unet(synth_input("pc_big", (4, 4, 64, 64)).cuda(), t.repeat(4), c.repeat(4, 1))      # grows the arena, drops the plans
        assert torch.isfinite(big).all() and len(eng.plans) == 1
        for w in (20, 64, 100):                                  # rebuilt after eviction / arena growth: same bits
            x = synth_input("pc_x%d" % w, (1, 4, 32, w)).cuda()
            assert torch.equal(unet(x, t, c), first[w])
    finally:
        unet.set_compute_dtype("fp32")


def test_mixed_width_requests_match_per_sample_oracle(unet, unet_sd):
    """One serving call over notes of widths {20, 64, 100} (track_maker.py:245): requests are bucketed by width, and every
    request's result equals the CPU oracle's single-sample run with that request's seed (fp32 tier, DDIM)."""
    from diffusynth_amd.serving import sample_mixed_widths
    from oracle.sampler_ref import RefSampler
    from oracle.unet_ref import RefUnet
    unet.set_compute_dtype("fp32")
    H, K = 32, 3
    reqs = [{"width": w, "condition": synth_input("mw_c%d" % i, (512,)), "seed": 100 + i} for i, w in enumerate([20, 64, 100, 64, 20])]
    got = sample_mixed_widths(unet, reqs, K, height=H, noise_device="cpu")
    ref_model = RefUnet(unet_sd)
    for r, g in zip(reqs, got):
        s = RefSampler(1000, height=H, max_batchsize=1)
        s.respace(list(np.linspace(0, 999, K, dtype=np.int32)))
        want, _ = s.sample(ref_model, (1, 4, H, r["width"]), condition=r["condition"][None], sampler="ddim", seed=r["seed"])
        assert g.shape == (4, H, r["width"])
        assert rel_err(g.cpu(), want[-1][0]) < 1e-3
    assert len(unet._engine.plans) <= 8
