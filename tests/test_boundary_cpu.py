"""CPU suite for the drop-in boundary: state-dict compatibility, C-ABI symbols, host-side sampler logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_keys, load_golden
from diffusynth_amd import _lib as L


def test_state_dict_names_and_shapes_match_reference():
    from diffusynth_amd.unet import ConditionedUnet, PRODUCTION_CONFIG, UNet
    assert UNet is ConditionedUnet
    for name, cfg in (("unet_production", PRODUCTION_CONFIG), ("unet_resnet", dict(PRODUCTION_CONFIG, use_convnext=False)),
                      ("unet_small_cat", dict(in_dim=4, down_dims=[32, 32, 64], up_dims=[64, 64, 32], attn_type="linear_cat",
                                              condition_type="natural_language_prompt", label_emb_dim=64))):
        m = ConditionedUnet(**cfg)
        got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        assert got == golden_keys(name), name
    m = ConditionedUnet(**PRODUCTION_CONFIG)
    assert sum(p.numel() for p in m.parameters()) == 106948900
    with pytest.raises(NotImplementedError):
        ConditionedUnet(4, attn_type="softmax")
    with pytest.raises(NotImplementedError):
        ConditionedUnet(4, condition_type="other")
    with pytest.raises(AssertionError):
        ConditionedUnet(4, down_dims=[32, 64], up_dims=[64, 32, 16])


def test_forward_fails_loudly_without_gpu():
    from diffusynth_amd.unet import ConditionedUnet
    m = ConditionedUnet(4, down_dims=[32, 32], up_dims=[32, 32])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, 8, 8), torch.zeros(1, dtype=torch.long))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a CPU-only box and exports exactly the header's entry points."""
    lib = L.load()
    with open(os.path.join(ROOT, "include", "diffusynth_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", text))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ds_abi_version() == 1
    assert lib.ds_conv_tile_bn(L.TILE_128x192) == 192 and lib.ds_conv_tile_bn(L.TILE_256x96) == 96
    # argument validation happens before any GPU work: callable without a device
    p = L.ConvParams()
    assert lib.ds_conv_igemm(ctypes.byref(p), None) == -1
    assert b"conv_igemm" in lib.ds_last_error_string()


def test_host_side_tiling_policies():
    """The batch-dependent tiling decisions of the bf16 tier live behind host-only C entry points (no device work): the attention segment
    count follows the kernel generation chosen for the batch, the depthwise kernel's GroupNorm partials follow its chunking, and both
    validate before touching a GPU."""
    lib = L.load()
    seg = lib.ds_attn_fused_segments
    # first generation below 96 samples: N / 128 segments, at most 32, at least 1
    assert [seg(2, n, 96) for n in (48, 256, 4096, 16384)] == [1, 2, 32, 32]
    assert seg(64, 16384, 96) == 32 and seg(16, 4096, 192) == 32
    # second generation from 96 samples: one round of blocks = 2048 / B segments (1024 / B at C = 384, images of >= 1024 pixels only)
    assert seg(128, 16384, 96) == 16 and seg(96, 16384, 96) == 21 and seg(128, 4096, 192) == 16
    assert seg(128, 1024, 384) == 8 and seg(128, 256, 384) == 2
    assert seg(128, 64, 96) == 2                                  # never more segments than 32-pixel tiles
    # depthwise: partials per sample = chunks per sample x channel blocks; long chunks only while two chunks per CU remain
    def parts(B, H, W, C0, C1=0):
        p = L.DwconvParams(C0=C0, C1=C1, H=H, W=W, H1=H, W1=W, B=B, dtype=L.DS_BF16)
        p.wexp = 1                                                 # (non-null: the MFMA kernel's packed taps are present)
        return lib.ds_dwconv_stats_parts(ctypes.byref(p))
    assert parts(128, 256, 64, 96) == 4 * 3                        # 32 tiles in chunks of 8, three channel blocks
    assert parts(16, 256, 64, 96) == 16 * 3                        # batch 16: chunks of 2 (768 chunks >= 512)
    assert parts(1, 256, 64, 96) == 32 * 3                         # batch 1: one tile per chunk
    assert parts(128, 64, 16, 384) == 12                           # two tall tiles per image: chunks run across samples, one partial each
    assert parts(128, 256, 64, 96, 192) == 4 * 9                   # the skip concat as two sources

    # r05, split-precision tier (fp32 tensors, plane output): the strip kernel from 512 blocks on — one statistics partial per (row range, strip,
    # 32-channel block); below that, in the fp32 parity tier (no plane output) and for narrow / short images the tile kernel's count
    def parts32(B, H, W, C0, C1=0, split=1, strip=0):
        p = L.DwconvParams(C0=C0, C1=C1, H=H, W=W, H1=H, W1=W, B=B, dtype=L.DS_F32, out_split=split, strip=strip)
        return lib.ds_dwconv_stats_parts(ctypes.byref(p))
    assert parts32(128, 256, 64, 96) == 1 * 4 * 3                  # 1536 strips: no row ranges
    assert parts32(16, 256, 64, 96) == 8 * 4 * 3                   # 192 strips: eight row ranges of 32 rows
    assert parts32(1, 256, 64, 96) == 16 * 4 * 3                   # too few blocks even in row ranges: the tile kernel (16 x 16 tiles)
    assert parts32(128, 128, 100, 96) == 7 * 3                     # ragged width: seven strips
    assert parts32(128, 256, 64, 96, split=0) == 16 * 4 * 3        # fp32 output (the parity tier): never by the library's own choice
    assert parts32(128, 256, 64, 96, split=0, strip=1) == 4 * 3    # ... only when the caller asks
    assert parts32(128, 256, 64, 96, strip=2) == 16 * 4 * 3        # and never when the caller forbids it
    assert parts32(128, 32, 8, 768) == 24                          # 8-wide images: the 8 x 32 tile kernel


def _sampler(**kw):
    from diffusynth_amd.sampler import DiffSynthSampler
    return DiffSynthSampler(1000, mute=True, device="cpu", **kw)


def test_sampler_schedule_respace_matches_reference():
    g = load_golden("schedule")
    s = _sampler()
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                 "sqrt_one_minus_alphas_cumprod", "posterior_variance"):
        np.testing.assert_allclose(getattr(s, name), g["raw_" + name], rtol=0, atol=1e-15)
    for K in (10, 20, 50, 100):
        s = _sampler()
        s.respace(list(np.linspace(0, 999, K, dtype=np.int32)))
        assert s.timestep_map == list(g[f"k{K}_timestep_map"]) and s.num_timesteps == K
        np.testing.assert_allclose(s.alphas_cumprod, g[f"k{K}_alphas_cumprod"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(s.alphas_cumprod_prev, g[f"k{K}_alphas_cumprod_prev"], rtol=0, atol=1e-15)
        with pytest.raises(AssertionError):
            s.define_beta_schedule()


def test_sampler_noise_layout_masks_and_errors():
    g = load_golden("noise_layout")
    s = _sampler(height=2, max_batchsize=1, channels=1)
    ref = torch.arange(64, dtype=torch.float32).reshape(1, 1, 1, 64).repeat(1, 1, 2, 1)
    for W in (20, 48, 64, 65, 100, 144, 256):
        n, pts = s.get_deterministic_noise_tensor(1, W, reference_noise=ref)
        assert list(n[0, 0, 0].long()) == list(g[f"w{W}_cols"]) and pts == list(g[f"w{W}_points"])
    torch.manual_seed(123)
    s2 = _sampler(height=4, max_batchsize=3, channels=2)
    n, _ = s2.get_deterministic_noise_tensor(2, 100)
    assert torch.equal(n, torch.from_numpy(g["seeded_b2_w100"]))
    n, _ = s2.get_deterministic_noise_tensor(1, 40)
    assert torch.equal(n, torch.from_numpy(g["seeded_b1_w40_second_draw"]))
    # sharded draw == slice of the global draw
    torch.manual_seed(123)
    s3 = _sampler(height=4, max_batchsize=1, channels=2, shard=(1, 3))
    n3, _ = s3.get_deterministic_noise_tensor(1, 100)
    assert torch.equal(n3, torch.from_numpy(g["seeded_b2_w100"])[1:2])
    gm = load_golden("masks")
    for W in (64, 100, 144):
        _, pts = s.get_deterministic_noise_tensor(1, W)
        for flex in (0.8, 1.0):
            m = s.get_dynamic_masks(20, (1, 1, 2, W), pts, mask_flexivity=flex)
            assert torch.equal(torch.stack([x[0, 0, 0] for x in m]), torch.from_numpy(gm[f"w{W}_f{int(flex * 10)}"]))
    s4 = _sampler(height=8)
    with pytest.raises(AssertionError, match="shape\\[1\\] != self.channels"):
        s4.p_sample_loop(None, (1, 3, 8, 64))
    with pytest.raises(AssertionError, match="shape\\[2\\] != self.height"):
        s4.p_sample_loop(None, (1, 4, 9, 64))
    with pytest.raises(AssertionError, match="guide_img must be given"):
        s4.p_sample_loop(None, (1, 4, 8, 64), start_noise_level_ratio=0.5)
    with pytest.raises(NotImplementedError):
        s4.p_sample(None, torch.zeros(1, 4, 8, 64), torch.zeros(1, dtype=torch.long), sampler="euler")
    with pytest.raises(AssertionError, match="unconditional_condition must be available"):
        s4.activate_classifier_free_guidance(3.0, None)
    gs = load_golden("step")
    x, nz = torch.from_numpy(gs["x"]), torch.from_numpy(gs["q_noise"])
    assert torch.equal(s4.q_sample(x, torch.full((2,), 500, dtype=torch.long), noise=nz), torch.from_numpy(gs["q_sample_t500"]))


def test_step_coefficients_follow_reference_rounding():
    """The [B][5] table handed to ds_ddim_step reproduces the reference's fp32 op order (checked via the oracle update)."""
    from oracle import sampler_ref as S
    g = load_golden("step")
    x = torch.from_numpy(g["x"])
    s = _sampler(height=8, max_batchsize=3)
    s.respace(list(np.linspace(0, 999, 20, dtype=np.int32)))
    for eta, name in ((0.0, "ddim"), (1.0, "ddpm")):
        for ti in (0, 1, 10, 19):
            t = torch.full((2,), ti, dtype=torch.long)
            c = s._step_coefficients(t, eta)
            eps = 0.1 * x + 0.01 * torch.tensor(s.timestep_map)[t].view(-1, 1, 1, 1).float() + 0.001 * torch.from_numpy(g["cond"]).mean(1).view(-1, 1, 1, 1)
            torch.manual_seed(77)
            nz = torch.randn(3, 4, 8, 64)[:2][..., S.repeat_layout(64, 100)[0]]
            cc = [c[:, i].view(-1, 1, 1, 1) for i in range(5)]
            got = cc[2] * ((x - cc[0] * eps) / cc[1]) + cc[3] * eps + cc[4] * nz
            assert torch.equal(got, torch.from_numpy(g[f"k20_{name}_cfg1_t{ti}"]))


def test_vqgan_state_dict_matches_reference():
    from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN
    m = VQGAN(**PRODUCTION_CONFIG)
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == golden_keys("vqgan_production")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m._decoder(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m._vq_vae(torch.zeros(1, 4, 8, 8))
    # decay == 0 selects the non-EMA VectorQuantizer (VQGAN.py:441-446): its own class, the reference's state-dict keys (a decay = 0
    # checkpoint loads) and its initialisation range (VQGAN.py:38)
    from diffusynth_amd.vqgan import VectorQuantizer, VectorQuantizerEMA
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vq_plain.npz"))
    m0 = VQGAN(**dict(PRODUCTION_CONFIG, decay=0.0))
    assert type(m0._vq_vae) is VectorQuantizer and type(m._vq_vae) is VectorQuantizerEMA
    assert sorted(m0._vq_vae.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert m0._vq_vae._embedding.weight.abs().max().item() <= float(g["init_absmax"])
    assert {k for k in m.state_dict() if k.startswith("_vq_vae")} - {k for k in m0.state_dict() if k.startswith("_vq_vae")} == \
        {"_vq_vae._ema_cluster_size", "_vq_vae._ema_w"}



def test_mixed_width_serving_argument_errors():
    """serving.sample_mixed_widths: the shared-bucket loop is only defined for the deterministic sampler, and CFG needs the
    negative-prompt embedding (both checked before any device work)."""
    from diffusynth_amd.serving import sample_mixed_widths
    reqs = [{"width": 20, "condition": None, "seed": 1}]
    with pytest.raises(NotImplementedError, match="ddim"):
        sample_mixed_widths(None, reqs, 5, sampler="ddpm")
    with pytest.raises(ValueError, match="unconditional_condition"):
        sample_mixed_widths(None, reqs, 5, cfg_scale=3.0)
