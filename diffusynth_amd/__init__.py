"""diffusynth_amd — MI355X-native (gfx950) implementation of diffusynth's text-conditioned
denoising sampling path: DiffSynthSampler.sample() -> ConditionedUnet.forward -> VQ -> VQGAN
decoder -> ISTFT+/iSTFT, as hand-written HIP kernels behind a C-ABI library
(include/diffusynth_hip.h), with Python host code that mirrors the reference's signatures.

Importing the package is cheap and works without a GPU; anything that computes loads
``libdiffusynth_hip.so`` and raises if it is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
