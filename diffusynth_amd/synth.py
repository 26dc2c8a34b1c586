"""Deterministic synthetic weights / inputs.

There are no checkpoints offline, so parity tests, the golden-vector generator and
bench.py all fill a reference-format state dict from the parameter *names and
shapes* alone: every tensor is drawn from numpy's PCG64 seeded by crc32(key), so the
build container and the GPU box regenerate bit-identical weights without shipping
428 MB.  (numpy Generator streams are platform independent.)
"""
import zlib

import numpy as np
import torch


def _rng(tag):
    return np.random.Generator(np.random.PCG64(zlib.crc32(tag.encode("utf-8"))))


def synth_tensor(key, shape):
    """One parameter tensor (fp32) for state-dict entry ``key`` of shape ``shape``."""
    shape = tuple(int(s) for s in shape)
    g = _rng(key)
    n = g.standard_normal(shape, dtype=np.float64)
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("_embedding.weight") and "_vq_vae" in key:      # VQ codebook ~ N(0,1) (VQGAN.py:88)
        arr = n
    elif leaf in ("_ema_cluster_size",):
        arr = np.abs(n)
    elif leaf == "running_var":                                        # BatchNorm2d buffers (VQGAN norm_type="batchnorm")
        arr = 0.5 + np.abs(n)
    elif leaf == "running_mean":
        arr = 0.3 * n
    elif leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    elif leaf == "bias":
        arr = 0.1 * n
    elif len(shape) == 1:                                              # norm gains
        arr = 1.0 + 0.2 * n
    else:
        fan_in = int(np.prod(shape[1:]))
        arr = n / np.sqrt(max(fan_in, 1))
    return torch.from_numpy(arr.astype(np.float32))


def synth_state_dict(spec):
    """spec: iterable of (key, shape) -> {key: fp32 tensor}."""
    return {k: synth_tensor(k, s) for k, s in spec}


def synth_input(tag, shape, scale=1.0):
    """Seeded N(0, scale^2) fp32 tensor for test inputs / synthetic text embeddings."""
    g = _rng("input:" + tag)
    return torch.from_numpy((scale * g.standard_normal(tuple(shape))).astype(np.float32))
