"""Oracle (test infrastructure): CPU restatement of the text-condition ProjectionHead
(model/multimodal_model.py:14-47; applied to CLAP text features at :114-116).  Eval mode: dropout is the identity.
Driven by a reference-format state dict (keys ``<prefix>.layers.N.{projection,fc,layer_norm}.*``).  Pinned by
tests/golden/head.npz (outputs of the imported reference, tools/gen_golden.py gen_head)."""
import torch
import torch.nn.functional as F


def projection_layer(sd, p, x):
    """multimodal_model.py:25-32."""
    projected = F.linear(x, sd[p + ".projection.weight"], sd[p + ".projection.bias"])
    h = F.linear(F.gelu(projected), sd[p + ".fc.weight"], sd[p + ".fc.bias"])
    h = h + projected
    return F.layer_norm(h, (h.shape[-1],), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5)


@torch.no_grad()
def projection_head(sd, prefix, x):
    """multimodal_model.py:44-47."""
    i = 0
    while f"{prefix}.layers.{i}.projection.weight" in sd:
        x = projection_layer(sd, f"{prefix}.layers.{i}", x)
        i += 1
    return x
