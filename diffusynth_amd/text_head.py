"""Text-condition head on device - drop-in for ProjectionLayer / ProjectionHead of
model/multimodal_model.py:14-47, which turns a CLAP text feature into the 512-d `condition` of the sampler
(multimodal_model.py:114-116, app.py:59).  Same constructor arguments and state-dict names
(``layers.N.{projection,fc,layer_norm}.{weight,bias}``); inference only (dropout is the identity in eval mode).
Per layer: two ds_linear launches (the second applies GELU to its input) and one ds_add_layernorm.  The CLAP text
tower itself needs remote weights and stays outside."""
import torch
from torch import nn

from . import _lib as L

GELU_IN = 1     # ds_linear act_in code (include/diffusynth_hip.h)


class ProjectionLayer(nn.Module):
    def __init__(self, input_dim, output_dim, dropout):
        super().__init__()
        self.projection = nn.Linear(input_dim, output_dim)
        self.gelu = nn.GELU()
        self.fc = nn.Linear(output_dim, output_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(output_dim)
        self.eval()

    @torch.no_grad()
    def forward(self, x):
        if self.training:
            raise RuntimeError("diffusynth_amd.ProjectionLayer is inference-only (dropout)")
        if not x.is_cuda:
            raise RuntimeError("diffusynth_amd text head runs on MI355X only (ds_linear / ds_add_layernorm); no CPU fallback")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        B, K = x2.shape
        D = self.projection.out_features
        st = L.current_stream()
        w = lambda t: t.detach().float().contiguous()
        projected = torch.empty(B, D, device=x.device)
        hidden = torch.empty(B, D, device=x.device)
        out = torch.empty(B, D, device=x.device)
        pw, pb, fw, fb = w(self.projection.weight), w(self.projection.bias), w(self.fc.weight), w(self.fc.bias)
        g, be = w(self.layer_norm.weight), w(self.layer_norm.bias)
        L.call("ds_linear", x2.data_ptr(), K, pw.data_ptr(), pb.data_ptr(), B, K, D, 0, projected.data_ptr(), D, st)
        L.call("ds_linear", projected.data_ptr(), D, fw.data_ptr(), fb.data_ptr(), B, D, D, GELU_IN, hidden.data_ptr(), D, st)
        L.call("ds_add_layernorm", hidden.data_ptr(), projected.data_ptr(), g.data_ptr(), be.data_ptr(), B, D,
               float(self.layer_norm.eps), out.data_ptr(), st)
        return out.reshape(*lead, D)


class ProjectionHead(nn.Module):
    def __init__(self, embedding_dim, projection_dim, dropout, num_layers=2):
        super().__init__()
        self.layers = nn.ModuleList([ProjectionLayer(embedding_dim if i == 0 else projection_dim, projection_dim, dropout)
                                     for i in range(num_layers)])
        self.eval()

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
