"""Curvature profile of the decoded circle (regular_cnn/geometry.py:5-64), all angles at once.

The reference walks 360 angles one at a time and differentiates the decoder twice per angle with reverse-mode
jacobians (720 autograd graphs).  The curve gamma(t) = decode([cos t, sin t]) is a function of ONE scalar, so its
velocity and acceleration are two nested forward-mode derivatives, and the decoder treats rows independently, so all
angles go through as one batch: three decoder evaluations' worth of work in total.  Plain PyTorch (torch.func) on
whatever device the model lives on; no kernels of this package are involved.
"""
import numpy as np
import torch
from torch.func import jvp


def compute_curvature_profile(model, device, num_points=360):
    """Same signature and return value as the reference: (thetas, |H(theta)|) as numpy arrays, with
    H = a_normal / |v|^2, a_normal = a - (<a, v> / |v|^2) v, and 0 where |v|^2 < 1e-6 (:45-47)."""
    model.eval()
    thetas = torch.linspace(0, 2 * np.pi, num_points).to(device)          # :13
    params = next(model.parameters(), None)
    t = thetas if params is None else thetas.to(params.dtype)

    def gamma(tt):                                                         # :21-25, every angle a row
        z_input = torch.stack([torch.cos(tt), torch.sin(tt)], dim=-1)
        return model.decode(z_input).reshape(tt.shape[0], -1)

    def velocity(tt):                                                      # d gamma / d t
        return jvp(gamma, (tt,), (torch.ones_like(tt),))[1]

    with torch.no_grad():
        v, a = jvp(velocity, (t,), (torch.ones_like(t),))                  # velocity and acceleration, (N, D)
        g = (v * v).sum(-1)                                                # pullback metric :40
        safe = torch.where(g < 1e-6, torch.ones_like(g), g)
        a_normal = a - ((a * v).sum(-1) / safe).unsqueeze(-1) * v          # :51-55
        h_norm = torch.linalg.vector_norm(a_normal / safe.unsqueeze(-1), dim=-1)   # :58-61
        h_norm = torch.where(g < 1e-6, torch.zeros_like(h_norm), h_norm)
    return thetas.detach().cpu().numpy(), h_norm.double().cpu().numpy()
