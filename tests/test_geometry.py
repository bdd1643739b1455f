"""Batched curvature profile (medicalimageanalysis_amd/geometry.py) against the reference's sequential double-jacobian
loop: golden values produced by executing regular_cnn/geometry.py:5-64, and a restatement of that loop run here."""
import numpy as np
import pytest
import torch

from medicalimageanalysis_amd.geometry import compute_curvature_profile
from medicalimageanalysis_amd.model.s1_vae import S1_VAE


def _model_from(g, tag, dtype):
    model = S1_VAE(input_dim=7, hidden_dim=12).to(dtype)
    sd = {k.split("::", 1)[1]: torch.as_tensor(g[k]) for k in g.files if k.startswith(f"w_{tag}::")}
    model.load_state_dict(sd)
    return model


def test_curvature_profile_matches_reference_execution(golden):
    g = golden("g_curvature")
    model = _model_from(g, "f32", torch.float32)
    thetas, prof = compute_curvature_profile(model, "cpu", num_points=45)
    np.testing.assert_array_equal(thetas, g["thetas_f32"])
    np.testing.assert_allclose(prof, g["profile_f32"], rtol=2e-4, atol=1e-6)


def test_curvature_profile_matches_sequential_loop_in_double(golden):
    """the reference's algorithm (one angle at a time, reverse-mode jacobian of a jacobian) restated for a double
    model, where the two differentiation orders must agree to rounding"""
    g = golden("g_curvature")
    model = _model_from(g, "f32", torch.float64)
    _, prof = compute_curvature_profile(model, "cpu", num_points=24)
    thetas = torch.linspace(0, 2 * np.pi, 24).double()
    ref = []
    for theta in thetas:
        tv = theta.unsqueeze(0)

        def dec(t):
            return model.decode(torch.stack([torch.cos(t), torch.sin(t)], dim=-1)).squeeze()

        def vel(t):
            return torch.autograd.functional.jacobian(dec, t, create_graph=True).squeeze()
        v = vel(tv)
        a = torch.autograd.functional.jacobian(vel, tv).squeeze()
        gg = torch.dot(v, v)
        an = a - torch.dot(a, v) / gg * v
        ref.append(float(torch.norm(an / gg).detach()))
    np.testing.assert_allclose(prof, np.array(ref), rtol=1e-9, atol=1e-12)


def test_curvature_of_a_flat_decoder_is_zero():
    class Flat(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def decode(self, z):
            return z[..., :1] * 0.0 + self.w          # constant curve: |v|^2 < 1e-6 everywhere

    _, prof = compute_curvature_profile(Flat(), "cpu", num_points=16)
    assert prof.shape == (16,) and not prof.any()


@pytest.mark.gpu
def test_curvature_profile_on_the_device(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    g = golden("g_curvature")
    model = _model_from(g, "f32", torch.float32).to("cuda")
    _, prof = compute_curvature_profile(model, "cuda", num_points=45)
    np.testing.assert_allclose(prof, g["profile_f32"], rtol=5e-4, atol=1e-5)
