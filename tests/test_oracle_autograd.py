"""Independent check of the oracle's ANALYTIC backward: an fp64 PyTorch-autograd model of the
render equation, written from the equations (EWA projection, SH colour, front-to-back blending),
not from the reference's backward code.  It shares no code with oracle/cugs_oracle.c.

Scene chosen so that the reference's non-differentiable quirks are inactive: no pixel saturates
(so Q1's count-from-the-end walk equals the true contributor set) and no alpha hits the 0.99
clamp.  SH directions are constants in the reference (Q4), so they are detached here too."""
import numpy as np
import torch

from util import oracle_backward, oracle_forward

SH_C = [0.28209479177387814, 0.4886025119029199,
        [1.0925484305920792, 1.0925484305920792, 0.31539156525252005, 1.0925484305920792, 0.5462742152960396],
        [0.5900435899266435, 2.890611442640554, 0.4570457994644658, 0.3731763325901154, 0.4570457994644658,
         1.4453057213202769, 0.5900435899266435]]


def _sh_basis(d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz = x * x, y * y, z * z
    c2, c3 = SH_C[2], SH_C[3]
    return torch.stack([torch.full_like(x, SH_C[0]), -SH_C[1] * y, SH_C[1] * z, -SH_C[1] * x,
                        c2[0] * x * y, c2[1] * y * z, c2[2] * (2 * zz - xx - yy), c2[3] * x * z, c2[4] * (xx - yy),
                        c3[0] * y * (3 * xx - yy), c3[1] * x * y * z, c3[2] * y * (4 * zz - xx - yy),
                        c3[3] * z * (2 * zz - 3 * xx - 3 * yy), c3[4] * x * (4 * zz - xx - yy),
                        c3[5] * z * (xx - yy), c3[6] * x * (xx - 3 * yy)], dim=1)


def _render_autograd(p, cam, tile_member, order, bg):
    """p: dict of fp64 leaf tensors.  tile_member[pixel, gaussian] and the depth order come from the
    oracle's forward (they are piecewise constant in the parameters)."""
    W = torch.tensor(cam.rotation, dtype=torch.float64)
    tvec = torch.tensor(cam.translation, dtype=torch.float64)
    K = cam.intrinsics
    pos = p["positions"]
    t = pos @ W.T + tvec
    mx = K.fx * t[:, 0] / t[:, 2] + K.cx
    my = K.fy * t[:, 1] / t[:, 2] + K.cy
    q = p["rotations"] / torch.sqrt((p["rotations"] ** 2).sum(1, keepdim=True) + 1e-12)
    w_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w_ * z), 2 * (x * z + w_ * y),
                     2 * (x * y + w_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w_ * x),
                     2 * (x * z - w_ * y), 2 * (y * z + w_ * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = R * torch.exp(p["scales"])[:, None, :]
    Sigma = M @ M.transpose(1, 2)
    tz = t[:, 2] + 1e-6
    zero = torch.zeros_like(tz)
    J = torch.stack([K.fx / tz, zero, -K.fx * t[:, 0] / tz ** 2, zero, K.fy / tz, -K.fy * t[:, 1] / tz ** 2],
                    1).reshape(-1, 2, 3)
    T = J @ W
    cov = T @ Sigma @ T.transpose(1, 2) + 0.3 * torch.eye(2, dtype=torch.float64)
    inv = torch.linalg.inv(cov)
    a, b, c = inv[:, 0, 0], inv[:, 0, 1], inv[:, 1, 1]
    o = torch.sigmoid(p["opacities"][:, 0])
    cc = -W.T @ tvec
    d = (pos - cc).detach()
    d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-8)
    rgb = (p["sh_coeffs"] * _sh_basis(d)[:, None, :]).sum(2) + 0.5
    rgb = torch.where(rgb > 0, rgb, torch.zeros_like(rgb))

    H, Wd = cam.height, cam.width
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64) + 0.5, torch.arange(Wd, dtype=torch.float64) + 0.5,
                            indexing="ij")
    px, py = xs.reshape(-1, 1), ys.reshape(-1, 1)
    idx = torch.as_tensor(order)
    dx, dy = px - mx[idx][None], py - my[idx][None]
    power = -0.5 * (a[idx] * dx * dx + 2 * b[idx] * dx * dy + c[idx] * dy * dy)
    alpha_raw = o[idx][None] * torch.exp(power)
    assert float(alpha_raw.max()) < 0.99, "scene must not reach the alpha clamp"
    keep = torch.as_tensor(tile_member[:, order]) & (power <= 0) & (alpha_raw >= 1.0 / 255.0)
    alpha = torch.where(keep, alpha_raw, torch.zeros_like(alpha_raw))
    Tcum = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha], 1), 1)
    assert float(Tcum[:, -1].min()) > 1.0 / 255.0 * 1.5, "scene must not saturate"
    color = ((alpha * Tcum[:, :-1])[:, :, None] * rgb[idx][None]).sum(1) + Tcum[:, -1:] * torch.tensor(bg, dtype=torch.float64)
    return color.reshape(H, Wd, 3)


def test_oracle_backward_matches_fp64_autograd(pkg, orc):
    w, h, n = 48, 32, 24
    cam = pkg.scene.make_camera(w, h, view=2)
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=3, seed=99, mu_s=-2.2)
    arrays["opacities"] = (arrays["opacities"] * 0.5 - 1.6).astype(np.float32)       # opacity ~0.17: no saturation
    bg = (0.3, 0.1, 0.2)
    fwd = oracle_forward(orc, arrays, cam, bg=bg)
    assert fwd["final_T"].min() > 1.5 / 255.0 and (fwd["radii"] > 0).sum() >= n // 2
    g = np.random.default_rng(3).standard_normal((h, w, 3)).astype(np.float32)
    ana = oracle_backward(orc, g, fwd, arrays, cam, bg=bg)

    # tile membership and global depth order from the oracle's forward
    ntx = (w + 15) // 16
    member = np.zeros((h * w, n), bool)
    tr, vals = fwd["tile_ranges"], fwd["values"]
    for py in range(h):
        for px in range(w):
            s, e = tr[(py // 16) * ntx + px // 16]
            member[py * w + px, vals[s:e]] = True
    order = np.argsort(fwd["depths"], kind="stable")
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in arrays.items()}
    color = _render_autograd(p, cam, member, order, bg)
    assert np.allclose(color.detach().numpy(), fwd["color"], atol=2e-5)               # forward agrees first
    (color * torch.tensor(g, dtype=torch.float64)).sum().backward()
    for name in ("positions", "rotations", "scales", "opacities", "sh_coeffs"):
        want = p[name].grad.numpy()
        got = ana["dL_d" + name].reshape(want.shape)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-4 * scale, (name, np.abs(got - want).max() / scale)
