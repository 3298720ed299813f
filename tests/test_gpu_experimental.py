"""Parity of the code paths that have not run on hardware yet (written after round 1's GPU access ended):
the 4-lanes-per-Gaussian tiny-splat gather (GSLOC_TINY_GATHER=4), the library variant with interleaved
64-byte records (GSLOC_AOS=1) and the one whose compositing backward is compiled for 5 waves per SIMD
(GSLOC_LIB_VARIANT=occ5).  Each runs in a child process (the switches are read at import) and is compared
with the default path of this process.  Skipped unless GSLOC_EXPERIMENTAL=1."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GSLOC_EXPERIMENTAL") != "1",
                                 reason="set GSLOC_EXPERIMENTAL=1 (paths not yet run on hardware)")]

CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from gsplatloc_amd.context import RenderContext
from gsplatloc_amd.synthetic import perturbed_pose, random_scene
N, W, H, sigma = 60_011, 640, 470, float(sys.argv[3])
dev = torch.device("cuda")
sc = random_scene(N, W, H, sigma_px=sigma, device=dev)
V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, sc["K"].contiguous())
ctx.calibrate(*inp)
g = torch.Generator().manual_seed(3)
v = torch.randn(H, W, 4, generator=g).to(dev)
va = torch.randn(H, W, 1, generator=g).to(dev)
for _ in range(2):
    render, alphas = ctx.forward(*inp)
    grads = ctx.backward(v, va, full=True)
torch.cuda.synchronize()
ctx.check_capacity()
out = {"render": render.cpu(), "alphas": alphas.cpu(), "tiny": ctx.tiny, "stride": ctx.lib.gsl_record_stride()}
out.update({k: t.cpu() for k, t in grads.items()})
torch.save(out, sys.argv[2])
"""


def _run(tmp_path, tag, sigma, env):
    out = tmp_path / f"{tag}.pt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    res = subprocess.run([sys.executable, "-c", CHILD, root, str(out), str(sigma)], env=e, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return torch.load(out, weights_only=True)


@pytest.mark.parametrize("sigma,env,expect", [
    (0.0, {"GSLOC_TINY_GATHER": "4"}, {"tiny": True, "stride": 1}),
    (0.0, {"GSLOC_AOS": "1"}, {"tiny": True, "stride": 4}),
    (1.0, {"GSLOC_AOS": "1"}, {"tiny": False, "stride": 4}),
    (0.0, {"GSLOC_AOS": "1", "GSLOC_TINY_GATHER": "4"}, {"tiny": True, "stride": 4}),
    (1.0, {"GSLOC_LIB_VARIANT": "occ5"}, {"tiny": False, "stride": 1}),
    (0.0, {"GSLOC_TINY_FUSED": "1"}, {"tiny": True, "stride": 1}),
    (1.0, {"GSLOC_LIB_VARIANT": "xcd"}, {"tiny": False, "stride": 1}),
    (0.0, {"GSLOC_LIB_VARIANT": "xcd"}, {"tiny": True, "stride": 1}),
])
def test_experimental_path_matches_default(tmp_path, sigma, env, expect):
    base = _run(tmp_path, "base", sigma, {"GSLOC_AOS": "0", "GSLOC_TINY_GATHER": "16", "GSLOC_LIB_VARIANT": "", "GSLOC_TINY_FUSED": "0"})
    got = _run(tmp_path, "exp", sigma, env)
    assert base["stride"] == 1 and got["stride"] == expect["stride"] and got["tiny"] == expect["tiny"] == base["tiny"]
    assert torch.equal(got["render"], base["render"]) and torch.equal(got["alphas"], base["alphas"])
    for k in ("viewmat", "means", "quats", "scales", "opacities", "colors"):
        a, b = got[k].double(), base[k].double()
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, k


def test_legacy_pair_on_the_gpu():
    """project_gaussians / rasterize_gaussians over the HIP stage operators against the fused HIP rasterization
    (the glue itself is covered on the CPU by tests/test_legacy_cpu.py; this is its first run on hardware)."""
    import gsplat
    from tests.scenes import random_scene, small_pose

    N, W, H = 4000, 200, 150
    sc = random_scene(N, W, H, seed=5, sigma_px=2.0, aniso=True, opacity=(0.3, 1.0), dtype=torch.float32)
    V = torch.linalg.inv(small_pose(1.0, 0.03, dtype=torch.float32)).cuda()
    cu = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in sc.items()}
    fx, fy, cx, cy = (float(sc["K"][0, 0]), float(sc["K"][1, 1]), float(sc["K"][0, 2]), float(sc["K"][1, 2]))
    m1, V1 = cu["means"].clone().requires_grad_(), V.clone().requires_grad_()
    xys, depths, radii, conics, comp, hit, cov3d = gsplat.project_gaussians(
        m1, cu["scales"], 1.0, cu["quats"], V1, fx, fy, cx, cy, H, W, 16)
    img, alpha = gsplat.rasterize_gaussians(xys, depths, radii, conics, hit, cu["rgbs"], cu["opacities"][:, None], H, W, 16,
                                            return_alpha=True)
    m2, V2 = cu["means"].clone().requires_grad_(), V.clone().requires_grad_()
    rc, ra, meta = gsplat.rasterization(m2, cu["quats"], cu["scales"], cu["opacities"], cu["rgbs"], V2[None], cu["K"][None],
                                        W, H, render_mode="RGB")
    assert torch.equal(radii, meta["radii"][0]) and cov3d.shape == (N, 6)
    assert float((img - rc[0]).abs().max()) < 1e-5 and float((alpha - ra[0, ..., 0]).abs().max()) < 1e-5
    w = torch.linspace(0.5, 1.5, img.numel(), device="cuda").reshape(img.shape)
    (img * w).sum().backward()
    (rc[0] * w).sum().backward()
    assert float((V1.grad - V2.grad).abs().max()) < 1e-4 * float(V2.grad.abs().max())
    assert float((m1.grad - m2.grad).abs().max()) < 1e-4 * float(m2.grad.abs().max())


@pytest.mark.parametrize("seed", range(4))
def test_hip_binning_on_adversarial_inputs(seed):
    """The HIP tile binning (stage operators) on the inputs of tests/test_c_oracle.py's adversarial case: centres
    outside the image, radii larger than the image, exact tile boundaries, equal depths.  Bit-exact against the
    oracle.  (New in the round without GPU access: first run pending.)"""
    import gsplatloc_amd as A
    from oracle import gsplat_oracle as G

    g = torch.Generator().manual_seed(100 + seed)
    N, W, H, ts = 400, 150 + 7 * seed, 90 + 5 * seed, 16
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    m2 = torch.stack([torch.rand(N, generator=g) * (W + 120) - 60, torch.rand(N, generator=g) * (H + 120) - 60], -1)
    m2[:40] = torch.round(m2[:40] / ts) * ts
    radii = torch.randint(1, 40, (N,), generator=g, dtype=torch.int32)
    radii[40:50] = 400
    radii[50:70] = 0
    dep = torch.rand(N, generator=g) * 5 + 0.5
    dep[70:90] = dep[70]
    tpg, ids, fids = G.isect_tiles(m2[None], radii[None], dep[None], ts, tw, th)
    offs = G.isect_offset_encode(ids, 1, tw, th)
    t2, i2, f2 = A.isect_tiles(m2[None].cuda(), radii[None].cuda(), dep[None].cuda(), ts, tw, th)
    o2 = A.isect_offset_encode(i2, 1, tw, th)
    assert torch.equal(t2.cpu(), tpg) and torch.equal(i2.cpu(), ids) and torch.equal(f2.cpu(), fids)
    assert torch.equal(o2.cpu(), offs)


def test_hip_frustum_clamp_branches():
    """The clamped branch of the EWA Jacobian (centres beyond 1.3x the half field of view) and the near / far /
    radius_clip culls through the HIP path, against the float64 oracle.  (First run pending.)"""
    import gsplatloc_amd as A
    from oracle import gsplat_oracle as G
    from tests.scenes import frustum_clamp_scene

    sc = frustum_clamp_scene()
    W, H, kw = sc["W"], sc["H"], sc["kw"]
    ins = [sc[k].clone().requires_grad_() for k in ("means", "quats", "scales", "opacities", "rgbs")]
    Vo = sc["V"].clone().requires_grad_()
    ro, ao, mo = G.rasterization(*ins, Vo[None], sc["K"][None], W, H, render_mode="RGB+D", **kw)
    ((ro[0] * sc["v_render"]).sum() + (ao[0, ..., 0] * sc["v_alphas"]).sum()).backward()
    gin = [sc[k].float().cuda().requires_grad_() for k in ("means", "quats", "scales", "opacities", "rgbs")]
    Vg = sc["V"].float().cuda().requires_grad_()
    rg, ag, mg = A.rasterization(*gin, viewmats=Vg[None], Ks=sc["K"].float().cuda()[None], width=W, height=H,
                                 render_mode="RGB+D", packed=False, **kw)
    assert torch.equal(mg["radii"][0].cpu(), mo["radii"][0])
    ((rg[0] * sc["v_render"].float().cuda()).sum() + (ag[0, ..., 0] * sc["v_alphas"].float().cuda()).sum()).backward()
    bad = (rg[0].cpu().double() - ro[0].detach()).abs() > 2e-5 + 1e-4 * ro[0].detach().abs()
    assert float(bad.double().mean()) < 3e-3
    for got, want in zip(gin + [Vg], ins + [Vo]):
        a, b = got.grad.cpu().double(), want.grad
        if a.shape == (4, 4):
            a, b = a[:3], b[:3]
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-9
