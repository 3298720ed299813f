"""Generates the golden fixtures under tests/golden/ from the float64 CPU oracle.

PARITY UNPINNED: /root/reference holds no fixture for the render path and the reference's
rasterizer (third-party CUDA gsplat) cannot run here, so these vectors are outputs of
oracle/gsplat_oracle.py (itself pinned by finite differences and the sequential twin, see
tests/test_oracle.py).  They freeze the oracle against regressions and give the GPU tests an
oracle-free reference.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gsplat_oracle as G  # noqa: E402
from tests.scenes import random_scene, sh_from_rgb, small_pose  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "rgbed_64x48_n512": dict(N=512, W=64, H=48, sigma_px=1.5, opacity=(0.3, 1.0), aniso=True, mode="RGB+ED"),
    "rgbed_96x64_n1024": dict(N=1024, W=96, H=64, sigma_px=1.2, opacity=None, aniso=False, mode="RGB+ED"),
    "ed_96x64_n1024": dict(N=1024, W=96, H=64, sigma_px=0.0, opacity=None, aniso=False, mode="ED"),
}


def build(name, c):
    sc = random_scene(c["N"], c["W"], c["H"], sigma_px=max(c["sigma_px"], 1e-3), opacity=c["opacity"],
                      aniso=c["aniso"], dtype=torch.float32)
    sh = sh_from_rgb(sc["rgbs"])
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32))[None]
    ins = [sc[k].double().clone().requires_grad_() for k in ("means", "quats", "scales", "opacities")]
    sho = sh.double().clone().requires_grad_()
    Vo = V.double().clone().requires_grad_()
    rc, ra, meta = G.rasterization(*ins, sho, Vo, sc["K"].double()[None], c["W"], c["H"], sh_degree=1,
                                   render_mode=c["mode"])
    g = torch.Generator().manual_seed(123)
    v_c = torch.randn(rc.shape, generator=g, dtype=torch.float64)
    v_a = torch.randn(ra.shape, generator=g, dtype=torch.float64)
    ((rc * v_c).sum() + (ra * v_a).sum()).backward()
    out = dict(
        means=sc["means"].numpy(), quats=sc["quats"].numpy(), scales=sc["scales"].numpy(),
        opacities=sc["opacities"].numpy(), sh=sh.numpy(), viewmat=V[0].numpy(), K=sc["K"].numpy(),
        W=c["W"], H=c["H"], mode=c["mode"], v_render=v_c.float().numpy(), v_alpha=v_a.float().numpy(),
        render=rc.detach().float().numpy(), alpha=ra.detach().float().numpy(),
        radii=meta["radii"].numpy(), n_isects=int(meta["flatten_ids"].numel()),
        g_viewmat=Vo.grad[0].numpy(), g_means=ins[0].grad.float().numpy(), g_scales=ins[2].grad.float().numpy(),
        g_opacities=ins[3].grad.float().numpy(),
    )
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "I/N", out["n_isects"] / c["N"], os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")


if __name__ == "__main__":
    for n, c in CASES.items():
        build(n, c)
