import sys, time, torch
sys.path.insert(0, '/root/repo')
import gsplatloc_amd as A
from gsplatloc_amd.synthetic import random_scene, perturbed_pose
N, W, H = int(sys.argv[1]), 1200, 680
sig = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sc = random_scene(N, W, H, sigma_px=sig, device='cuda')
V = torch.linalg.inv(perturbed_pose()).cuda()[None]
def step(full):
    Vg = V.clone().requires_grad_()
    m = sc['means'].clone().requires_grad_(full)
    rc, ra, meta = A.rasterization(means=m, quats=sc['quats'], scales=sc['scales'], opacities=sc['opacities'], colors=sc['sh'],
        sh_degree=1, viewmats=Vg, Ks=sc['K'][None], width=W, height=H, packed=False, render_mode='RGB+ED', near_plane=1e-2, far_plane=1e10)
    (rc[..., 3] * 0.5).sum().backward()
    return meta
for full in (False, True):
    for _ in range(3): meta = step(full)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): step(full)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(f"full={full} N={N} I={meta['flatten_ids'].numel()} I/N={meta['flatten_ids'].numel()/N:.2f} ms/step={dt*1e3:.3f} G/s={N/dt:.3e}")
