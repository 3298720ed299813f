"""Per-kernel timing of the fused pipeline with HIP events (dev tool)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplatloc_amd as A
from gsplatloc_amd.synthetic import random_scene, perturbed_pose
N, W, H = int(sys.argv[1]), 1200, 680
sig = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sc = random_scene(N, W, H, sigma_px=sig, device='cuda')
V = torch.linalg.inv(perturbed_pose()).cuda()[None]
g = torch.Generator().manual_seed(1)
v = torch.zeros(1, H, W, 4); v[..., 3] = torch.randn(1, H, W, generator=g); v = v.cuda()
def step():
    Vg = V.clone().requires_grad_()
    rc, ra, meta = A.rasterization(means=sc['means'], quats=sc['quats'], scales=sc['scales'], opacities=sc['opacities'], colors=sc['sh'],
        sh_degree=1, viewmats=Vg, Ks=sc['K'][None], width=W, height=H, packed=False, render_mode='RGB+ED')
    rc.backward(v)
    return meta
for _ in range(3): meta = step()
torch.cuda.synchronize()
import time; t = time.time()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.time() - t) / 10
print(f"sigma={sig} I/N={meta['flatten_ids'].numel()/N:.2f} ms/step={dt*1e3:.3f} G/s={N/dt:.3e}")
