"""Dev tool: wall time of the drop-in call (gsplat.rasterization forward + backward through autograd, allocation and
the size read-back included) next to RenderContext on the same scene."""
import sys, time
import torch
sys.path.insert(0, ".")
import gsplatloc_amd as A
from gsplatloc_amd.synthetic import depth_frame_scene, random_scene, perturbed_pose

def api(sc, V, W, H, full, n=400):
    def step():
        Vg = V.clone().requires_grad_()
        m = sc["means"].clone().requires_grad_(full)
        rc, ra, meta = A.rasterization(means=m, quats=sc["quats"], scales=sc["scales"], opacities=sc["opacities"],
                                       colors=sc["sh"], sh_degree=1, viewmats=Vg[None], Ks=sc["K"][None], width=W, height=H,
                                       packed=False, render_mode="RGB+ED", near_plane=1e-2, far_plane=1e10)
        (rc[..., 3] * 0.5).sum().backward()
    for _ in range(50): step()  # context creation, calibration, clocks
    best = 1e9
    for _ in range(3):  # best of three windows of n calls (a 16-CPU share of a busy host: windows differ by 20 %)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best

api(depth_frame_scene(640, 480, stride=3), depth_frame_scene(640, 480, stride=3)["viewmat"], 640, 480, True, n=100)  # (the
# first measurement of a process reads 30 % high whatever it measures: library load, allocator, clocks)
for name in ("S", "R"):
    if name == "S":
        sc = depth_frame_scene(640, 480, stride=3); W, H = 640, 480; V = sc["viewmat"]
    else:
        W, H = 1200, 680
        sc = random_scene(1_000_000, W, H, device="cuda"); V = torch.linalg.inv(perturbed_pose()).cuda()
    print(name, f"drop-in API pose-only {api(sc, V, W, H, False):.3f} ms, with Gaussian gradients {api(sc, V, W, H, True):.3f} ms per fwd+bwd")
