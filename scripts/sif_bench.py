"""Dev tool: whole step (graph replay) at R / X / D with and without the forward that sorts its own bins."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsplatloc_amd.context import RenderContext
from gsplatloc_amd.synthetic import depth_frame_scene, perturbed_pose, random_scene

dev = torch.device("cuda")
for name in sys.argv[1:] or ["R", "D"]:
    if name == "D":
        W, H = 1200, 680
        sc = depth_frame_scene(W, H, stride=1, holes=False, device=dev); V = sc["viewmat"]
    else:
        N, W, H = (5_000_000, 1920, 1080) if name == "X" else (1_000_000, 1200, 680)
        sc = random_scene(N, W, H, sigma_px=1.0, device=dev); V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    N = sc["means"].shape[0]
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, sc["K"].contiguous())
    g = torch.Generator().manual_seed(1)
    v = torch.zeros(H, W, 4); v[..., 3] = torch.randn(H, W, generator=g); v = v.to(dev)
    va = torch.zeros(H, W, 1, device=dev)
    for sif in (False, True):
        ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True, sort_in_forward=sif)
        ctx.calibrate(*inp)
        def step():
            ctx.forward(*inp); ctx.backward(v, va, full=True)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step(); step(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                step()
        torch.cuda.synchronize()
        for _ in range(5): gr.replay()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for a, b in ev:
            a.record(); gr.replay(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print(name, "sort_in_forward", sif, "active", ctx.sorts_in_forward(), "bin_cap", ctx.bin_cap, "median ms", round(ms[30], 4), flush=True)
        ctx.check_capacity()
