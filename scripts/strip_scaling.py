"""One-GPU estimate of the tile-strip strong-scaling curve of bench.py: for world = 1,2,4,8 build every
rank's strip context in turn on this GPU, replay its graph, and report max-over-ranks step time
(excludes the 16-float all-reduce).  python scripts/strip_scaling.py [sigma_px] [order] [R|X]
(X: 5 M Gaussians, 1920x1080, fp16-staged records -- BASELINE.json configs[4])"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsplatloc_amd.context import RenderContext, time_stages
from gsplatloc_amd.parallel import strip_rows, gaussians_for_strip
from gsplatloc_amd.synthetic import random_scene, perturbed_pose

sig = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
order = sys.argv[2] if len(sys.argv) > 2 else "random"
size = sys.argv[3] if len(sys.argv) > 3 else "R"
N, W, H = (5_000_000, 1920, 1080) if size == "X" else (1_000_000, 1200, 680)
staging = "fp16" if size == "X" else "fp32"
dev = torch.device("cuda")
sc = random_scene(N, W, H, sigma_px=sig, device=dev, order=order)
viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
K = sc["K"].contiguous()
g = torch.Generator().manual_seed(1)
v = torch.zeros(H, W, 4); v[..., 3] = torch.randn(H, W, generator=g); v = v.to(dev)
va = torch.zeros(H, W, 1, device=dev)
cal = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False, reorder=False)  # (indexes sc[...])
cal.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
base = None
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world):
        rows = strip_rows(cal.offs, cal.tw, cal.th, world)[rank] if world > 1 else (0, cal.th)
        if world > 1:
            idx = cal.gaussians_reaching(rows)
            loc = {k: sc[k][idx].contiguous() for k in ("means", "quats", "scales", "opacities", "sh")}
        else:
            loc = sc
        n = loc["means"].shape[0]
        ctx = RenderContext(n, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, tile_rows=rows, full_grads=True,
                            staging=staging)
        inp = (loc["means"], loc["quats"], loc["scales"], loc["opacities"], loc["sh"], viewmat, K)
        n_strip = ctx.calibrate(*inp)
        if world > 1:
            want = int(cal.offs[rows[1] * cal.tw] - cal.offs[rows[0] * cal.tw])
            assert n_strip == want, (rows, n_strip, want)  # the strip's lists = the full frame's entries of its rows
        def step():
            ctx.forward(*inp); ctx.backward(v, va, full=True)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                step()
        torch.cuda.synchronize()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): gr.replay()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t) / 20 * 1e3)
        if rank == world // 2 or world == 2:
            st = time_stages(ctx, inp, v, va, True, steps=10)
            print(f"   world {world} rank {rank}: rows {rows} n={n} isects={int(ctx.n_is.item())} placed={ctx.order_ids is not None} tiny={ctx.tiny} stages " + " ".join(f"{k}={x*1e3:.0f}us" for k, x in st.items()), flush=True)
        del gr, ctx
    worst = max(times)
    base = base or worst
    print(f"world {world}: max-over-ranks {worst:.3f} ms  (ranks: {' '.join('%.3f' % x for x in times)})  speed-up {base / worst:.2f}x", flush=True)
