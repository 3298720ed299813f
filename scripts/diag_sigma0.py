#!/usr/bin/env python3
"""Dev tool: where does a sigma_px -> 0 step at R size spend its time?  Per-step HIP events over many graph replays
(distribution, not a wall-clock mean over 15 replays), the five stage timers, eager and graph, both Gaussian orders.
usage: diag_sigma0.py [--n 1000000] [--steps 200] [--orders raster,random] [--sigmas 0.0]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=680)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--orders", default="raster,random")
    ap.add_argument("--sigmas", default="0.0")
    a = ap.parse_args()
    from gsplatloc_amd import context as C
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    dev = torch.device("cuda", 0)
    for sigma in [float(s) for s in a.sigmas.split(",")]:
        for order in a.orders.split(","):
            sc = random_scene(a.n, a.width, a.height, sigma_px=sigma, device=dev, order=order)
            V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
            K = sc["K"].contiguous()
            ctx = C.RenderContext(a.n, a.width, a.height, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
            inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, K)
            n_is = ctx.calibrate(*inp)
            g = torch.Generator().manual_seed(1)
            v = torch.zeros(a.height, a.width, 4)
            v[..., 3] = torch.randn(a.height, a.width, generator=g)
            v = v.to(dev)
            va = torch.zeros(a.height, a.width, 1, device=dev)

            def step():
                ctx.forward(*inp)
                ctx.backward(v, va, full=True)

            out = {"sigma_px": sigma, "order": order, "n_isects": n_is, "tiny": ctx.tiny, "bin_cap": ctx.bin_cap,
                   "capacity": ctx.capacity}
            for mode in ("eager", "graph"):
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    run = step
                    if mode == "graph":
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, stream=side):
                            step()
                        run = graph.replay
                    torch.cuda.synchronize()
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
                    for e0, e1 in ev:
                        e0.record()
                        run()
                        e1.record()
                    torch.cuda.synchronize()
                ms = sorted(x.elapsed_time(y) for x, y in ev)
                out[mode] = {"min": ms[0], "p10": ms[len(ms) // 10], "median": ms[len(ms) // 2], "p90": ms[9 * len(ms) // 10],
                             "max": ms[-1], "mean": sum(ms) / len(ms)}
            out["stages"] = C.time_stages(ctx, inp, v, va, True, steps=20)
            out["flags"] = ctx.flags.tolist()
            ctx.check_capacity()
            print(json.dumps(out), flush=True)
            del ctx


if __name__ == "__main__":
    main()
