#!/usr/bin/env python3
"""Dev: iteration-0 render of the GraphTracker's context against a plain RenderContext on the normal-term test scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_tracker as TT
from gsplatloc_amd.graph_tracker import GraphTracker
from gsplatloc_amd.context import RenderContext

DEV = torch.device("cuda")
M, fp, K, pts0, pts1, scales0, scales1 = TT._setup()
W, H = fp["W"], fp["H"]
src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
src_depth = src_depth[None, ..., None]
for nl in (0.05, 0.0):
    cfg = M.TrackerConfig(max_steps=2, min_step=5, patience=1000, depth_lambda=0.7, normal_lambda=nl)
    ref = M.PoseTracker(cfg, engine="context").track_frame(pts0.to(DEV), fp["rgb"].to(DEV), src_depth, fp["c2w0"].to(DEV),
                                                            fp["c2w1"].to(DEV), K.to(DEV), W, H, scales=scales0.to(DEV))
    gt = GraphTracker(pts0.shape[0], W, H, cfg)
    gt.load_frame(pts0.to(DEV), fp["rgb"].to(DEV), scales0.to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV))
    res = gt.run()
    print("normal_lambda", nl, "graph", res.losses[:2], "ref", ref.losses[:2], "rel0", abs(res.losses[0] - ref.losses[0]) / ref.losses[0])
    print("  tiny", gt.rc.tiny, "sorts_in_forward", gt.rc.sorts_in_forward(), "render mode D", gt.rc.D)
# the same frame through two plain contexts: the sorting forward against the separate sort launch
from gsplatloc_amd.my_gsplat.model import GSModel
gs = GSModel(pts0.to(DEV), fp["rgb"].to(DEV), config=cfg.gs, scales=scales0.to(DEV))
opac = torch.sigmoid(gs.opacities).contiguous(); sh = torch.cat([gs.sh0, gs.shN], 1).contiguous()
statics = (gs.means3d.contiguous(), gs.quats.contiguous(), gs.scales.contiguous(), opac, sh)
V = torch.linalg.inv(fp["c2w0"].to(DEV)).contiguous()
outs = {}
for name, kw in (("plain", dict(sort_in_forward=False)), ("sif", dict(sort_in_forward=True))):
    for mode in ("RGB+ED", "ED"):
        rc = RenderContext(pts0.shape[0], W, H, mode, sh_degree=cfg.gs.sh_degree, K_sh=(cfg.gs.sh_degree + 1) ** 2, device=DEV,
                           near_plane=cfg.gs.near_plane, far_plane=cfg.gs.far_plane, full_grads=False, **kw)
        with torch.no_grad():
            rc.calibrate(*statics, V, K.to(DEV).contiguous(), headroom=1.5)
            rc.forward(*statics, V, K.to(DEV).contiguous())
        torch.cuda.synchronize()
        outs[(name, mode)] = rc.render[..., -1].clone()
        print(name, mode, "sorts_in_forward", rc.sorts_in_forward(), "sum", float(outs[(name, mode)].double().sum()), "zeros", int((outs[(name, mode)] == 0).sum()))
base = outs[("plain", "RGB+ED")]
for k, v in outs.items():
    d = (v - base).abs()
    print(k, "max abs diff vs plain RGB+ED", float(d.max()), "pixels differing", int((d > 0).sum()))
