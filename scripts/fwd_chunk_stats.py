"""Dev tool: chunk statistics of the compositing forward at workload R (no early termination modelled): per (tile, batch
of 256, quadrant) the number of candidates, hence 64-candidate chunks and their fill -- the cost model of the per-chunk
mask construction (109 VALU per chunk) against the per-trip cost (72 VALU)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsplatloc_amd.context import RenderContext
from gsplatloc_amd.synthetic import perturbed_pose, random_scene

dev = torch.device("cuda")
N, W, H = 1_000_000, 1200, 680
sc = random_scene(N, W, H, sigma_px=1.0, device=dev)
V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False)
inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, sc["K"].contiguous())
ctx.calibrate(*inp)
ctx.forward(*inp)
torch.cuda.synchronize()
n = ctx.check_capacity()
offs = ctx.offs.long()
ids = ctx.flatten_ids[:n].long()
sizes = offs[1:] - offs[:-1]
tile = torch.repeat_interleave(torch.arange(ctx.n_tiles, device=dev), sizes)
pos = torch.arange(n, device=dev) - offs[:-1][tile]
batch = pos // 256
x, y, r = ctx.Q0[ids, 0], ctx.Q0[ids, 1], ctx.Q1[ids, 3]
tx0, ty0 = (tile % ctx.tw).float() * 16, (tile // ctx.tw).float() * 16
tot_chunks = tot_cand = 0
for q in range(4):
    cx, cy = tx0 + 4 + 8 * (q & 1), ty0 + 4 + 8 * (q >> 1)
    hit = ((x - cx).abs() <= r + 3.5) & ((y - cy).abs() <= r + 3.5)
    key = (tile * 64 + batch)[hit]
    cnt = torch.bincount(key)
    cnt = cnt[cnt > 0]
    chunks = ((cnt + 63) // 64).sum()
    tot_chunks += int(chunks); tot_cand += int(cnt.sum())
print(f"intersections {n}; (quadrant, batch) candidates {tot_cand} ({tot_cand / n:.2f} per intersection); chunks {tot_chunks} "
      f"(mean fill {tot_cand / tot_chunks:.1f} of 64); mask construction {tot_chunks * 109 / 1e6:.1f} M VALU per launch")

# ---- trips per chunk (no early termination modelled): two 32-candidate halves, as the kernel walks them, against one
# 64-candidate mask per lane
tot_now = tot_one = tot_pairs = 0
for q in range(4):
    cx, cy = tx0 + 4 + 8 * (q & 1), ty0 + 4 + 8 * (q >> 1)
    hit = ((x - cx).abs() <= r + 3.5) & ((y - cy).abs() <= r + 3.5)
    key = (tile * 64 + batch)[hit]
    xs, ys, rs_ = x[hit] - (cx[hit] - 4 + 0.5), y[hit] - (cy[hit] - 4 + 0.5), r[hit]   # pixel centres of the quadrant at 0..7
    uniq, inv = torch.unique(key, return_inverse=True)          # key is ascending (list order): rank inside the group
    first = torch.zeros(uniq.numel(), dtype=torch.long, device=dev).scatter_reduce(0, inv, torch.arange(key.numel(), device=dev), "amin", include_self=False)
    rank = torch.arange(key.numel(), device=dev) - first[inv]
    base = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.long, device=dev), (torch.bincount(inv) + 63) // 64]), 0)[:-1]
    chunk = base[inv] + rank // 64
    half = (rank % 64) // 32
    nch = int(chunk.max()) + 1
    lox, hix = torch.ceil(xs - rs_).clamp(min=0), torch.floor(xs + rs_).clamp(max=7)
    loy, hiy = torch.ceil(ys - rs_).clamp(min=0), torch.floor(ys + rs_).clamp(max=7)
    cnt = torch.zeros(nch * 2, 64, dtype=torch.int32, device=dev)
    for p in range(64):
        px_, py_ = p % 8, p // 8
        m = (lox <= px_) & (px_ <= hix) & (loy <= py_) & (py_ <= hiy)
        cnt[:, p] = torch.bincount((chunk * 2 + half)[m], minlength=nch * 2).int()
    c2 = cnt.view(nch, 2, 64)
    tot_now += int(((c2 + 1) // 2).amax(2).sum())
    tot_one += int(((c2.sum(1) + 1) // 2).amax(1).sum())
    tot_pairs += int(cnt.sum())
print(f"trips without early termination: two halves {tot_now}, one 64-candidate mask {tot_one} ({tot_one / tot_now:.3f}); "
      f"(pixel, candidate) pairs inside a box {tot_pairs}: lane use {tot_pairs / (tot_now * 128):.2f} now")
