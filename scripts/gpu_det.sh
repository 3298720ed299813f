#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -k "deterministic or two_rank or two_ranks" > gpurun_out/t_det.log 2>&1; rc=$?
tail -5 gpurun_out/t_det.log; if [ $rc -ge 124 ]; then exit $rc; fi
python - <<'PY'
import torch, time, sys
sys.path.insert(0, ".")
from gsplatloc_amd.context import RenderContext, time_stages
from gsplatloc_amd.synthetic import perturbed_pose, random_scene
dev = torch.device("cuda"); N, W, H = 1_000_000, 1200, 680
sc = random_scene(N, W, H, sigma_px=1.0, device=dev)
V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous(); K = sc["K"].contiguous()
v = torch.zeros(H, W, 4, device=dev); v[..., 3] = torch.randn(H, W, device=dev); va = torch.zeros(H, W, 1, device=dev)
for det in (False, True):
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True, deterministic=det)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, K)
    ctx.calibrate(*inp)
    st = time_stages(ctx, inp, v, va, True, steps=10)
    print("deterministic" if det else "atomic       ", {k: round(x * 1e3) for k, x in st.items()}, "us, total", round(sum(st.values()) * 1e3), flush=True)
PY
