"""GPU: repeated-run consistency of the integer part of the path.  The sorted tile lists, offsets and the forward
image must not depend on the order in which workgroups reserve bin slots or waves finish: 200 forwards of the same
inputs (binned projection, register sort, compositing) give bit-identical lists and images, at a depth-frame cloud
and at a random cloud; with a moving pose in between so that stale state would show."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["depth_frame", "random"])
def test_repeated_forwards_are_bit_identical(kind):
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene, perturbed_pose, random_scene
    dev = "cuda"
    if kind == "depth_frame":
        sc = depth_frame_scene(640, 480, stride=2, device=dev)
        W, H, V = 640, 480, sc["viewmat"]
    else:
        W, H = 400, 300
        sc = random_scene(150_000, W, H, sigma_px=1.0, device=dev)
        V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    N = sc["means"].shape[0]
    V2 = torch.linalg.inv(perturbed_pose(0.8, 0.02, seed=3)).to(dev).contiguous() @ V
    rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False)
    ins = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"])
    K = sc["K"].contiguous()
    rc.calibrate(*ins, V, K, headroom=1.6)
    v = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(1)).to(dev)
    va = torch.zeros(H, W, 1, device=dev)
    ref = {}
    for it in range(200):
        pose = V if it % 2 == 0 else V2
        rc.forward(*ins, pose, K)
        g = rc.backward(v, va, full=False)
        n = int(rc.n_is.item())
        cur = (n, rc.offs.clone(), rc.flatten_ids[:n].clone(), rc.render.clone(), rc.last_ids.clone())
        key = it % 2
        if key not in ref:
            ref[key] = cur
            assert n > 0 and rc.check_capacity() == n
            continue
        want = ref[key]
        assert cur[0] == want[0], (it, cur[0], want[0])
        for a, b, name in zip(cur[1:], want[1:], ("offsets", "flatten_ids", "render", "last_ids")):
            assert torch.equal(a, b), (it, name)
        assert torch.isfinite(g["viewmat"]).all()
    assert rc.check_capacity() > 0
