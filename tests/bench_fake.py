"""Run bench.main() WITHOUT a GPU: torch.cuda, the process-group backend and the RenderContext are replaced by
host stand-ins, so that the control flow of the benchmark driver -- argument handling, strip assignment, the
collective, barrier/timing, max over ranks, the JSON line -- can be exercised on the CPU, also with WORLD_SIZE > 1
(gloo).  Test helper (tests/test_bench_flow_cpu.py starts it as `python tests/bench_fake.py <bench args>`); the
numbers it prints mean nothing.
"""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

_real_device = torch.device


class _Stream:
    def __init__(self, *a, **k):
        pass


class _Event:
    def __init__(self, enable_timing=False):
        pass

    def record(self):
        pass

    def elapsed_time(self, other):
        return 0.125


def _device(*args, **kwargs):
    if args and isinstance(args[0], str) and args[0].startswith("cuda"):
        return _real_device("cpu")
    return _real_device(*args, **kwargs)


class FakeContext:
    """Shape-compatible stand-in of gsplatloc_amd.context.RenderContext (host tensors, no kernels)."""

    def __init__(self, N, width, height, render_mode="RGB+ED", sh_degree=1, K_sh=4, device="cpu", tile_rows=None,
                 full_grads=True, **kw):
        self.N, self.W, self.H = N, width, height
        self.tw, self.th = (width + 15) // 16, (height + 15) // 16
        self.n_tiles = self.tw * self.th
        self.rows = tile_rows if tile_rows is not None else (0, self.th)
        self.tiny = False
        g = torch.Generator().manual_seed(0)
        self.Q0 = torch.rand(N, 4, generator=g) * torch.tensor([float(width), float(height), 1.0, 1.0])
        self.radii = torch.full((N,), 4, dtype=torch.int32)
        per_tile = 7  # (a constant: the strip contexts of the N > 1 path must count what the calibration context counted)
        self.offs = torch.arange(self.n_tiles + 1, dtype=torch.int32) * per_tile
        self.v_viewmat = torch.zeros(4, 4)
        self.calls = 0

    def calibrate(self, *a, **k):
        return int(self.offs[self.rows[1] * self.tw] - self.offs[self.rows[0] * self.tw])

    def forward(self, *a):
        self.calls += 1

    def backward(self, v_render, v_alphas, full=True):
        self.v_viewmat = torch.full((4, 4), float(self.N))
        return {"viewmat": self.v_viewmat}

    def check_capacity(self):
        return self.calibrate()

    def _project(self, *a):
        pass

    order_ids = storage_of = None

    def _place(self, *a):
        return a

    def gaussians_reaching(self, rows, guard_tiles=1):
        from gsplatloc_amd.parallel import gaussians_for_strip
        return gaussians_for_strip(self.Q0[:, 0:2], self.radii, rows, guard_tiles=guard_tiles)

    _bin = _raster_fwd = _project
    _raster_bwd = _project_bwd = _project


def main():
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Stream = _Stream
    torch.cuda.Event = _Event
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.device = _device
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **kw: real_init("gloo")
    import gsplatloc_amd.context as CX
    CX.RenderContext = FakeContext
    CX.pack_pose_reduce = lambda v_viewmat, out16, loss_partials=None: out16.copy_(v_viewmat.reshape(16))
    import bench
    sys.argv = ["bench.py"] + sys.argv[1:]
    bench.main()


if __name__ == "__main__":
    main()
