import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsplatloc_amd import _lib
lib = _lib.load_library()
x = (torch.arange(32)[None, :] * 1000 + torch.arange(64)[:, None]).float().cuda()
out = torch.empty(64, device='cuda')
_lib.check(lib.gsl_debug_reduce_scatter(x.data_ptr(), out.data_ptr(), None), 'rs')
exp = (torch.arange(32) * 64000 + 2016).float().repeat_interleave(2)
print('max err', float((out.cpu() - exp).abs().max()))
print(out.cpu().tolist()[:16])
