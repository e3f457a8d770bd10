"""isolated timing of the NonLocal block forward / backward at the bench size (B=2, C=256, 50x84); ARFE_NL_T selects routes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arfe_amd import functional as Fn
from tools.kernel_bench import timeit
dev = torch.device('cuda:0')
B, C, h, w = 2, 256, 50, 84
g = torch.Generator().manual_seed(3)
m = torch.randn(B, C, h, w, generator=g).to(dev)
ws = [(torch.randn(C, C, generator=g) * 0.05).to(dev) for _ in range(4)]
bs = [(torch.randn(C, generator=g) * 0.1).to(dev) for _ in range(4)]
bsf, saved = Fn.nonlocal_fwd_op(m, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], ws[3], bs[3])
dbsf = torch.randn_like(bsf)
out = Fn.nonlocal_bwd_op(dbsf, m, ws[0], ws[1], ws[2], ws[3], saved)
torch.cuda.synchronize()
print("check", float(bsf.double().abs().sum()), float(out[0].double().abs().sum()), float(sum(t.double().abs().sum() for t in out[1])))
f = timeit(lambda: Fn.nonlocal_fwd_op(m, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], ws[3], bs[3]), iters=10)
b = timeit(lambda: Fn.nonlocal_bwd_op(dbsf, m, ws[0], ws[1], ws[2], ws[3], saved), iters=10)
print(f"NL_T={os.environ.get('ARFE_NL_T','0'):>3}  fwd {f*1e3:8.1f} us   bwd {b*1e3:8.1f} us")
