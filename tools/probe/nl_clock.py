"""In-kernel clock of the bf16x6 contraction (NonLocal logits): build with -DARFE_PROBES -DARFE_CLOCK_PROBE and run with
ARFE_LIB=arfe_amd/libarfe_var_clk.so; launches the NonLocal forward back to back for about two seconds, read the last lines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arfe_amd import functional as Fn
dev = torch.device('cuda:0')
B, C, h, w = 2, 256, 50, 84
g = torch.Generator().manual_seed(3)
m = torch.randn(B, C, h, w, generator=g).to(dev)
ws = [(torch.randn(C, C, generator=g) * 0.05).to(dev) for _ in range(4)]
bs = [(torch.randn(C, generator=g) * 0.1).to(dev) for _ in range(4)]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    Fn.nonlocal_fwd_op(m, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], ws[3], bs[3])
torch.cuda.synchronize()
