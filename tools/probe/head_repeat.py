"""run-to-run reproducibility of the AR-RFF head forward (plain vs prepacked)"""
import sys, torch
sys.path.insert(0, '.')
import arfe_amd
dev = torch.device('cuda:0')
torch.manual_seed(5)
head = arfe_amd.build_head(dict(type='MultiRoIsBBoxHead', in_channels=32, conv_out_channels=32, fc_out_channels=48,
                                roi_feat_size=7, num_classes=80)).to(dev)
head.init_weights()
x = torch.randn(70, 96, 7, 7, device=dev)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
with torch.no_grad():
    outs = [head(x) for _ in range(4)] + [head(x, head.prepack()) for _ in range(4)]
torch.cuda.synchronize()
for i, (c, r) in enumerate(outs):
    print(i, rel(c, outs[0][0]), rel(r, outs[0][1]))
