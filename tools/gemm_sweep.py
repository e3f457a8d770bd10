"""Sweep tile kind x split-K for the GEMM shapes of the path (development aid; uses ARFE_GEMM_PLAN, one process per plan
because the library reads the variable once).  usage: gemm_sweep.py            -> spawns the sweep
                                                      gemm_sweep.py --one      -> times the shapes under the current env"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # (R = N rows, in = K, out = M) through arfe_linear_fwd: y[R,out] = x[R,in] w[out,in]^T
    (8400, 768, 256),      # NonLocal dm      (M=256, N=B*L, K=3C)
    (8400, 256, 256),      # NonLocal dy / out projection
    (256, 8400, 256),      # NonLocal dw_out  (K = B*L)
    (256, 8400, 768),      # NonLocal dwcat
    (8400, 4200, 256),     # NonLocal y / dg / dtheta / dphi (per image K = L)
    (1024, 12544, 1024),   # fc1
    (1024, 1024, 1024),    # fc2
]
if '--one' in sys.argv:
    import torch
    from tools.kernel_bench import timeit, dev, L, ptr, stream, check
    for (R, i, o) in SHAPES:
        x = torch.randn(R, i, device=dev); w = torch.randn(o, i, device=dev) * 0.02; b = torch.zeros(o, device=dev)
        y = torch.empty(R, o, device=dev)
        ms = timeit(lambda: check(L.arfe_linear_fwd(R, i, o, ptr(x), ptr(w), ptr(b), 0, ptr(y), 0, 0, stream()), 'lin'), iters=20)
        print(f"{os.environ.get('ARFE_GEMM_PLAN', 'auto'):>5s}  R={R:5d} in={i:5d} out={o:4d}: {ms*1e3:8.1f} us", flush=True)
else:
    for plan in ['auto', 'S1', 'S2', 'S4', 'S8', 'S16', 'S32', 'S64', 'L1', 'L2', 'L4', 'L8', 'L16', 'L32']:
        env = dict(os.environ)
        if plan != 'auto':
            env['ARFE_GEMM_PLAN'] = plan
        subprocess.run([sys.executable, os.path.abspath(__file__), '--one'], env=env, check=True)
