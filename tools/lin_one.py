"""Time one plain GEMM through arfe_linear_fwd (development aid): lin_one.py R in out"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import linear, conv
R, i, o = (int(v) for v in sys.argv[1:4])
linear(R, i, o)
