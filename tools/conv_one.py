"""Time the 3x3 conv forward GEMM at the bench size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import conv
conv(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
