import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spe_amd import kernels as K
from tools.bench_gemm import run
K.set_precision("bf16")
run("fc1 fwd NT", 8300, 1536, 384, False, True)
run("fc2 fwd NT", 8300, 384, 1536, False, True)
