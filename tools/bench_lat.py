import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spe_amd import kernels as K
from tools.bench_gemm import run
K.set_precision("bf16")
for (M, N, Kd) in [(2048, 2048, 384), (2048, 2048, 3840), (4096, 3072, 384), (4096, 3072, 3840), (8192, 6144, 384), (8192, 6144, 3840), (128, 128, 3840), (128, 128, 38400)]:
    run("NT lat probe", M, N, Kd, False, True)
