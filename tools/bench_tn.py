import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K
from tools.bench_gemm import run
K.set_precision("bf16")
R = 8300
for sk in (1, 2, 4, 8, 14, 28, 56):
    run("fc1 dW TN", 1536, 384, R, True, False, splitk=sk)
for sk in (1, 4, 14):
    run("fc1 dW as NN-shape check (M=1536,N=384,K=8300) NT", 1536, 384, R, False, True, splitk=sk)
