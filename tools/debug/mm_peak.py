import torch, time
torch.manual_seed(0)
def t(a, b, n=20):
    for _ in range(3): a @ b
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): a @ b
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for dt in (torch.bfloat16, torch.float16):
    for M,N,K in ((8192,8192,8192),(8300,4608,384),(16384,16384,4096)):
        for kind in ("zeros","randn"):
            a = (torch.zeros if kind=="zeros" else torch.randn)(M,K,device="cuda",dtype=dt)
            b = (torch.zeros if kind=="zeros" else torch.randn)(K,N,device="cuda",dtype=dt)
            ms=t(a,b); print(f"hipBLASLt/rocBLAS {dt} {M}x{N}x{K} {kind}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
