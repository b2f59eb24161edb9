import time, torch
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (b, M, K, N) in ((16, 32768, 2048, 512), (16, 32768, 512, 2048), (16, 32768, 512, 512), (36, 8192, 2048, 512), (1, 524288, 2048, 512), (1, 131072, 18432, 512)):
    A = torch.randn(b, M, K, device="cuda"); B = torch.randn(b, K, N, device="cuda"); C = torch.empty(b, M, N, device="cuda")
    ms = t(lambda: torch.bmm(A, B, out=C))
    print(f"bmm b={b} M={M} K={K} N={N}: {ms:.3f} ms  {2*b*M*K*N/ms/1e9:.1f} TFLOP/s", flush=True)
    Bt = torch.randn(b, N, K, device="cuda")
    ms = t(lambda: torch.bmm(A, Bt.transpose(1, 2), out=C))
    print(f"   (B transposed view)            : {ms:.3f} ms  {2*b*M*K*N/ms/1e9:.1f} TFLOP/s", flush=True)
    del A, B, C, Bt
