"""float64 GEMM time of the learner's layers versus the batch size N (rocBLAS picks kernels by shape: look for cliffs)."""
import sys, time, torch
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda")
def bench(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for N in (46080, 51200, 61440, 66560):
    tot_f = tot_x = tot_w = 0.0
    line = f"N={N}:"
    for out, inn in ((2048, 657), (1024, 2048), (512, 1024), (105, 512)):
        x = torch.randn(N, inn, device=dev); gy = torch.randn(N, out, device=dev); w = torch.randn(out, inn, device=dev); b = torch.zeros(out, device=dev)
        fl = 2.0 * N * out * inn
        tf = bench(lambda: torch.addmm(b, x, w.t())); tx = bench(lambda: gy.mm(w))
        best = None
        for S in (4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 24, 32):
            if N % S: continue
            t = bench(lambda: torch.bmm(gy.view(S, N // S, out).transpose(1, 2), x.view(S, N // S, inn)).sum(0))
            if best is None or t < best[1]: best = (S, t)
        t8 = bench(lambda: torch.bmm(gy.view(8, N // 8, out).transpose(1, 2), x.view(8, N // 8, inn)).sum(0))
        line += f" [{out}x{inn}: fwd {fl/tf/1e9:.0f} dX {fl/tx/1e9:.0f} dW(S=8) {fl/t8/1e9:.0f} best S={best[0]} {fl/best[1]/1e9:.0f} TF]"
    print(line)
