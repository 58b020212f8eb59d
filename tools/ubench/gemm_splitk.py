"""Weight-gradient GEMMs of the PPO update (dW = dY^T X with K = batch) at float64: plain mm vs batched split-K."""
import sys
import time

import torch

torch.set_default_dtype(torch.float64)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
dev = torch.device("cuda")


def bench(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for out, inn in ((2048, 657), (1024, 2048), (512, 1024), (105, 512), (1, 512)):
    x = torch.randn(N, inn, device=dev)
    gy = torch.randn(N, out, device=dev)
    w = torch.randn(out, inn, device=dev)
    fl = 2.0 * N * out * inn
    t_plain = bench(lambda: gy.t().mm(x))
    line = f"dW {out}x{inn} K={N}: plain {t_plain:.2f} ms ({fl / t_plain / 1e9:.1f} TF)"
    for S in (4, 8, 16, 32):
        if N % S:
            continue
        t = bench(lambda: torch.bmm(gy.view(S, N // S, out).transpose(1, 2), x.view(S, N // S, inn)).sum(0))
        line += f" | S={S}: {t:.2f} ms ({fl / t / 1e9:.1f} TF)"
    print(line)
    t_f = bench(lambda: torch.addmm(w.new_zeros(out), x, w.t()))
    t_b = bench(lambda: gy.mm(w))
    print(f"   fwd {t_f:.2f} ms ({fl / t_f / 1e9:.1f} TF)  dX {t_b:.2f} ms ({fl / t_b / 1e9:.1f} TF)")
