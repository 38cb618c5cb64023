"""Host->device rate of one step's inputs (50 MB in six tensors from pinned memory) on 1 and 8 streams, no kernels running (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
B, N = 32, 20480
shapes = [(B, 3, N), (B, 1, N), (B, 3, N), (B, 3, 128), (B, 3, 128), (B, 3, 160, 512)]
for S in (1, 2, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    host = [[torch.empty(s, dtype=torch.float32).pin_memory() for s in shapes] for _ in range(S)]
    devt = [[torch.empty(s, dtype=torch.float32, device=dev) for s in shapes] for _ in range(S)]
    nbytes = sum(t.numel() * 4 for t in host[0])
    def run(n):
        for i in range(n):
            k = i % S
            with torch.cuda.stream(streams[k]):
                for h, d in zip(host[k], devt[k]):
                    d.copy_(h, non_blocking=True)
    run(2 * S); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 40; run(n); t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d stream(s): %.1f MB per step, %.2f ms per step, %.1f GB/s (host enqueue %.3f ms per step)" % (S, nbytes / 1e6, dt / n * 1e3, nbytes * n / dt / 1e9, t_enq / n * 1e3))
