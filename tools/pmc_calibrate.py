"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this GPU:
a 256 MiB device-to-device copy (16 B/lane loads and stores) and a 256 MiB fill (stores only).
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes)."""
import torch
n = 256 * 1024 * 1024 // 4
src = torch.arange(n, dtype=torch.float32, device='cuda')
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(src)          # reads 256 MiB, writes 256 MiB
    dst.fill_(1.0)          # writes 256 MiB
torch.cuda.synchronize()
