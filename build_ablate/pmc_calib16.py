"""does WRITE_SIZE / FETCH_SIZE count 16-byte-per-lane plain stores / loads like the 8-byte calibration kernels?  Known-size
device-to-device copies (rocclr copyBuffer: dwordx4 lanes) and a torch elementwise kernel (diagnostic)."""
import torch
n = 1 << 25          # 32 Mi doubles = 256 MiB
a = torch.randn(n, dtype=torch.float64, device='cuda')
b = torch.empty_like(a)
for _ in range(5):
    b.copy_(a)               # __amd_rocclr_copyBuffer
    c = a * 1.0000001        # vectorized elementwise kernel: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
print("known bytes per kernel: read", n * 8, "write", n * 8)
