"""Diagnostic: HBM write rate of a plain fill of the Hessian-sized output (upper bound for any Hessian kernel)."""
import torch, time
x = torch.empty(32 * 200 * 64 * 200 * 64, dtype=torch.float32, device="cuda:0")
for _ in range(2): x.zero_()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): x.zero_()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"fill {x.numel()*4/1e9:.2f} GB in {dt*1e3:.2f} ms = {x.numel()*4/dt/1e12:.2f} TB/s")
