"""PMC calibration workload: one streaming copy of a known size (256 MiB read + 256 MiB
written) so FETCH_SIZE / WRITE_SIZE can be scaled to bytes (MI355X_MICROARCH.md: on gfx950
FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE is uncalibrated)."""
import torch
x = torch.ones(64 * 2 ** 20, dtype=torch.float32, device='cuda')
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
print(float(y[123]))
