"""forward_warp on a torn field (i.i.d. U(-50, 50) px, 16 x 768 x 1024) a few times — for `rocprofv3 --kernel-trace --stats`:
which of the tile / bin / scan / fill / gather kernels carries the time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import check, ptr, stream
dev = torch.device("cuda:0")
N, H, W = 16, 768, 1024
g = torch.Generator().manual_seed(0)
fl = (torch.rand(N, H, W, 2, generator=g) * 100 - 50).to(dev)
lib = _lib.lib()
ws = torch.empty(lib.unflow_forward_warp_workspace_bytes(N, H, W, 1) // 4 + 64, dtype=torch.float32, device=dev)
out = torch.empty(N, H, W, 1, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    check(lib.unflow_forward_warp_fwd(ptr(fl), ptr(out), N, H, W, 1, ptr(ws), _lib.csz(ws.numel() * 4), stream()), "fw")
torch.cuda.synchronize()
print("done", out.sum().item())
