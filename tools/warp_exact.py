import sys; sys.path.insert(0, 'cardiac-segmentation-optical-flow_amd'); sys.path.insert(0, '.')
import torch, numpy as np
from cineflow import ops
from oracle import ops as OO, metrics as OM
dev = torch.device('cuda')
g = torch.Generator().manual_seed(5)
for (B, C, H, W, amp) in [(2, 4, 256, 256, 3.0), (3, 1, 40, 24, 5.0), (2, 2, 33, 47, 8.0), (1, 4, 256, 256, 60.0)]:
    flow = amp * torch.randn(B, 2, H, W, generator=g); src = torch.randn(B, C, H, W, generator=g)
    a = ops.warp_bilinear(flow.to(dev), src.to(dev)).cpu(); b = OO.warp_bilinear(flow, src)
    print('warp', (B, C, H, W), 'maxdiff', float((a - b).abs().max()), 'n_diff', int((a != b).sum()))
    v = ops.vecint(flow.to(dev), 7).cpu(); vb = OO.vecint(flow, 7)
    print('  vecint maxdiff', float((v - vb).abs().max()), int((v != vb).sum()))
    j = ops.jacobian_det(flow.to(dev)).cpu().numpy()
    jb = np.stack([OM.jacobian_determinant(flow[i].permute(1, 2, 0).numpy().astype(np.float64)) for i in range(B)])
    jb32 = np.stack([OM.jacobian_determinant(flow[i].permute(1, 2, 0).numpy()) for i in range(B)])
    print('  jac maxdiff f64-input', float(np.abs(j - jb).max()), int((j != jb).sum()), ' f32-input', float(np.abs(j - jb32).max()), j.dtype, jb32.dtype)
    lab = (torch.rand(B, H, W, generator=g) * 4).to(torch.uint8)
    wl = ops.warp_labels(flow[None].to(dev), lab.to(dev)).cpu(); wb = OO.warp_labels(flow[None], lab[:, None].float())[:, :, 0]
    print('  labels mismatches', int((wl.long() != wb).sum()))
