"""Errors of the fused atomic networks (mlp_fused.hip) against a float64 evaluation, config 2's shapes, for several activation scales:
energy (relative to the largest) and dE/dAEV (relative to the largest component).   python tools/mlp_split_error.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_mlp_fused_gpu as T
from nnpops_amd.capi import FusedMLP

gen = torch.Generator().manual_seed(0)
n = 2001
x = torch.rand((n, 1008), generator=gen) * (torch.rand((n, 1008), generator=gen) < 0.3)
species = torch.tensor([3, 0, 0] * 667)
kinds = []
for s, widths in ((0, (256, 192, 160)), (3, (192, 160, 128))):
    kd = T._networks(widths, 8, 1008, seed=10 + s)
    kd["atoms"] = torch.nonzero(species == s).flatten().to(torch.int32)
    kinds.append(kd)
e_ref, dx_ref = T._host_reference(kinds, x)
# the same in float32 on the device (what a plain fp32 implementation reaches)
xs = x.to("cuda").requires_grad_(True)
outs = []
for kd in kinds:
    xa = xs[kd["atoms"].long().cuda()]
    pm = []
    for m in range(8):
        y = torch.nn.functional.celu(xa @ kd["w0"][m].cuda().t() + kd["b0"][m].cuda(), alpha=0.1)
        y = torch.nn.functional.celu(y @ kd["w2"][m].cuda().t() + kd["b2"][m].cuda(), alpha=0.1)
        y = torch.nn.functional.celu(y @ kd["w4"][m].cuda().t() + kd["b4"][m].cuda(), alpha=0.1)
        pm.append(y @ kd["w6"][m].cuda() + kd["b6"][m].cuda())
    outs.append(torch.stack(pm, dim=1))
e32 = torch.cat(outs, 0)
e32.sum().backward()
print("torch fp32      energy %.2e  dx %.2e" % (float((e32.detach().cpu().double() - e_ref).abs().max() / e_ref.abs().max()),
                                                float((xs.grad.cpu().double() - dx_ref).abs().max() / dx_ref.abs().max())))
for k in (4, 6, 8, 10):
    kinds_dev = [{kk: v.to("cuda") for kk, v in kd.items()} for kd in kinds]
    mlp = FusedMLP(kinds_dev, 1008, act_scale_log2=k)
    xd = x.to("cuda").contiguous()
    e = mlp.forward(xd, with_gradient=True).clone()
    dx = mlp.input_grad(xd)
    print("fused, scale 2^-%-2d energy %.2e  dx %.2e" % (k, float((e.cpu().double() - e_ref).abs().max() / e_ref.abs().max()),
                                                        float((dx.cpu().double() - dx_ref).abs().max() / dx_ref.abs().max())))
