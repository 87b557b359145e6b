import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
from ssg_amd import _lib
from ssg_amd._lib import check, ptr, stream
from ssg_amd.resnet import _h8l8, _weight_scale, pack_weight_khwc
L = _lib.lib(); dev = torch.device("cuda", 0)
B, H, W, Cin, Cout, k, stride, pad, use_res = 400, 16, 8, 256, 1024, 1, 1, 0, True
g = torch.Generator().manual_seed(B + Cin + Cout)
x = torch.randn(B, H, W, Cin, generator=g).to(dev)
w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
wk = pack_weight_khwc(w.permute(0, 2, 3, 1)); sc = _weight_scale(wk); ws = _h8l8(wk * sc).to(dev)
bias = torch.randn(Cout, generator=g).to(dev)
xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")
OH, OW = H, W
r = torch.randn(B, OH, OW, Cout, generator=g).to(dev)
rs = torch.empty_like(r); check(L.ssg_h8l8_encode(ptr(r), ptr(rs), r.numel(), 1.0, stream()), "enc")
def run(lo, hi, res=True, flags=3):
    out = torch.empty(hi - lo, OH, OW, Cout, device=dev)
    check(L.ssg_conv2d_nhwc_x(ptr(xs[lo:hi]), ptr(ws), ptr(bias), ptr(rs[lo:hi]) if res else None, ptr(out), hi - lo, H, W, Cin, Cout, k, k, stride, pad, 1, flags, 1.0 / sc, None, None, stream()), "convx")
    return out
big = run(0, B); big2 = run(0, B)
print("deterministic:", torch.equal(big.view(torch.int32), big2.view(torch.int32)))
for step in (64, 16, 100, 200):
    for lo in range(0, B, step):
        hi = min(lo + step, B)
        s = run(lo, hi)
        ne = (s.view(torch.int32) != big[lo:hi].view(torch.int32))
        if ne.any():
            idx = ne.nonzero()
            print("step %d [%d,%d): %d words differ; first %r; channels %r" % (step, lo, hi, int(ne.sum()), idx[0].tolist(), sorted(set(idx[:, 3].tolist()))[:16]))
        else:
            print("step %d [%d,%d): equal" % (step, lo, hi))
# no residual
bn = run(0, B, res=False); 
for lo in range(0, B, 64):
    hi = min(lo + 64, B); s = run(lo, hi, res=False)
    print("nores [%d,%d): %s" % (lo, hi, torch.equal(s.view(torch.int32), bn[lo:hi].view(torch.int32))))
# which one is right: decode and compare with fp64
def dec(t):
    d = torch.empty_like(t); check(L.ssg_h8l8_decode(ptr(t.contiguous()), ptr(d), d.numel(), 1.0, stream()), "dec"); return d
lo, hi = 384, 400
ref = torch.relu(torch.nn.functional.conv2d(x[lo:hi].cpu().permute(0, 3, 1, 2).double(), w.double(), bias.cpu().double()) + r[lo:hi].cpu().permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
a = dec(big[lo:hi]).cpu().double(); b = dec(run(lo, hi)).cpu().double()
print("err big %.3g small %.3g ; max |big-small| %.3g" % ((a - ref).abs().max(), (b - ref).abs().max(), (a - b).abs().max()))
print("=== part 2")
bigs = [run(0, B) for _ in range(4)]
for i in range(1, 4):
    ne = bigs[i].view(torch.int32) != bigs[0].view(torch.int32)
    print("big run %d vs 0: %d words differ" % (i, int(ne.sum())))
smalls = [torch.cat([run(lo, min(lo + 16, B)) for lo in range(0, B, 16)], 0) for _ in range(3)]
for i in range(1, 3):
    print("small(16) run %d vs 0: %d words differ" % (i, int((smalls[i].view(torch.int32) != smalls[0].view(torch.int32)).sum())))
print("big0 vs small0: %d" % int((bigs[0].view(torch.int32) != smalls[0].view(torch.int32)).sum()))
# magnitude against fp64 on the whole tensor, in chunks
def full_err(t):
    worst = 0.0
    for lo in range(0, B, 50):
        hi = min(lo + 50, B)
        ref = torch.relu(torch.nn.functional.conv2d(x[lo:hi].permute(0, 3, 1, 2).double(), w.to(dev).double(), bias.double()) + r[lo:hi].permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
        worst = max(worst, float((dec(t[lo:hi]).double() - ref).abs().max()))
    return worst
print("max err vs fp64: big0 %.3g big1 %.3g small0 %.3g" % (full_err(bigs[0]), full_err(bigs[1]), full_err(smalls[0])))
ne = bigs[1].view(torch.int32) != bigs[0].view(torch.int32)
if ne.any():
    idx = ne.nonzero()
    print("example diffs:", idx[:5].tolist(), dec(bigs[0])[ne][:8].tolist(), dec(bigs[1])[ne][:8].tolist())
