import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
from ssg_amd import _lib
if os.environ.get("SSG_LIB_PATH"):
    _lib.SO_PATH = os.environ["SSG_LIB_PATH"]
    import ctypes
    L = ctypes.CDLL(_lib.SO_PATH)
    for name, (res, args) in _lib.parse_header(os.environ.get("SSG_HDR", _lib.HEADER)).items():
        if hasattr(L, name):
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
    _lib._lib = L
from ssg_amd._lib import check, ptr, stream
from ssg_amd.resnet import _h8l8, _weight_scale, pack_weight_khwc
L = _lib.lib(); dev = torch.device("cuda", 0)
B, H, W, Cin, Cout, k = 400, 16, 8, 256, 1024, 1
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, W, Cin, generator=g).to(dev)
w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
wk = pack_weight_khwc(w.permute(0, 2, 3, 1)); sc = _weight_scale(wk); ws = _h8l8(wk * sc).to(dev)
bias = torch.randn(Cout, generator=g).to(dev)
xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")
r = torch.randn(B, H, W, Cout, generator=g).to(dev)
rs = torch.empty_like(r); check(L.ssg_h8l8_encode(ptr(r), ptr(rs), r.numel(), 1.0, stream()), "enc")
zs = torch.zeros_like(rs)
def run(res, flags=3, relu=1):
    out = torch.empty(B, H, W, Cout, device=dev)
    check(L.ssg_conv2d_nhwc_x(ptr(xs), ptr(ws), ptr(bias), ptr(res), ptr(out), B, H, W, Cin, Cout, k, k, 1, 0, relu, flags, 1.0 / sc, None, None, stream()), "convx")
    return out
def ndiff(f, n=4):
    outs = [f() for _ in range(n)]
    return [int((o.view(torch.int32) != outs[0].view(torch.int32)).sum()) for o in outs[1:]]
print("lib", _lib.SO_PATH)
print("nores            :", ndiff(lambda: run(None)))
print("res split        :", ndiff(lambda: run(rs)))
print("res split norelu :", ndiff(lambda: run(rs, relu=0)))
print("res fp32 out fp32:", ndiff(lambda: run(r, flags=1)))
print("res = zeros      :", ndiff(lambda: run(zs)), "equal to nores:", torch.equal(run(zs).view(torch.int32), run(None).view(torch.int32)))
print("=== pattern")
def dec(t):
    d = torch.empty_like(t); check(L.ssg_h8l8_decode(ptr(t.contiguous()), ptr(d), d.numel(), 1.0, stream()), "dec"); return d
a = run(None); 
for rep in range(2):
    b = run(zs)
    da, db = dec(a).view(-1, Cout), dec(b).view(-1, Cout)
    ne = (da != db)
    idx = ne.nonzero()
    print("rep", rep, "decoded values differ:", int(ne.sum()))
    if len(idx):
        ms = idx[:, 0]; cs = idx[:, 1]
        print("  rows(m) %% 128 histogram of first 20:", (ms[:20] % 128).tolist())
        print("  m//128 (tile) of first 20:", (ms[:20] // 128).tolist())
        print("  cols first 20:", cs[:20].tolist())
        print("  nores vals:", da[ne][:10].tolist())
        print("  zres  vals:", db[ne][:10].tolist())
        import collections
        print("  tiles hit:", len(set((ms // 128).tolist())), "col tiles(256):", collections.Counter((cs // 256).tolist()), "wn(64):", collections.Counter(((cs % 256) // 64).tolist()), "j:", collections.Counter(((cs % 64) // 32).tolist()))
        print("  m%32:", collections.Counter((ms % 32).tolist()).most_common(8), " (m%128)//32:", collections.Counter(((ms % 128) // 32).tolist()))
        print("  c%8:", collections.Counter((cs % 8).tolist()))
