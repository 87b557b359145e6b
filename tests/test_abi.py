"""CPU suite: the C-ABI library loads, exports every symbol include/ssg_hip.h declares, and
its argument validation / host-only helpers work without a GPU (no compute is launched)."""
import ctypes
import math
import os

import numpy as np
import pytest

import ssg_amd
from ssg_amd import _lib


def test_library_is_built_and_exports_header_symbols():
    assert os.path.exists(_lib.SO_PATH), "run __graft_entry__.build() first"
    protos = _lib.parse_header()
    assert len(protos) >= 25
    raw = ctypes.CDLL(_lib.SO_PATH)
    for name in protos:
        assert hasattr(raw, name), "libssg_hip.so does not export %s" % name
    assert _lib.lib().ssg_version() >= 100


def test_host_helpers():
    L = _lib.lib()
    for x in [0.9, 0.7, 1.0, 0.0, 1e-8, 65504.0, 65520.0, 1e9, -0.3, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 0.1 + 0.2]:
        assert L.ssg_double_to_half_bits(x) == np.float64(x).astype(np.float16).view(np.uint16), x
    rng = np.random.default_rng(0)
    for x in rng.standard_normal(2000) * 10.0 ** rng.integers(-9, 6, 2000):
        assert L.ssg_double_to_half_bits(float(x)) == np.float64(x).astype(np.float16).view(np.uint16)
    assert L.ssg_krecip_row_capacity(20) == 21 * 12       # (k1+1) * (round(k1/2)+2)
    assert L.ssg_krecip_row_capacity(5) == 6 * 4           # np.around(2.5) == 2


def test_argument_validation_without_gpu():
    L = _lib.lib()
    assert L.ssg_topk_rank(None, None, 10, 10, 65, None, None) == -1
    assert b"K" in L.ssg_last_error()
    assert L.ssg_sqdist_self_f16(None, None, 8, 6, 0, 8, 0, None, None, None) == -1     # d % 4 != 0
    assert L.ssg_sort_u64(None, 1000, None) == -1
    assert L.ssg_krecip(None, None, None, 100, 0, 100, 21, 20, 10, None, None, None, None) == -1   # cap too small
    assert L.ssg_eps_hist(None, None, 10, 0, 10, 0, 0.1, 0, 51, 12, 1, None, None) == -1
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    # entry points added with the split-half embedding, the int8 Gram, the evaluation step and the kNN-set variant
    assert L.ssg_conv2d_nhwc_x(None, None, None, None, None, 2, 8, 8, 48, 64, 1, 1, 1, 0, 1, 3, 1.0, None, None, None) == -1          # Cin % 32
    assert b"unsupported shape" in L.ssg_last_error()
    assert L.ssg_conv1x1_dual_nhwc_x(None, None, None, None, None, 2, 8, 8, 64, 4, 4, 64, 2, 96, 1, 3, 1.0, None, None, None) == -1   # Cout % 64, grid
    assert L.ssg_h8l8_encode(None, None, 12, 1.0, None) == -1 and L.ssg_h8l8_decode(None, None, 0, 1.0, None) == -1       # n % 8
    assert L.ssg_gram_i8_encode(None, 4, 64, 5, None, None, None, None) == -1                                             # digits must be 3 or 4
    assert L.ssg_gram_i8_encode(None, 4, 20000, 3, None, None, None, None) == -1                                          # d > 16384 (int32 headroom)
    assert L.ssg_sqdist_self_i8(None, None, 8, 64, 3, 4, 8, 0, None, None, None, None) == -1                                 # row block outside N
    assert L.ssg_gram_i8_encoded_bytes(10, 70, 3) == 64 * 3 * 32 * 3                                                      # one 64-row panel, 3 k blocks of 32, 3 digits
    assert L.ssg_source_rowmin_filtered(None, None, 8, 100, 100, 64, 1e-3, 0.0, 0.0, None, None, None) == -1              # Ns_pad % 128
    assert L.ssg_rank_metrics(None, 4, 10, 8, None, None, None, None, 0, None, None, None, None) == -1                     # ld < n
    assert L.ssg_knn_sets(None, None, 10, 0, 10, 11, 64, None, None, None, None, None) == -1                              # K > N
    assert L.ssg_set_jaccard_rows(None, None, 0, None, None, 10, 0, 10, 0, None, None) == -1


def test_no_cpu_fallback(monkeypatch):
    """The product path must fail loudly when the HIP extension is missing."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libssg_hip.so")
    with pytest.raises(ssg_amd.SSGError):
        _lib.lib()


def test_dbscan_parameter_errors():
    from ssg_amd import DBSCAN
    for bad in (float("nan"), -1.0, 0.0, "x"):
        with pytest.raises(ValueError):
            DBSCAN(eps=bad, min_samples=4, metric="precomputed").fit(np.zeros((2, 2)))
    with pytest.raises(ValueError):
        DBSCAN(eps=0.5, min_samples=4, metric="euclidean").fit(np.zeros((2, 2)))


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "ssg_oracle" not in txt.replace("oracle/ssg_oracle.c", ""), f


def test_split_half_host_formats():
    """Host-side encoders of the split-half embedding path (ssg_amd/resnet.py): layout and value properties that the HIP
    kernels rely on (include/ssg_hip.h: h8l8 / h4l4, reduction order of the packed weights)."""
    import torch
    from ssg_amd.resnet import _h4l4, _h8l8, _weight_scale, pack_weight_khwc
    g = torch.Generator().manual_seed(0)
    v = torch.randn(3, 32, generator=g) * torch.tensor([1e-3, 1.0, 300.0]).view(3, 1)
    enc = _h8l8(v)
    assert enc.dtype == torch.float32 and enc.shape == v.shape                       # same 4 bytes per value, same addressing
    halves = enc.view(torch.float16).view(3, 4, 2, 8)                                # [row][group of 8][hi | lo][8]
    hi, lo = halves[:, :, 0, :].reshape(3, 32), halves[:, :, 1, :].reshape(3, 32)
    assert torch.equal(hi, v.half())                                                 # hi = half(v)
    assert torch.equal(lo, (v - hi.float()).half())                                  # lo = half(v - hi)
    rec = hi.float() + lo.float()
    assert ((rec - v).abs() <= 2.0 ** -21 * v.abs() + 2.0 ** -24).all()              # 22 significand bits, absolute floor 2^-24
    t4 = _h4l4(v[:, :8]).view(torch.float16).view(3, 2, 2, 4)                        # stem: per tap [hi4 | lo4]
    assert torch.equal(t4[:, :, 0, :].reshape(3, 8), v[:, :8].half())
    # weight scale: a power of two that keeps every scaled weight inside the half range
    for mx in (1e-4, 0.03, 1.0, 90.0, 3000.0):
        sc = _weight_scale(torch.tensor([mx, -mx / 3]))
        assert sc <= 256.0 and mx * sc <= 16384.0 and math.log2(sc) == int(math.log2(sc))
    # reduction order k = ((c // 32) * KH*KW + r*KW + s) * 32 + c % 32
    cout, kh, kw, cin = 2, 3, 3, 64
    w = torch.arange(cout * kh * kw * cin, dtype=torch.float32).view(cout, kh, kw, cin)
    pk = pack_weight_khwc(w)
    for (o, r, s, c) in ((0, 0, 0, 0), (1, 2, 1, 37), (0, 1, 2, 63), (1, 0, 0, 32)):
        k = ((c // 32) * kh * kw + r * kw + s) * 32 + c % 32
        assert pk[o, k] == w[o, r, s, c]


def test_comm_entry_points_validate_without_gpu():
    """the collective entry points of the ABI refuse bad arguments before touching RCCL"""
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.ssg_comm_unique_id(None) == -1
    assert L.ssg_comm_init(ctypes.byref(h), 0, 0, (ctypes.c_ubyte * 128)()) == -1 and b"world" in L.ssg_last_error()
    assert L.ssg_comm_init(ctypes.byref(h), 2, 2, (ctypes.c_ubyte * 128)()) == -1
    assert L.ssg_allgather(None, None, None, 16, None) == -1
    assert L.ssg_allreduce_sum_i64(None, None, 4, None) == -1
    assert L.ssg_comm_destroy(None) == 0


def test_one_rccl_build_per_process():
    """libssg_hip.so must not bring a second RCCL into a process that runs torch.distributed -- it has no NEEDED
    entry for librccl, and its collectives bind (lazily) to the RCCL torch already maps."""
    import subprocess
    import sys
    import torch  # noqa: F401  (maps torch/lib/librccl.so)
    out = subprocess.run(["readelf", "-d", _lib.SO_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in out, "libssg_hip.so links RCCL again"
    code = ("import sys; sys.path.insert(0, %r); import torch, ssg_amd; from ssg_amd import _lib; L = _lib.lib(); "
            "p = L.ssg_comm_library().decode(); "
            "maps = sorted({l for l in open('/proc/self/maps').read().split() if 'librccl' in l}); print(p); print(len(maps)); print(maps[0] if maps else '')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    bound, nmaps, mapped = r.stdout.strip().splitlines()[-3:]
    assert int(nmaps) == 1, "two RCCL builds mapped: %s" % r.stdout
    assert os.path.realpath(bound) == os.path.realpath(mapped) and "torch" in bound, (bound, mapped)


@pytest.mark.gpu
def test_comm_entry_points_world1_on_rccl():
    """ssg_comm_unique_id -> ssg_comm_init -> ssg_allgather / ssg_allreduce_sum_i64 -> ssg_comm_destroy on a one-rank RCCL
    communicator (the test box has one GPU): the entry points a non-torch host binds run end to end on the device."""
    import torch
    from ssg_amd.dist import AbiComm
    torch.cuda.set_device(0)
    comm = AbiComm(1, 0, AbiComm.unique_id())
    assert "torch" in _lib.lib().ssg_comm_library().decode()      # bound to the RCCL torch maps, not to a second copy
    t = torch.arange(40, dtype=torch.float32, device="cuda").view(10, 4)
    assert torch.equal(comm.all_gather_rows(t), t)
    hh = torch.arange(7, dtype=torch.int64, device="cuda")
    assert torch.equal(comm.all_reduce_sum_(hh.clone()), hh)
    torch.cuda.synchronize()
    comm.destroy()
