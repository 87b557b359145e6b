"""The scalar numerics rules of the device code, executed on the host and compared with numpy.

csrc/ssg_common.h (numpy's half arithmetic, the one-rounding double -> half, final_dist from its compact form) and the integer
half(sqrt(.)) rounding of the int8 Gram epilogue (csrc/gram_i8.hip) are scalar C++ behind `__device__`.  tools/hostexec/scalar_rules.cpp
maps that to host functions and compiles the SAME SOURCE TEXT for x86 (ROCm's clang, host only); this file checks it bit for bit against
numpy -- the arithmetic the reference runs (reid/rerank.py:33-122 in np.float16, scipy's float64 cdist + sqrt).  No GPU.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def hx(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ not present")
    d = tmp_path_factory.mktemp("hx")
    csrc = os.path.join(ROOT, "self-similarity-grouping_amd", "csrc")

    def cut(fname, start, stop=None):
        """the lines of csrc/<fname> from the one that starts with `start` to the function's closing brace (or up to `stop`)"""
        out = []
        for line in open(os.path.join(csrc, fname)):
            if not out and not line.startswith(start):
                continue
            if stop is not None and stop in line:
                break
            out.append(line)
            if stop is None:
                code = line.split("//")[0].rstrip()
                if (len(out) == 1 and code.endswith(("}", ";"))) or code in ("}", "};"):
                    break
        assert out, (fname, start)
        return "".join(out)

    text = [cut("gram_i8.hip", "__device__ __forceinline__ long long half_units24", stop="halfwave_max"),
            cut("jaccard.hip", "__device__ __forceinline__ hbits jaccard_scaled"),
            cut("krecip.hip", "__device__ float pairwise_sum_f32"),
            "template <typename T>\n" + cut("cluster.hip", "__device__ T pw_leaf"),
            cut("cluster.hip", "__device__ __forceinline__ int sur_bin(float x)"),
            cut("cluster.hip", "__device__ __forceinline__ float sur_bin_upper"),
            cut("topk_intro.hip", "constexpr uint32_t KEY_NAN"),
            cut("topk_intro.hip", "__device__ __forceinline__ uint32_t norm_key"),
            "template <int NSLOT>\n" + cut("cluster.hip", "__device__ __forceinline__ int ss_bucket")]
    assert "sqrt_units48_to_half" in text[0]
    (d / "rules_cut.inc").write_text("\n".join(text))
    (d / "preprocess_cut.inc").write_text(cut("preprocess.hip", "__global__ __launch_bounds__(256) void resize_h_u8_kernel") + "\n" +
                                          cut("preprocess.hip", "__global__ __launch_bounds__(256) void resize_v_normalize_kernel"))
    (d / "cc_cut.inc").write_text(cut("cluster.hip", "// ------------------------------------------------------------------ K12 union-find", stop="}  // namespace ssg"))
    (d / "split_cut.inc").write_text(cut("conv.hip", "__device__ __forceinline__ unsigned pack_h2", stop="struct ConvParams"))
    (d / "pool_cut.inc").write_text(cut("conv.hip", "__device__ __forceinline__ void h8l8_load8", stop="// out = (a + b) / ||a + b||_2 per row"))
    so = str(d / "libhx.so")
    r = subprocess.run([CLANG, "-x", "hip", "--offload-host-only", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", "-I" + str(d), "-o", so,
                        os.path.join(ROOT, "tools", "hostexec", "scalar_rules.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _interesting_doubles(rng, n):
    """random doubles over the whole half range + everything around the half grid: values, midpoints, their float64 neighbours"""
    hb = np.arange(0, 0x7C00, dtype=np.uint16)
    hv = hb.view(np.float16).astype(np.float64)
    mid = (hv[:-1] + hv[1:]) / 2
    grid = np.concatenate([hv, mid, np.nextafter(mid, np.inf), np.nextafter(mid, -np.inf), np.nextafter(hv, np.inf), np.nextafter(hv, -np.inf)])
    x = np.concatenate([grid, -grid, rng.standard_normal(n) * np.exp(rng.uniform(-30, 12, n)), rng.uniform(0, 4, n),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65519.999, 65520.0, 1e300, 5e-324, 2.0 ** -25, 2.0 ** -24 * 1.5])])
    return np.ascontiguousarray(x)


def test_double_to_half_is_numpys_single_rounding(hx):
    """d2h (ssg_common.h) == numpy's float64 -> float16 cast (npy_double_to_half: ONE round-to-nearest-even, subnormals, overflow to inf)"""
    x = _interesting_doubles(np.random.default_rng(0), 2_000_000)
    out = np.empty(x.size, np.uint16)
    hx.hx_d2h(_p(x), ctypes.c_long(x.size), _p(out))
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    nan = np.isnan(x)
    assert np.array_equal(out[~nan], ref[~nan])
    assert np.all((out[nan] & 0x7FFF) > 0x7C00)


def test_half_binary_ops_are_numpys(hx):
    """h_add / h_sub / h_mul / h_div == numpy's float16 loops (float32 op + one rounding) on every class of operand"""
    rng = np.random.default_rng(1)
    n = 4_000_000
    a = rng.integers(0, 1 << 16, n).astype(np.uint16); b = rng.integers(0, 1 << 16, n).astype(np.uint16)
    a[:2 * 0x7C00] = np.repeat(np.arange(0, 0x7C00, dtype=np.uint16), 2)            # every non-negative finite half, twice
    af, bf = a.view(np.float16), b.view(np.float16)
    for op, fn in enumerate((np.add, np.subtract, np.multiply, np.divide)):
        out = np.empty(n, np.uint16)
        hx.hx_binop(op, _p(a), _p(b), ctypes.c_long(n), _p(out))
        with np.errstate(all="ignore"):
            ref = fn(af, bf).view(np.uint16)
        nan = np.isnan(ref.view(np.float16))
        assert np.array_equal(out[~nan], ref[~nan]), fn
        assert np.all((out[nan] & 0x7FFF) > 0x7C00), fn


def test_float_half_conversions(hx):
    rng = np.random.default_rng(2)
    with np.errstate(over="ignore"):
        x = np.concatenate([rng.standard_normal(2_000_000).astype(np.float32) * np.exp(rng.uniform(-20, 12, 2_000_000)).astype(np.float32),
                            _interesting_doubles(rng, 10).astype(np.float32)])
    x = np.ascontiguousarray(x[~np.isnan(x)])
    out = np.empty(x.size, np.uint16)
    hx.hx_f2h(_p(x), ctypes.c_long(x.size), _p(out))
    with np.errstate(over="ignore"):
        assert np.array_equal(out, x.astype(np.float16).view(np.uint16))
    hb = np.arange(0, 1 << 16, dtype=np.uint16)
    f = np.empty(hb.size, np.float32)
    hx.hx_h2f(_p(hb), ctypes.c_long(hb.size), _p(f))
    ref = hb.view(np.float16).astype(np.float32)
    assert np.array_equal(f.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])


def test_final_dist_from_compact_form(hx):
    """final_dist_value == reid/rerank.py:122 `jaccard*(1-lambda) + original*lambda` as the reference evaluates it: J' is the half product
    the matrix stores, the source term half(v_i + v_k) is promoted to float64 and scaled by the python float lambda"""
    rng = np.random.default_rng(3)
    n = 2_000_000
    jp = rng.uniform(0, 1, n).astype(np.float16); vi = rng.uniform(0, 1, n).astype(np.float16); vk = rng.uniform(0, 1, n).astype(np.float16)
    for lam in (0.3, 0.0, 1.0, 0.1):
        out = np.empty(n, np.float64)
        hx.hx_final_dist(_p(jp.view(np.uint16)), _p(vi.view(np.uint16)), _p(vk.view(np.uint16)), ctypes.c_double(lam), ctypes.c_long(n), _p(out))
        ref = jp.astype(np.float64) + (vk + vi).astype(np.float64) * lam
        assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)), lam


def test_integer_sqrt_rounding_is_half_of_float64_sqrt(hx):
    """sqrt_units48_to_half(U) == float16(sqrt(float64(U * 2^-48))) -- scipy cdist's float64 root, then rerank.py:33's cast -- for
    U < 2^50 (squared distances below 4: every pair of L2-normalised features), whatever the 1-ulp error of the device's approximate
    square root does to the candidate (modes -1 / 0 / +1): random U, every half midpoint square +- 3, every half value squared +- 3."""
    rng = np.random.default_rng(4)
    hb = np.arange(0, 0x4000, dtype=np.uint32)                   # halves below 2.0
    units = np.empty(hb.size, np.int64)
    hx.hx_units24(_p(hb), ctypes.c_long(hb.size), _p(units))
    assert np.array_equal(units, np.round(hb.astype(np.uint16).view(np.float16).astype(np.float64) * 2.0 ** 24).astype(np.int64))
    mid2 = ((units[:-1] + units[1:]) ** 2) // 4                  # (midpoint in units of 2^-25)^2 / 4 = U of the midpoint's square (floor)
    sq = units ** 2
    near = np.concatenate([(mid2[:, None] + np.arange(-3, 4)[None, :]).ravel(), (sq[:, None] + np.arange(-3, 4)[None, :]).ravel()])
    u = np.concatenate([near, rng.integers(0, 1 << 50, 2_000_000), rng.integers(0, 1 << 30, 500_000), rng.integers(0, 1 << 12, 5000), np.array([0, 1, 2, (1 << 50) - 1])])
    u = np.ascontiguousarray(u[(u >= 0) & (u < (1 << 50))].astype(np.int64))
    ref = np.sqrt(u.astype(np.float64) * 2.0 ** -48).astype(np.float16).view(np.uint16)
    for mode in (0, 1, -1):
        out = np.empty(u.size, np.uint16)
        hx.hx_sqrt48(_p(u), ctypes.c_long(u.size), ctypes.c_int(mode), _p(out))
        bad = np.flatnonzero(out != ref)
        assert bad.size == 0, (mode, bad[:5], u[bad[:5]], out[bad[:5]], ref[bad[:5]])


def test_jaccard_scaled_is_the_references_half_expression(hx):
    """jaccard_scaled(t, half(1-lambda)) == rerank.py:120-122 in np.float16: 1 - t/(2 - t), negatives to 0, times the python float 1-lambda
    (a weak scalar: the product is a half op) -- for EVERY half t in [0, 2] and a range of lambda"""
    t = np.arange(0, 0x4001, dtype=np.uint16)
    tf = t.view(np.float16)
    for lam in (0.3, 0.0, 0.1, 0.5, 0.9, 1.0):
        om = np.float16(1.0 - lam)
        out = np.empty(t.size, np.uint16)
        hx.hx_jaccard_scaled(_p(t), ctypes.c_uint16(int(om.view(np.uint16))), ctypes.c_long(t.size), _p(out))
        with np.errstate(all="ignore"):
            j = np.float16(1) - tf / (np.float16(2) - tf)
            j[j < 0] = 0
            ref = (j * (1.0 - lam)).astype(np.float16)       # float16 array * python float
        ok = ~np.isnan(ref)
        assert ref.dtype == np.float16 and np.array_equal(out[ok], ref.view(np.uint16)[ok]), lam


def test_pairwise_sums_are_numpys(hx):
    """pairwise_sum_f32 (the V row normalisation, rerank.py:91) and the leaves of the eps mean (selftraining.py:293) add in numpy's
    pairwise order: equal to np.sum / np.add.reduce bit for bit for every length that occurs (V rows <= a few hundred entries; leaves <= 128)"""
    rng = np.random.default_rng(6)
    hx.hx_pairwise_sum_f32.restype = ctypes.c_float
    hx.hx_pw_leaf_f64.restype = ctypes.c_double
    hx.hx_pw_leaf_f32.restype = ctypes.c_float
    for n in list(range(0, 300)) + [511, 512, 1000, 4097]:
        a = np.exp(-rng.uniform(0, 3, n)).astype(np.float32)
        got = hx.hx_pairwise_sum_f32(_p(a), ctypes.c_int(n))
        assert np.float32(got).view(np.uint32) == np.add.reduce(a, dtype=np.float32).view(np.uint32), n
    for n in range(0, 129):
        x = rng.uniform(0, 2, n + 3)
        keys = np.ascontiguousarray(x.view(np.uint64))
        got = hx.hx_pw_leaf_f64(_p(keys), ctypes.c_longlong(3), ctypes.c_int(n))
        assert np.float64(got).view(np.uint64) == np.add.reduce(x[3:]).view(np.uint64), n
        got32 = hx.hx_pw_leaf_f32(_p(keys), ctypes.c_longlong(3), ctypes.c_int(n))
        assert np.float32(got32).view(np.uint32) == np.add.reduce(x[3:].astype(np.float32), dtype=np.float32).view(np.uint32), n


def test_rank_key_and_surrogate_bins(hx):
    """norm_key = the bit pattern of half(D / rowmax) (rerank.py:68: original_dist / max in np.float16; non-negative halves order like
    their bits, NaN last like numpy's sort); the eps rule's surrogate bins are monotone and their upper edges bound their members."""
    raw = np.arange(0, 0x7C01, dtype=np.uint32)
    d = raw.astype(np.uint16).view(np.float16)
    for mx in (np.float16(1.0), np.float16(0.37), np.float16(3.998), np.float16(6e-5), np.float16(0.0)):
        out = np.empty(raw.size, np.uint32)
        hx.hx_norm_key(_p(raw), ctypes.c_float(float(mx)), ctypes.c_long(raw.size), _p(out))
        with np.errstate(all="ignore"):
            ref = (d / mx)
        nan = np.isnan(ref)
        assert np.array_equal(out[~nan], ref.view(np.uint16)[~nan].astype(np.uint32)) and np.all(out[nan] == 0xFFFF), mx
    x = np.sort(np.concatenate([np.random.default_rng(7).uniform(0, 4.5, 1_000_000), np.exp(np.random.default_rng(8).uniform(-30, 2, 200_000)), [0.0]])).astype(np.float32)
    b = np.empty(x.size, np.int32)
    hx.hx_sur_bin(_p(x), ctypes.c_long(x.size), _p(b))
    assert np.all(np.diff(b) >= 0) and b.min() == 0 and b.max() == 4095
    hx.hx_sur_bin_upper.restype = ctypes.c_float
    up = np.array([hx.hx_sur_bin_upper(int(k)) for k in range(4096)], np.float32)
    inside = b < 4095
    assert np.all(x[inside] < up[b[inside]]) and np.all(np.diff(up) > 0)
    lo = b > 0
    assert np.all(x[lo] >= up[b[lo] - 1])


def test_sample_sort_bucket_rule(hx):
    """`ss_bucket` of the device-sized sample sort (cluster.hip, round 5), its own text on the host: bucket = 2 * #(splitters < key) + (key equals
    that splitter) against numpy's searchsorted, on splitter tables with duplicates -- and the property the sort rests on: the bucket
    number never decreases with the key, and an odd bucket holds copies of ONE key only"""
    rng = np.random.default_rng(21)
    for trial in range(6):
        hi = [1 << 62, 5000, 300, 3, 1 << 40, 1][trial]
        sp = np.sort(rng.integers(0, hi, 1023, dtype=np.uint64))
        table = np.concatenate([sp, np.array([0xFFFFFFFFFFFFFFFF], dtype=np.uint64)])
        keys = np.concatenate([rng.integers(0, max(hi, 2) * 2, 200000, dtype=np.uint64), sp, sp + np.uint64(1), np.array([0, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)])
        out = np.empty(keys.size, dtype=np.int32)
        hx.hx_ss_bucket(table.ctypes.data_as(ctypes.c_void_p), keys.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(keys.size), out.ctypes.data_as(ctypes.c_void_p))
        lo = np.searchsorted(sp, keys, side="left")
        eq = table[lo] == keys
        assert np.array_equal(out, (2 * lo + eq).astype(np.int32)), trial
        order = np.argsort(keys, kind="stable")
        assert (np.diff(out[order]) >= 0).all(), trial
        odd = out % 2 == 1
        for b in np.unique(out[odd])[:50]:
            assert np.unique(keys[out == b]).size == 1, (trial, b)


def test_resize_kernels_on_host_match_pillow(hx):
    """preprocess.hip's two kernels (grid-stride, no wave-level operation: executed by ONE host thread) with the product's coefficient
    tables (ssg_amd.preprocessor.bilinear_coeffs) == Resize((H, W)) + ToTensor + Normalize of the reference loaders
    (selftraining.py:43-47, preprocessor.py:28-30) computed with Pillow + numpy float32: up- and down-scaling, odd sizes."""
    from PIL import Image
    from ssg_amd import preprocessor as pp
    rng = np.random.default_rng(9)
    mean = np.array(pp.MEAN, np.float32); std = np.array(pp.STD, np.float32)
    for (h, w, H, W) in ((128, 64, 256, 128), (300, 117, 256, 128), (57, 31, 64, 32), (256, 128, 256, 128), (640, 480, 256, 128)):
        B = 2
        src = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        src[1] = np.clip(np.add.outer(np.arange(h), np.arange(w))[..., None] * 255.0 / (h + w) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
        (xf, xc, xk), (yf, yc, yk) = pp.bilinear_coeffs(w, W), pp.bilinear_coeffs(h, H)
        xk = np.ascontiguousarray(xk); yk = np.ascontiguousarray(yk)
        tmp = np.empty((B, h, W, 3), np.uint8); out = np.empty((B, 3, H, W), np.float32)
        hx.hx_preprocess(_p(src), B, h, w, H, W, _p(xf), _p(xc), _p(xk), xk.shape[1], _p(yf), _p(yc), _p(yk), yk.shape[1], _p(mean), _p(std), _p(tmp), _p(out))
        for b in range(B):
            pil = np.asarray(Image.fromarray(src[b]).resize((W, H), Image.BILINEAR))
            ref = ((pil.astype(np.float32) / np.float32(255)).transpose(2, 0, 1) - mean[:, None, None]) / std[:, None, None]
            assert ref.dtype == np.float32 and np.array_equal(out[b].view(np.uint32), ref.view(np.uint32)), (h, w, H, W, b)


def test_union_find_kernels_on_host_number_labels_like_sklearn(hx):
    """The connected-components kernels of cluster.hip (union-find with path halving, roots = smallest core index, clusters numbered
    by an exclusive scan over the roots, border points to the smallest adjacent cluster) executed on the host in ssg_dbscan_cc's launch
    order == sklearn.cluster.DBSCAN(metric='precomputed').fit_predict (selftraining.py:299-306) incl. its numbering, whatever the
    order of the edge list -- blobs with noise, border points shared by two clusters, min_samples 1..6, no edges at all."""
    from sklearn.cluster import DBSCAN
    rng = np.random.default_rng(10)
    for trial in range(12):
        n = int(rng.integers(40, 400))
        k = int(rng.integers(2, 9))
        centres = rng.uniform(0, 6, (k, 2))
        x = np.concatenate([centres[rng.integers(0, k, n)] + rng.normal(0, 0.25, (n, 2)), rng.uniform(0, 6, (n // 5, 2))])
        x = x[rng.permutation(x.shape[0])]
        N = x.shape[0]
        dist = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
        for eps, ms in ((0.2, 4), (0.35, 4), (0.3, int(rng.integers(1, 7))), (1e-9, 2)):
            ref = DBSCAN(eps=eps, min_samples=ms, metric="precomputed", n_jobs=1).fit_predict(dist).astype(np.int64)
            hit = dist <= eps
            cnt = hit.sum(1).astype(np.int32)                         # the point itself counts (sklearn's radius neighbours)
            ii, kk = np.nonzero(hit)
            for order in range(2):
                perm = rng.permutation(ii.size) if order else np.arange(ii.size)
                edges = np.ascontiguousarray(np.stack([ii[perm], kk[perm]], 1).astype(np.int32))
                labels = np.empty(N, np.int64)
                hx.hx_dbscan_cc(_p(cnt), _p(edges), ctypes.c_ulonglong(edges.shape[0]), N, ms, _p(labels))
                assert np.array_equal(labels, ref), (trial, eps, ms, order)


def test_split_half_format_carries_22_bits(hx):
    """The embedding's operand format (conv.hip "split-half activations"): hi = half(v), lo = half(v - hi), both numpy's roundings;
    hi + lo reproduces v to 2^-22 relative (absolute floor 2^-25: the half subnormal grid) for |v| < 65504 -- the bound behind the
    5e-6 parity tolerance of the embedding -- and values outside the format are detected by the non-finite test."""
    rng = np.random.default_rng(12)
    n = 4_000_000
    with np.errstate(over="ignore"):
        v = (rng.standard_normal(n) * np.exp(rng.uniform(-25, 11, n))).astype(np.float32)
    v[:8] = [0.0, -0.0, 65503.9, -65503.9, 2.0 ** -24, 2.0 ** -26, 1.0, 1.0 + 2.0 ** -12]
    v = np.ascontiguousarray(np.clip(v, -65503.9, 65503.9))
    hi = np.empty(n // 2, np.uint32); lo = np.empty(n // 2, np.uint32); dec = np.empty(n, np.float32); nf = np.empty(n // 4, np.uint8)
    hx.hx_split(_p(v), ctypes.c_long(n // 4), _p(hi), _p(lo), _p(dec), _p(nf))
    h_ref = v.astype(np.float16)
    l_ref = (v - h_ref.astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi.view(np.uint16), h_ref.view(np.uint16)) and np.array_equal(lo.view(np.uint16), l_ref.view(np.uint16))
    assert np.array_equal(dec.view(np.uint32), (h_ref.astype(np.float32) + l_ref.astype(np.float32)).view(np.uint32)) and not nf.any()
    err = np.abs(dec.astype(np.float64) - v.astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(v.astype(np.float64)) * 2.0 ** -22, 2.0 ** -25))
    big = np.ascontiguousarray(np.array([1.0, 70000.0, 2.0, 3.0, 1.0, 2.0, np.inf, 3.0, np.nan, 0.0, 0.0, 0.0, 65519.9, 0, 0, 0], np.float32))
    nf4 = np.empty(4, np.uint8)
    with np.errstate(all="ignore"):
        hx.hx_split(_p(big), ctypes.c_long(4), _p(hi), _p(lo), _p(dec), _p(nf4))
    assert nf4.tolist() == [1, 1, 1, 0]


def test_pooling_kernels_on_host_vs_torch(hx):
    """The embedding's non-GEMM layers on split-half tensors (conv.hip, executed by one host thread): encode -> 3x3 stride-2 max pooling
    (pad 1) -> decode is EXACTLY torch's max_pool2d of the decoded input (base.py:104 maxpool), and the global + stripe average
    pooling (resnet.py:93-111) equals avg_pool2d over the whole map and over each of the S row stripes to float32 summation accuracy."""
    import torch
    rng = np.random.default_rng(13)
    B, H, W, C = 2, 14, 10, 16
    x = rng.standard_normal((B, H, W, C)).astype(np.float32) * 3
    enc = np.empty_like(x); dec = np.empty_like(x)
    hx.hx_h8l8_encode(_p(x), _p(enc), ctypes.c_long(x.size))
    hx.hx_h8l8_decode(_p(enc), _p(dec), ctypes.c_long(x.size))
    assert np.all(np.abs(dec - x) <= np.abs(x) * 2.0 ** -22 + 2.0 ** -25)
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = np.empty((B, OH, OW, C), np.float32); out_dec = np.empty_like(out)
    hx.hx_maxpool_h8l8(_p(enc), _p(out), B, H, W, C, OH, OW)
    hx.hx_h8l8_decode(_p(out), _p(out_dec), ctypes.c_long(out.size))
    ref = torch.nn.functional.max_pool2d(torch.from_numpy(dec).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(out_dec, ref)                       # (re-encoding a decoded value is exact: hi + lo is its own split)
    for S in (0, 2, 7):
        nsets = S + 1 if S > 1 else 1
        g = np.empty((nsets, B, C), np.float32)
        hx.hx_gap_h8l8(_p(enc), _p(g), B, H, W, C, S)
        t = torch.from_numpy(dec).permute(0, 3, 1, 2).double()
        want = [t.mean(dim=(2, 3))]
        if S > 1:
            hs = H // S
            want += [t[:, :, hs * k:hs * (k + 1)].mean(dim=(2, 3)) for k in range(S)]
        want = torch.stack(want).numpy()
        assert np.abs(g - want).max() < 2e-6 * np.abs(want).max() + 1e-7, S
