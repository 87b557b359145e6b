"""ResNet-50 embedder -- host-side mirror of reid/models/resnet.py (wrapper, :17-148) over the
torchvision ResNet-50 architecture spelled out in reid/models/base.py:57-152.

`create('resnet50', num_classes=0, num_split=S, cluster=False)` returns an object with the
reference's call surface: `state_dict()` / `load_state_dict()` with the reference's key names
(`base.conv1.weight`, `base.layer1.0.bn2.running_var`, `feat.weight`, `feat_bn.*` ...),
`eval()`, `cuda()`, and `model(x, for_eval)` -> `(x1, x2)` where x1 is the list of S+1
pooled feature sets (or their concatenation when for_eval=True), resnet.py:86-134.

The forward runs on hand-written HIP kernels (csrc/conv.hip) through the C ABI: NHWC fp32,
eval-mode BatchNorm folded into the convolution weights at load time, bias/residual/ReLU
fused into the GEMM epilogue.  precision='split' (default) carries every fp32 activation /
weight as two halves (hi + lo, 22 significand bits) and evaluates the products on the fp16
matrix cores with fp32 accumulation (include/ssg_hip.h, ssg_conv2d_nhwc_x); precision='f32'
keeps everything on the fp32 matrix cores.  There is no training path here (fine-tuning is out of scope,
SURVEY.md section 2 rows 10-11).
"""
from collections import OrderedDict

import math
import os

import torch

from . import _lib
from ._lib import check, ptr, stream

_LAYERS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
_BN_EPS = 1e-5


def _arch(depth):
    """[(prefix, cin, cout, k, stride, pad)] conv list + block structure of a Bottleneck ResNet."""
    blocks = []
    inplanes = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), _LAYERS[depth])):
        for b in range(n):
            stride = 2 if (b == 0 and li > 0) else 1
            down = b == 0 and (stride != 1 or inplanes != planes * 4)
            blocks.append(dict(prefix="base.layer%d.%d" % (li + 1, b), inplanes=inplanes, planes=planes, stride=stride, down=down))
            inplanes = planes * 4
    return blocks


def synthetic_state_dict(seed=1, depth=50, num_features=2048, randomize_bn=True):
    """Deterministic random weights with the reference's shapes and key names: Kaiming-normal
    fan_out convolutions (reid/models/base.py:113-118), BatchNorm statistics drawn at random so
    that the BN folding is exercised (gamma ~ U(.5,1.5), beta, mean ~ N(0,.1), var ~ U(.5,1.5)).
    There is no network access for ImageNet checkpoints; benchmarks and parity tests use this."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(name, cout, cin, k):
        std = (2.0 / (cout * k * k)) ** 0.5
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std

    def bn(name, c):
        if randomize_bn:
            sd[name + ".weight"] = torch.rand(c, generator=g) + 0.5
            sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
            sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
            sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5
        else:
            sd[name + ".weight"] = torch.ones(c); sd[name + ".bias"] = torch.zeros(c)
            sd[name + ".running_mean"] = torch.zeros(c); sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    conv("base.conv1", 64, 3, 7); bn("base.bn1", 64)
    for blk in _arch(depth):
        p, ip, pl = blk["prefix"], blk["inplanes"], blk["planes"]
        conv(p + ".conv1", pl, ip, 1); bn(p + ".bn1", pl)
        conv(p + ".conv2", pl, pl, 3); bn(p + ".bn2", pl)
        conv(p + ".conv3", pl * 4, pl, 1); bn(p + ".bn3", pl * 4)
        if blk["down"]:
            conv(p + ".downsample.0", pl * 4, ip, 1); bn(p + ".downsample.1", pl * 4)
    sd["base.fc.weight"] = torch.randn(1000, 2048, generator=g) * 0.01
    sd["base.fc.bias"] = torch.zeros(1000)
    if num_features > 0:
        sd["feat.weight"] = torch.randn(num_features, 2048, generator=g) * 0.001      # init.normal_(std=0.001) resnet.py:67
        sd["feat_bn.weight"] = torch.ones(num_features); sd["feat_bn.bias"] = torch.zeros(num_features)
        sd["feat_bn.running_mean"] = torch.zeros(num_features); sd["feat_bn.running_var"] = torch.ones(num_features)
        sd["feat_bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return sd


class _FoldedConv:
    __slots__ = ("w", "bias", "cin", "cout", "k", "stride", "pad", "split", "acc_scale", "cscale")


_IN_SPLIT, _OUT_SPLIT = 1, 2      # include/ssg_hip.h SSG_CONV_IN_SPLIT / SSG_CONV_OUT_SPLIT


def _h8l8(v):
    """float32 [rows, K] (K % 8 == 0) -> the same shape of float32 CONTAINERS in the split-half layout:
    per 8 values 32 bytes = [8 x half hi][8 x half lo], hi = half(v), lo = half(v - hi)."""
    rows, K = v.shape
    hi = v.half()
    lo = (v - hi.float()).half()
    g = torch.stack([hi.view(rows, K // 8, 8), lo.view(rows, K // 8, 8)], dim=2)      # [rows, K/8, 2, 8]
    return g.reshape(rows, 2 * K).contiguous().view(torch.float32)


def _h4l4(v):
    """stem layout: per filter tap (4 channels) 16 bytes = [4 x half hi][4 x half lo]"""
    rows, K = v.shape
    hi = v.half()
    lo = (v - hi.float()).half()
    g = torch.stack([hi.view(rows, K // 4, 4), lo.view(rows, K // 4, 4)], dim=2)      # [rows, K/4, 2, 4]
    return g.reshape(rows, 2 * K).contiguous().view(torch.float32)


def _weight_scale(w):
    """power of two that lifts the weights towards 1 (out of the half subnormals) without overflow"""
    mx = float(w.abs().max())
    return 256.0 if mx == 0 else float(min(256.0, 2.0 ** math.floor(math.log2(16384.0 / mx))))


def _row_scales(w):
    """per-output-channel powers of two s_c with max|w_c| * s_c in (8192, 16384]: every weight row uses the full half range
    (hi AND lo parts normal) whatever the folded BatchNorm scale of its channel is -- real checkpoints have per-channel
    scales spanning more than 10^3, which one per-layer scale pushes into the half subnormals.  Exponent clamped to +-40."""
    mx = w.abs().amax(dim=1).double()
    e = torch.floor(torch.log2(16384.0 / mx.clamp_min(1e-300))).clamp(-40, 40)
    e = torch.where(mx > 0, e, torch.zeros_like(e))
    return torch.pow(2.0, e).float()


def pack_weight_khwc(w):
    """[Cout, KH, KW, Cin] (Cin % 32 == 0) -> [Cout, K] in the kernels' reduction order
    k = ((c // 32) * KH*KW + r*KW + s) * 32 + c % 32  (include/ssg_hip.h, ssg_conv2d_nhwc_f32)."""
    cout, kh, kw, cin = w.shape
    return w.reshape(cout, kh * kw, cin // 32, 32).permute(0, 2, 1, 3).reshape(cout, kh * kw * cin).contiguous()


def _fold(sd, conv_name, bn_name, stride, pad, device, split=False):
    """conv + eval BatchNorm -> (w [Cout][Kpad] with k=(r,s,c), bias [Cout]); float64 fold.
    split=True: w in the h8l8 split-half layout, pre-multiplied by 1/acc_scale."""
    w = sd[conv_name + ".weight"].double()
    gamma, beta = sd[bn_name + ".weight"].double(), sd[bn_name + ".bias"].double()
    mean, var = sd[bn_name + ".running_mean"].double(), sd[bn_name + ".running_var"].double()
    scale = gamma / torch.sqrt(var + _BN_EPS)
    w = w * scale.view(-1, 1, 1, 1)
    bias = beta - mean * scale
    cout, cin, k, _ = w.shape
    w = w.permute(0, 2, 3, 1).contiguous()                     # [Cout, KH, KW, Cin]
    if cin == 3:                                                 # stem: RGB0 pixels, one tap per float4
        w = torch.nn.functional.pad(w, (0, 1))
        cin = 4
        kpad = 32 * ((k * k + 7) // 8)
        w = torch.nn.functional.pad(w.reshape(cout, k * k * 4), (0, kpad - k * k * 4))
    else:
        w = pack_weight_khwc(w)
    f = _FoldedConv()
    w = w.float().contiguous()
    f.split, f.acc_scale, f.cscale = bool(split), 1.0, None
    if split:
        sc = _row_scales(w)                          # exact: powers of two
        w = w * sc.view(-1, 1)
        w = _h4l4(w) if cin == 4 else _h8l8(w)
        f.cscale = (1.0 / sc).contiguous().to(device)
    f.w = w.to(device); f.bias = bias.float().contiguous().to(device)
    f.cin, f.cout, f.k, f.stride, f.pad = cin, cout, k, stride, pad
    return f


class ResNet:
    """Mirror of reid.models.resnet.ResNet (resnet.py:17-148), forward only."""

    def __init__(self, depth=50, checkpoint=None, pretrained=True, num_features=2048, dropout=0.1, num_classes=0, num_split=1,
                 mode='Dissimilarity', cluster=False, seed=1, precision=None):
        if depth not in _LAYERS:
            raise KeyError("Unsupported depth:", depth)
        if cluster:
            raise NotImplementedError("cluster=True (DEC head, reid/models/dce.py) is outside the grouping hot path")
        if num_classes > 0:
            raise NotImplementedError("num_classes > 0 (dropout + classifier head, resnet.py:118-120) is a training-only path; "
                                      "the grouping path uses num_classes=0 (selftraining.py:121-123)")
        if num_features > 0 and num_features % 64:
            raise ValueError("num_features must be a multiple of 64")
        self.depth, self.num_features, self.dropout, self.num_classes = depth, num_features, dropout, num_classes
        self.num_split, self.cluster, self.pretrained, self.training = num_split, cluster, pretrained, False
        self.precision = precision or os.environ.get("SSG_EMBED_PRECISION", "split")
        if self.precision not in ("split", "f32"):
            raise ValueError("precision must be 'split' or 'f32'")
        self.device = torch.device("cpu")
        self._sd = synthetic_state_dict(seed, depth, num_features)
        self._folded = None
        self._twin = None
        self.flip_streams = True         # embed_with_flip: the original and the flipped forward on two HIP streams (SSG_FLIP_STREAMS=0: one)
        self._weights = "synthetic"      # until load_state_dict puts real backbone weights in
        if checkpoint:
            self.load_state_dict(torch.load(checkpoint, map_location="cpu"), strict=False)
        elif pretrained:
            # reference: torchvision's ImageNet weights are downloaded here (resnet.py:49); this build has no network and
            # no torchvision, so the backbone starts from the seeded synthetic initialisation until a checkpoint is loaded
            import warnings
            warnings.warn("ssg_amd ResNet(pretrained=True): no ImageNet weights are available offline; the backbone holds seeded "
                          "synthetic weights until load_state_dict()/checkpoint= supplies real ones", stacklevel=2)

    # ---- torch.nn.Module-like surface used by selftraining.py / evaluators.py
    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        """torch.nn.Module.load_state_dict look-alike.  Accepts what the reference's call sites pass (selftraining.py:129-132,
        serialization.py:31-40): the checkpoint dict itself ({'state_dict': ...}) and nn.DataParallel's 'module.' key prefix.
        With strict=False torch leaves unmatched keys at their previous (ImageNet) values; here the previous values are
        synthetic, so backbone keys that stay unmatched are reported with a warning instead of silently producing garbage."""
        if isinstance(state_dict, dict) and "state_dict" in state_dict and not any(k in self._sd for k in state_dict):
            state_dict = state_dict["state_dict"]
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = OrderedDict((k[len("module."):], v) for k, v in state_dict.items())
        missing = [k for k in self._sd if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing %r unexpected %r" % (missing[:5], unexpected[:5]))
        for k, v in state_dict.items():
            if k in self._sd:
                if tuple(v.shape) != tuple(self._sd[k].shape):
                    raise RuntimeError("size mismatch for %s: %r vs %r" % (k, tuple(v.shape), tuple(self._sd[k].shape)))
                self._sd[k] = v.detach().to("cpu").clone()
        self._invalidate()
        miss_base = [k for k in missing if k.startswith("base.") and not k.endswith("num_batches_tracked")]
        if miss_base:
            import warnings
            warnings.warn("ssg_amd ResNet.load_state_dict: %d backbone tensors were not in the state dict and keep their synthetic "
                          "values (first: %s)" % (len(miss_base), miss_base[0]), stacklevel=2)
        else:
            self._weights = "loaded"
        return missing, unexpected

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("ssg_amd.resnet.ResNet is an inference-only embedder")
        return self.eval()

    def _invalidate(self):
        """weights or device changed: drop the folded weights AND the cached fp32 fallback model (it keeps folded weights and a
        device of its own; INTEGRATION.md's loop calls load_state_dict() every self-training iteration)"""
        self._folded = None
        self._twin = None

    def cuda(self, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._invalidate()
        return self

    def to(self, device):
        self.device = torch.device(device); self._invalidate()
        return self

    def cpu(self):
        return self.to("cpu")

    @property
    def module(self):   # nn.DataParallel(model).module compatibility (selftraining.py:135,230)
        return self

    # ---- weights
    def _prepare(self):
        if self._folded is not None:
            return self._folded
        if self.device.type != "cuda":
            raise _lib.SSGError("the embedder runs on the GPU only: call model.cuda() first (no CPU fallback)")
        sd, dev = self._sd, self.device
        sp = self.precision == "split"
        net = dict(stem=_fold(sd, "base.conv1", "base.bn1", 2, 3, dev, split=sp), blocks=[], split=sp)
        for blk in _arch(self.depth):
            p = blk["prefix"]
            if blk["down"]:
                # downsample branch fused into conv3: one GEMM over K = planes + inplanes (the residual tensor
                # is never written to / re-read from HBM); both halves share one weight scale
                c3 = _fold(sd, p + ".conv3", p + ".bn3", 1, 0, dev)
                ds = _fold(sd, p + ".downsample.0", p + ".downsample.1", blk["stride"], 0, dev)
                wcat = torch.cat([c3.w, ds.w], dim=1)
                if sp:
                    sc = _row_scales(wcat.cpu())
                    ds.w = _h8l8((wcat.cpu() * sc.view(-1, 1))).to(dev); ds.acc_scale = 1.0; ds.cscale = (1.0 / sc).contiguous().to(dev); ds.split = True
                else:
                    ds.w = wcat.contiguous()
                ds.bias = (c3.bias + ds.bias).contiguous()
            else:
                c3 = _fold(sd, p + ".conv3", p + ".bn3", 1, 0, dev, split=sp)
                ds = None
            net["blocks"].append(dict(
                c1=_fold(sd, p + ".conv1", p + ".bn1", 1, 0, dev, split=sp),
                c2=_fold(sd, p + ".conv2", p + ".bn2", blk["stride"], 1, dev, split=sp),
                c3=c3, ds=ds))
        self._folded = net
        return net

    # ---- forward
    @staticmethod
    def _conv(L, x, f, res=None, relu=True, out_split=False, ovf=None):
        B, H, W, _ = x.shape
        OH = (H + 2 * f.pad - f.k) // f.stride + 1; OW = (W + 2 * f.pad - f.k) // f.stride + 1
        out = torch.empty((B, OH, OW, f.cout), dtype=torch.float32, device=x.device)
        flags = (_IN_SPLIT if f.split else 0) | (_OUT_SPLIT if out_split else 0)
        check(L.ssg_conv2d_nhwc_x(ptr(x), ptr(f.w), ptr(f.bias), ptr(res), ptr(out), B, H, W, f.cin, f.cout, f.k, f.k, f.stride, f.pad,
                                  1 if relu else 0, flags, f.acc_scale, ptr(getattr(f, "cscale", None)), ptr(ovf), stream()), "ssg_conv2d_nhwc_x")
        return out

    @staticmethod
    def _conv_dual(L, o, x, c3, ds, out_split=False, ovf=None):
        """relu(conv3(o) + downsample(x)) as one GEMM (ssg_conv1x1_dual_nhwc_x)."""
        B, H, W, _ = o.shape
        _, H2, W2, _ = x.shape
        out = torch.empty((B, H, W, c3.cout), dtype=torch.float32, device=o.device)
        flags = (_IN_SPLIT if ds.split else 0) | (_OUT_SPLIT if out_split else 0)
        check(L.ssg_conv1x1_dual_nhwc_x(ptr(o), ptr(x), ptr(ds.w), ptr(ds.bias), ptr(out), B, H, W, c3.cin, H2, W2, ds.cin, ds.stride, c3.cout, 1,
                                        flags, ds.acc_scale, ptr(getattr(ds, "cscale", None)), ptr(ovf), stream()), "ssg_conv1x1_dual_nhwc_x")
        return out

    @staticmethod
    def _bottleneck(L, y, blk, ovf=None):
        """whole bottleneck block in one launch (ssg_bottleneck_nhwc_x / ssg_bottleneck_ds_nhwc_x); None when there is no fused
        kernel for this block (stride-2 blocks, other shapes)"""
        B, H, W, CIN = y.shape
        c1, c2, c3, ds = blk["c1"], blk["c2"], blk["c3"], blk["ds"]
        if os.environ.get("SSG_FUSED_BOTTLENECK", "1") == "0" or c2.stride != 1 or (ds is not None and ds.stride != 1):
            return None
        if not L.ssg_bottleneck_supported(H, W, CIN, c3.cout, c1.cout):
            return None
        out = torch.empty((B, H, W, c3.cout), dtype=torch.float32, device=y.device)
        if ds is None:
            check(L.ssg_bottleneck_nhwc_x(ptr(y), ptr(c1.w), ptr(c1.bias), ptr(c1.cscale), ptr(c2.w), ptr(c2.bias), ptr(c2.cscale),
                                          ptr(c3.w), ptr(c3.bias), ptr(c3.cscale), ptr(out), B, H, W, CIN, c1.cout, ptr(ovf), stream()),
                  "ssg_bottleneck_nhwc_x")
        else:       # ds.w = [conv3 | downsample] weights along K, ds.bias = b3 + b_ds (see _prepare)
            check(L.ssg_bottleneck_ds_nhwc_x(ptr(y), ptr(c1.w), ptr(c1.bias), ptr(c1.cscale), ptr(c2.w), ptr(c2.bias), ptr(c2.cscale),
                                             ptr(ds.w), ptr(ds.bias), ptr(ds.cscale), ptr(out), B, H, W, CIN, c3.cout, c1.cout, ptr(ovf), stream()),
                  "ssg_bottleneck_ds_nhwc_x")
        return out

    def _fmap(self, x, flip=False, hooks=None):
        """-> (layer4 map [B,h,w,2048], is_split): with precision='split' the float32 tensor is a container of
        h8l8 split halves (decode with ssg_h8l8_decode).  hooks: {block index: callable} run on the host in front of that block's
        launches (the staggered two-stream schedule records / waits for its events there)."""
        L = _lib.lib()
        net = self._prepare()
        sp = net["split"]
        x = x.to(self.device, torch.float32).contiguous()
        B, C, H, W = x.shape
        if C != 3:
            raise ValueError("expected RGB images [B,3,H,W]")
        ovf = self._overflow_flag() if sp else None
        if sp and os.environ.get("SSG_FUSED_STEM", "1") != "0" and L.ssg_stem_pool_supported(H, W):
            # conv1 + bn1 + relu + maxpool (+ the layout change and the flip) in one launch: the stem map stays on chip
            st = net["stem"]
            y = torch.empty((B, H // 4, W // 4, 64), dtype=torch.float32, device=self.device)
            check(L.ssg_stem_pool_nchw_x(ptr(x), 1 if flip else 0, ptr(st.w), ptr(st.bias), ptr(st.cscale), ptr(y), B, H, W, ptr(ovf), stream()),
                  "ssg_stem_pool_nchw_x")
        else:
            x4 = torch.empty((B, H, W, 4), dtype=torch.float32, device=self.device)
            if sp:
                check(L.ssg_nchw_to_nhwc4_h4l4(ptr(x), ptr(x4), B, H, W, 1 if flip else 0, stream()), "ssg_nchw_to_nhwc4_h4l4")
            else:
                check(L.ssg_nchw_to_nhwc4(ptr(x), ptr(x4), B, H, W, 1 if flip else 0, stream()), "ssg_nchw_to_nhwc4")
            y = self._conv(L, x4, net["stem"], out_split=sp, ovf=ovf)
            _, H2, W2, _ = y.shape
            p = torch.empty((B, (H2 + 1) // 2, (W2 + 1) // 2, 64), dtype=torch.float32, device=self.device)
            if sp:
                check(L.ssg_maxpool3x3s2_h8l8(ptr(y), ptr(p), B, H2, W2, 64, stream()), "ssg_maxpool3x3s2_h8l8")
            else:
                check(L.ssg_maxpool3x3s2_nhwc(ptr(y), ptr(p), B, H2, W2, 64, stream()), "ssg_maxpool3x3s2_nhwc")
            y = p
        # SSG_CONV_PAIR=1 (round 6, experimental, off by default: measured in DESIGN.md section 11): conv3 + residual of an identity block and
        # conv1 of the next block as one launch (ssg_conv_pair_nhwc_x) where a kernel exists (layer3) and the tiles fill the chip
        pair = sp and os.environ.get("SSG_CONV_PAIR", "0") == "1"
        blocks, o1_next = net["blocks"], None
        for bi, blk in enumerate(blocks):
            if hooks and bi in hooks:
                hooks[bi]()
            if sp:
                fused = self._bottleneck(L, y, blk, ovf)
                if fused is not None:
                    y = fused
                    continue
            o = o1_next if o1_next is not None else self._conv(L, y, blk["c1"], out_split=sp, ovf=ovf)
            o1_next = None
            o = self._conv(L, o, blk["c2"], out_split=sp, ovf=ovf)
            if blk["ds"] is not None:
                y = self._conv_dual(L, o, y, blk["c3"], blk["ds"], out_split=sp, ovf=ovf)
                continue
            nxt = blocks[bi + 1]["c1"] if bi + 1 < len(blocks) else None
            Bo, Ho, Wo, _ = o.shape
            if (pair and nxt is not None and nxt.k == 1 and nxt.stride == 1 and nxt.cin == blk["c3"].cout and Bo * Ho * Wo >= 128 * 256
                    and L.ssg_conv_pair_supported(blk["c3"].cin, blk["c3"].cout, nxt.cout)):
                y, o1_next = self._conv_pair(L, o, blk["c3"], y, nxt, ovf)
            else:
                y = self._conv(L, o, blk["c3"], res=y, relu=True, out_split=sp, ovf=ovf)
        return y, sp

    @staticmethod
    def _conv_pair(L, o, c3, res, c1n, ovf=None):
        """(relu(conv3(o) + res), relu(conv1_next(that))) in one launch (ssg_conv_pair_nhwc_x): split-half tensors, 1x1 convolutions"""
        B, H, W, _ = o.shape
        out = torch.empty((B, H, W, c3.cout), dtype=torch.float32, device=o.device)
        y1n = torch.empty((B, H, W, c1n.cout), dtype=torch.float32, device=o.device)
        check(L.ssg_conv_pair_nhwc_x(ptr(o), ptr(c3.w), ptr(c3.bias), ptr(c3.cscale), ptr(res), ptr(out), ptr(c1n.w), ptr(c1n.bias), ptr(c1n.cscale),
                                     ptr(y1n), B * H * W, c3.cin, c3.cout, c1n.cout, ptr(ovf), stream()), "ssg_conv_pair_nhwc_x")
        return out, y1n

    # ---- split-half range guard: activations are half pairs, |v| >= 65520 cannot be stored.  Every convolution raises a
    # device flag when it has to encode such a value; the public entry points read it once per call and recompute the batch
    # on the fp32 matrix cores (same weights) instead of returning inf / NaN / silently clipped features.
    def _side_streams(self):
        """the two HIP streams `embed_with_flip` runs its two forwards on"""
        st = getattr(self, "_streams", None)
        if st is None or st[0].device != self.device:
            # SSG_FLIP_PRIO=1 (A/B knob, measured in DESIGN.md section 11): the first stream at high priority, so that its launches take the CUs
            # first and the other forward only fills what they leave -- one kernel's working set in the L2s at a time instead of two
            prio = -1 if os.environ.get("SSG_FLIP_PRIO", "0") == "1" else 0
            st = self._streams = (torch.cuda.Stream(self.device, priority=prio), torch.cuda.Stream(self.device))
        return st

    def _overflow_flag(self):
        if getattr(self, "_ovf", None) is None or self._ovf.device != self.device:
            self._ovf = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._ovf

    def _overflowed(self):
        """read and clear the flag (one host round trip)"""
        if self.precision != "split" or getattr(self, "_ovf", None) is None:
            return False
        hit = bool(int(self._ovf.item()))
        if hit:
            self._ovf.zero_()
        return hit

    def _f32_twin(self):
        if getattr(self, "_twin", None) is None:
            import warnings
            warnings.warn("ssg_amd ResNet: an activation left the half range of the split-half path (|v| >= 65520); this batch and any "
                          "later one that overflows is recomputed with precision='f32'", stacklevel=3)
            t = ResNet.__new__(ResNet)
            t.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_folded", "_ovf", "_twin")})
            t.precision, t._folded, t._ovf, t._twin = "f32", None, None, None
            self._twin = t
        return self._twin

    def feature_map(self, x, flip=False):
        """images [B,3,H,W] float32 (NCHW, any device) -> layer4 map [B,H/32,W/32,2048] NHWC fp32
        (resnet.py:87-92: every base module up to, not including, avgpool)."""
        y, sp = self._fmap(x, flip)
        if sp:
            if self._overflowed():
                return self._f32_twin().feature_map(x, flip)
            out = torch.empty_like(y)
            check(_lib.lib().ssg_h8l8_decode(ptr(y), ptr(out), y.numel(), 1.0, stream()), "ssg_h8l8_decode")
            return out
        return y

    def pooled(self, fmap, split=False):
        """[B,h,w,2048] -> [(S+1), B, 2048] (whole + S stripes) or [1,B,2048] (resnet.py:93-111)."""
        L = _lib.lib()
        B, h, w, C = fmap.shape
        S = self.num_split if self.num_split > 1 else 1
        nsets = S + 1 if S > 1 else 1
        out = torch.empty((nsets, B, C), dtype=torch.float32, device=fmap.device)
        if split:
            check(L.ssg_gap_stripes_h8l8(ptr(fmap), ptr(out), B, h, w, C, S, stream()), "ssg_gap_stripes_h8l8")
        else:
            check(L.ssg_gap_stripes(ptr(fmap), ptr(out), B, h, w, C, S, stream()), "ssg_gap_stripes")
        return out

    def _x2(self, gap):
        """x2 = relu(feat_bn(feat(gap))) (resnet.py:112-117; unused by the extraction path, which takes
        model(...)[0]) -- Linear + eval BatchNorm1d folded into one 1x1 GEMM on the same HIP kernel."""
        if self.num_features <= 0:
            return None
        net = self._prepare()
        if "feat" not in net:
            sd = dict(self._sd)
            sd["feat.weight4"] = sd["feat.weight"].view(self.num_features, 2048, 1, 1)
            f = _fold({"c.weight": sd["feat.weight4"], "b.weight": sd["feat_bn.weight"], "b.bias": sd["feat_bn.bias"],
                       "b.running_mean": sd["feat_bn.running_mean"], "b.running_var": sd["feat_bn.running_var"]}, "c", "b", 1, 0, self.device)
            net["feat"] = f
        B = gap.shape[0]
        out = self._conv(_lib.lib(), gap.reshape(B, 1, 1, 2048).contiguous(), net["feat"], relu=True)
        return out.reshape(B, self.num_features)

    def __call__(self, x, for_eval=False):
        sets = self.pooled(*self._fmap(x))
        if self._overflowed():
            return self._f32_twin()(x, for_eval)
        x2 = self._x2(sets[0])
        if self.num_split > 1:
            x1 = [sets[s] for s in range(sets.shape[0])]
            if for_eval:
                return torch.cat(x1, dim=1), x2
            return x1, x2
        return sets[0], x2

    forward = __call__

    def embed_with_flip(self, x, for_eval=False, check_overflow=True):
        """Fused reid/evaluators.py:28-35: features of x and fliplr(x) summed and L2-normalised.
        Returns [(S+1), B, 2048] (per-set norm) or, for_eval / single set, [B, (S+1)*2048].
        check_overflow=False leaves the split-half range flag unread (no host round trip): the caller reads it once after its last
        batch with `_overflowed()` and recomputes on the fp32 path if it is set (`evaluators.extract_embeddings` does)."""
        L = _lib.lib()
        x = x.to(self.device, torch.float32)          # one H2D copy for both orientations
        if self.flip_streams and self.precision == "split" and os.environ.get("SSG_FLIP_STREAMS", "1") != "0":
            # the two orientations are independent: one HIP stream each, so that the tail of one launch (its last, partial wave of
            # workgroups) and the gap to the next are filled by the other forward's launches (+2.5 ... 3.4 % measured at B = 1000)
            cur = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event(); ready.record(cur)                    # x (and whatever produced it) is ready
            ab = []
            for st, flip in zip(self._side_streams(), (False, True)):
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    r = self.pooled(*self._fmap(x, flip=flip))
                r.record_stream(cur)                                         # allocated on the side stream's pool, consumed on `cur`
                ab.append(r)
            for st in self._side_streams():
                cur.wait_stream(st)
            a, b = ab
        else:
            a = self.pooled(*self._fmap(x, flip=False))
            b = self.pooled(*self._fmap(x, flip=True))
        if check_overflow and self._overflowed():
            return self._f32_twin().embed_with_flip(x, for_eval)
        nsets, B, C = a.shape
        if for_eval or nsets == 1:
            a = a.permute(1, 0, 2).reshape(B, nsets * C).contiguous(); b = b.permute(1, 0, 2).reshape(B, nsets * C).contiguous()
            out = torch.empty_like(a)
            check(L.ssg_flip_sum_l2norm(ptr(a), ptr(b), ptr(out), B, nsets * C, stream()), "ssg_flip_sum_l2norm")
            return out
        out = torch.empty_like(a)
        check(L.ssg_flip_sum_l2norm(ptr(a), ptr(b), ptr(out), nsets * B, C, stream()), "ssg_flip_sum_l2norm")
        return out


def resnet50(**kwargs):
    return ResNet(50, **kwargs)


def resnet101(**kwargs):
    return ResNet(101, **kwargs)


def resnet152(**kwargs):
    return ResNet(152, **kwargs)


_factory = {"resnet50": resnet50, "resnet101": resnet101, "resnet152": resnet152}


def names():
    return sorted(_factory)


def create(name, *args, **kwargs):
    """reid/models/__init__.py create(): models.create('resnet50', num_classes=0, num_split=2, cluster=False)."""
    if name not in _factory:
        raise KeyError("Unknown model:", name)
    return _factory[name](*args, **kwargs)
