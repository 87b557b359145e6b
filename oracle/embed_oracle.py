"""Plain PyTorch fp32 CPU restatement of the embedding path (TEST INFRASTRUCTURE ONLY).

Floating-point kernels keep a torch fp32 reference (the task's rule for fp kernels); this
restates, with torch.nn.functional on the CPU:
  reid/models/base.py:57-93,96-152   Bottleneck ResNet-50 (conv / eval BatchNorm / ReLU / maxpool)
  reid/models/resnet.py:86-111       stop before avgpool; global + stripe average pooling
  reid/evaluators.py:12-16,28-35     fliplr, sum of both orientations, per-set L2 normalisation
from a state_dict with the reference's key names.  It is pinned against the real reference
model (run with stub torchvision/h5py/metric_learn modules) by tools/make_golden.py ->
tests/golden/embed_ref.npz.
"""
import torch
import torch.nn.functional as F

_LAYERS = (3, 4, 6, 3)


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"], False, 0.0, 1e-5)


def feature_map(sd, x):
    """x [B,3,H,W] float32 -> layer4 output [B,2048,H/32,W/32] (resnet.py:87-92)."""
    x = F.relu(_bn(F.conv2d(x, sd["base.conv1.weight"], None, 2, 3), sd, "base.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(_LAYERS):
        for b in range(n):
            p = "base.layer%d.%d" % (li + 1, b)
            stride = 2 if (b == 0 and li > 0) else 1
            out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1), sd, p + ".bn2"))
            out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            res = x
            if (p + ".downsample.0.weight") in sd:
                res = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
            x = F.relu(out + res)
    return x


def pooled(fmap, num_split):
    """resnet.py:93-111 -> list of S+1 [B,2048] (or a single tensor when num_split <= 1)."""
    if num_split > 1:
        h = fmap.size(2)
        x1 = [F.avg_pool2d(fmap, fmap.size()[2:]).view(fmap.size(0), -1)]
        for s in range(num_split):
            xx = fmap[:, :, h // num_split * s: h // num_split * (s + 1), :]
            x1.append(F.avg_pool2d(xx, xx.size()[2:]).view(xx.size(0), -1))
        return x1
    return F.avg_pool2d(fmap, fmap.size()[2:]).view(fmap.size(0), -1)


def fliplr(img):
    return img.index_select(3, torch.arange(img.size(3) - 1, -1, -1).long())


def embed_with_flip(sd, imgs, num_split):
    """evaluators.py:28-35: per set (a + b) / ||a + b||."""
    sd = {k: v.float() for k, v in sd.items() if v.dtype.is_floating_point}
    with torch.no_grad():
        a = pooled(feature_map(sd, imgs), num_split)
        b = pooled(feature_map(sd, fliplr(imgs)), num_split)
        if not isinstance(a, list):
            a, b = [a], [b]
        out = []
        for x, y in zip(a, b):
            s = x + y
            out.append(s / torch.norm(s, p=2, dim=1, keepdim=True))
    return out
