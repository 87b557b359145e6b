#!/usr/bin/env python3
"""tools/pmc_embed.sh output (per-kernel FETCH_SIZE / WRITE_SIZE sums of 4 forwards) -> <out>.md + <out>.json
(the `roofline.traffic` source of bench.py).  usage: pmc_embed_summary.py gpurun_out/pmc_embed.txt <batch> <out prefix>"""
import json
import sys


def algorithmic_bytes(B):
    """every launch's input + output + residual + weights once, 4 bytes per value, for the current launch set"""
    def t(h, w, c):
        return B * h * w * c * 4
    tot = [0, 0]

    def add(b):
        tot[0] += b; tot[1] += 1
    add(B * 3 * 256 * 128 * 4 + t(64, 32, 64) + 64 * 224 * 4)                                   # stem + maxpool
    add(t(64, 32, 64) + t(64, 32, 256) + (64 * 64 + 64 * 576 + 256 * 128) * 4)                   # layer1 block 1 (fused)
    for _ in (2, 3):
        add(2 * t(64, 32, 256) + (64 * 256 + 64 * 576 + 256 * 64) * 4)                           # layer1 identity blocks (fused)

    def block(h, w, cin, mid, stride, ds, fused=False):
        oh, ow = h // stride, w // stride
        if fused:
            add(2 * t(h, w, cin) + (cin * mid + 9 * mid * mid + mid * 4 * mid) * 4)
            return
        add(t(h, w, cin) + t(h, w, mid) + cin * mid * 4)
        add(t(h, w, mid) + t(oh, ow, mid) + 9 * mid * mid * 4)
        if ds:
            add(t(oh, ow, mid) + t(oh, ow, cin) + t(oh, ow, 4 * mid) + (mid + cin) * 4 * mid * 4)
        else:
            add(t(oh, ow, mid) + 2 * t(oh, ow, 4 * mid) + mid * 4 * mid * 4)
    block(64, 32, 256, 128, 2, True)
    for _ in (2, 3, 4):
        block(32, 16, 512, 128, 1, False, fused=True)
    block(32, 16, 512, 256, 2, True)
    for _ in range(2, 7):
        block(16, 8, 1024, 256, 1, False)
    block(16, 8, 1024, 512, 2, True)
    for _ in (2, 3):
        block(8, 4, 2048, 512, 1, False)
    return tot[0], tot[1]


def main():
    txt, B, out = open(sys.argv[1]).read(), int(sys.argv[2]), sys.argv[3]
    sec, cur = {}, None
    for line in txt.splitlines():
        if line.startswith("== "):
            cur = line[3:].strip(); sec[cur] = []
        elif cur and "," in line and not line.startswith("kernel,"):
            sec[cur].append(line)

    def conv_sum(rows):
        s, n, keep = 0.0, 0, []
        for r in rows:
            if any(k in r for k in ("bottleneck_kernel", "conv_dma_kernel", "conv_igemm_kernel", "stem_pool_kernel")):
                parts = r.rsplit(",", 3)
                s += float(parts[3]); n += int(parts[2]); keep.append(r)
        return s, n, keep
    fs, fn, fk = conv_sum(sec["FETCH_SIZE"]); ws, wn, wk = conv_sum(sec["WRITE_SIZE"])
    fetch, write = fs * 1024 * 2 / 4, ws * 1024 / 4
    alg, nl = algorithmic_bytes(B)
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from bench import build_fingerprint, embed_fingerprint
    js = {"source": "profiles/%s.md" % __import__("os").path.basename(out), "build": build_fingerprint(), "embed_build": embed_fingerprint(), "batch": B, "launches_per_forward": fn // 4, "fetch_bytes_per_forward": fetch,
          "write_bytes_per_forward": write, "algorithmic_bytes_per_forward": alg}
    json.dump(js, open(out + ".json", "w"))
    md = """# rocprofv3 PMC: HBM traffic of the embedding's convolution launches (fused stem, layer1 and layer2 identity blocks)

Separate passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (`--output-format csv`, `tools/pmc_embed.sh`) over
`python tools/time_embed.py --B %d --iters 1` = 4 forwards of %d images (warm-up + timed, original + flipped), aggregated per
kernel with `tools/pmc_agg.py`, summarised by `tools/pmc_embed_summary.py`.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE taken as is.  Counter unit KB (1024 B).

| | all convolution-carrying launches (%d = 4 forwards x %d) |
|---|---:|
| FETCH_SIZE sum | %.4g KB -> corrected %.1f GB = **%.1f GB per forward** |
| WRITE_SIZE sum | %.4g KB = %.1f GB = **%.1f GB per forward** |
| HBM traffic per forward (B = %d) | **%.1f GB** = %.1f MB per image (%.0f MB per launch on average) |
| algorithmic bytes per forward of THIS launch set (%d launches: every launch's input + output + residual + weights once, 4 B per value) | %.1f GB |
| traffic / algorithmic | %.2f |
| round 1 (49 conv launches + maxpool + layout kernel, B = 512) | 36.2 GB = 70.7 MB per image |

Raw per-kernel rows (kernel, grid, calls, KB):

```
FETCH_SIZE
%s
WRITE_SIZE
%s
```
""" % (B, B, fn, fn // 4, fs, fs * 1024 * 2 / 1e9, fetch / 1e9, ws, ws * 1024 / 1e9, write / 1e9, B, (fetch + write) / 1e9,
       (fetch + write) / B / 1e6, (fetch + write) / max(fn // 4, 1) / 1e6, nl, alg / 1e9, (fetch + write) / alg, "\n".join(fk), "\n".join(wk))
    open(out + ".md", "w").write(md)
    print(js)


if __name__ == "__main__":
    main()
