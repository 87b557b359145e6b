#!/usr/bin/env python3
"""tools/pmc_generic.sh output (per-kernel counter averages) -> a markdown table of matrix-pipe utilisation and wave states per kernel.
usage: pmc_mfma_summary.py <pmc txt> [<pmc txt> ...] > profiles/r03_pmc_mfma.md"""
import collections
import re
import sys

SIMDS, XCDS = 1024, 8
rows = collections.OrderedDict()
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r"(.{62}) (\S+)\s+([\d.e+-]+) per launch \((\d+) launches\)", line)
        if not m:
            continue
        k = m.group(1).strip()
        rows.setdefault(k, {})[m.group(2)] = float(m.group(3)); rows[k]["launches"] = int(m.group(4))
print("| kernel | launches | GPU cycles per launch | MFMA busy (of %d SIMDs) | waves: issuing | waiting to issue | parked (waitcnt / barrier) |" % SIMDS)
print("|---|---:|---:|---:|---:|---:|---:|")
for k, c in rows.items():
    if "GRBM_GUI_ACTIVE" not in c or not k.startswith(("void ssg", "ssg::", "_ZN3ssg")):
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / XCDS                        # the counter is summed over the 8 XCDs
    if cyc < 2e5 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0:
        continue
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1.0)
    print("| `%s` | %d | %.3g | %.1f %% | %.0f %% | %.0f %% | %.0f %% |" % (k[:58], c["launches"], cyc, 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * SIMDS),
                                                                       100.0 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100.0 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100.0 * c.get("SQ_WAIT_ANY", 0) / wc))
