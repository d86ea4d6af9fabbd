#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from a PMC summary (tools/pmc_summary.py output of the FETCH_SIZE and
WRITE_SIZE passes of tools/pmc_passes.sh): HBM bytes per launch of the write and pixel kernels, which bench.py
scales per pixel into roofline.traffic.  usage: pmc_traffic.py <summary.txt> <images per launch> <commit> > json"""
import json, re, sys
summary, images, commit = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kern, cur = {}, None
for line in open(summary):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", line)
    if m and cur:
        kern.setdefault(cur, {})[m.group(1)] = float(m.group(2))
def pick(prefix):
    # the instantiation that did the work: the one with the largest WRITE_SIZE
    c = [(v.get("WRITE_SIZE", 0), k) for k, v in kern.items() if k.startswith(prefix) and "FETCH_SIZE" in v]
    k = max(c)[1]
    return k, {"FETCH_SIZE_KiB": kern[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": kern[k]["WRITE_SIZE"]}
kp, pixel = pick("k_pixel")
kw, write = pick("k_huff_write")
print(json.dumps({
    "source": f"{summary}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_passes.sh), C3 workload, "
              f"{images} images per launch (the benched launch size); both counters in KiB; FETCH_SIZE doubled by bench.py (gfx950 "
              "reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md)",
    "kernel_commit": commit, "images": images, "pixels_per_image": 1920 * 1080,
    "kernel_names": {"pixel": kp, "write": kw},
    "kernels": {"pixel": pixel, "write": write}}, indent=1))
