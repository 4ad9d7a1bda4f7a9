"""Write one entry of profiles/hbm_traffic.json from a tools/pmc.sh summary, stamped with the digest of the kernel
sources the counters were measured on (bench.py reports `roofline.traffic` only for that build).

  python tools/update_traffic.py <workload key> <summary.txt> [<summary.txt> ...] [--kernels=substr,substr] [--unit=fa_bwd_capi.hip]

The stamp is the digest of the translation unit the kernels are compiled from (default: the forward's, fa_capi.hip).

HBM bytes per launch = sum over the listed kernels of FETCH_SIZE x 2 (gfx950 correction: the counter tallies 128-byte
requests at 64 bytes, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE, both in KiB in the summaries.
"""
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from flash_attention_impls_amd import _build  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = [a for a in sys.argv[1:] if a.startswith("--kernels")]
unit = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--unit=")] or ["fa_capi.hip"])[0]
key, files = args[0], args[1:]
want = opts[0].split("=", 1)[1].split(",") if opts and "=" in opts[0] else None

per_kernel = {}
for f in files:
    for line in open(f):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+mean=([0-9.e+]+)", line)
        if m:
            per_kernel.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
total = 0.0
parts = []
for name, c in per_kernel.items():
    if want and not any(s in name for s in want):
        continue
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    total += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    parts.append(f"{name}: FETCH_SIZE {c['FETCH_SIZE']:.0f} KB x2 + WRITE_SIZE {c['WRITE_SIZE']:.0f} KB")
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
try:
    data = json.load(open(path))
except Exception:  # noqa: BLE001
    data = {}
data[key] = {"bytes_per_launch": int(total), "unit": unit, "unit_digest": _build.unit_digest(unit),
             "source": "; ".join(parts) + f" ({', '.join(os.path.relpath(f, ROOT) for f in files)}; separate rocprofv3 --pmc passes of tools/prof_run.py)"}
json.dump(data, open(path, "w"), indent=1)
print(key, data[key])
