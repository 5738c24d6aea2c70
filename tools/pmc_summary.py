#!/usr/bin/env python
"""Summarise a rocprofv3 counter_collection.csv for the gemv1 kernel: mean counter value per dispatch -> bytes.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section) -> doubled."""
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
kernel_filter = sys.argv[3] if len(sys.argv) > 3 else "gemv1p_kernel"
vals = []
with open(path) as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
        if kernel_filter in name and row.get("Counter_Name", row.get("Counter Name", "")) == counter:
            vals.append(float(row.get("Counter_Value", row.get("Counter Value", 0))))
if not vals:
    print(f"{counter}: no {kernel_filter} rows found in {path}")
    sys.exit(0)
if counter not in ("FETCH_SIZE", "WRITE_SIZE"):
    print(f"{counter}: kernel={kernel_filter} dispatches={len(vals)} mean={sum(vals) / len(vals):.4g} min={min(vals):.4g} max={max(vals):.4g}")
    sys.exit(0)
mean_kib = sum(vals) / len(vals)
corr = 2.0 if counter == "FETCH_SIZE" else 1.0
alg = {"decode_engine_kernel": "13361 MB of weights + ~69 MB of K/V rows at context ~134 (+ ~1.3 MB of mailbox granules per layer, swept by 256 CUs)"}.get(
    kernel_filter, "180.4 MB weights (+ KBs of activations / outputs)")
print(f"{counter}: dispatches={len(vals)} mean={mean_kib:.1f} KiB raw -> {mean_kib * 1024 * corr / 1e6:.1f} MB per launch "
      f"(x{corr:g} gfx950 correction); algorithmic = {alg}")
