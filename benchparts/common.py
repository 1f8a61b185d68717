"""Shared helpers of the benchmark.

Part of bench.py (the repo-root benchmark driver), split out in round 6: bench.py keeps the command line, the timed regions
of the headline metric and the assembly of the ONE JSON line; this module holds the source digest that ties PMC traffic figures to a build, logging and the algorithmic bytes per stage."""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def sources_digest() -> str:
    """SHA-256 over the library's sources (csrc/*, include/gsraster.h).  profiles/collect_r05.sh stores it next to the PMC
    counters it collects; a bench line quotes those counters as `roofline.traffic` only while the digest still matches --
    a kernel change silently keeping the old traffic figure was possible before."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "3d-gaussian-splat-attack_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "gsraster.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def log(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def stage_bytes(P: int, V: int, N: int, HW: int) -> dict:
    """Algorithmic bytes per launch of each stage (SURVEY.md section 8d derivation; objects off)."""
    return {
        "preprocess": 48 * P + 240 * V,          # K1: 44P in + 192V SH in, 48V geometry + 4P radii out
        "depth_sort": 0,                         # (the reference's single 64-bit pair sort is priced under tile_sort)
        "bin": 8 * P + 12 * N,                   # K2 scan + K3 emit
        "tile_sort": 24 * N,                     # K4 counted as ONE read + one write of the pairs (lower bound)
        "render_fwd": 40 * N + 20 * HW,          # K6
        "render_bwd": 20 * HW + 40 * N + 36 * V,  # K7
        "preprocess_bwd": 272 * V + 248 * P,     # K8+K9
    }
