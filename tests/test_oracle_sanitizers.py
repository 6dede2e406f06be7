"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`): every solver entry point is driven through the
instrumented library in a child process (libasan preloaded); any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
import numpy as np, sys
sys.path.insert(0, %(root)r)
import mapfree_reloc_amd
from mapfree_reloc_amd import synth
from oracle import oracle_lib as O
n_ok = 0
for seed, n, outl in ((1, 300, 0.3), (2, 64, 0.6), (3, 5, 0.0), (4, 4, 0.0), (5, 1100, 0.5)):
    p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.01, zero_depth_frac=0.05)
    for score in (O.EMAT_MAGSAC, O.EMAT_COUNT):
        e = O.emat_solve(p["pts0"], p["pts1"], p["K0"], p["K1"], 2.0, 0.9999, 300, 0, seed, want_counts=True, score=score)
        if e["status"] == 0:
            sc = O.scale_lift(p["pts0"], p["pts1"], e["mask"], p["depth0"], p["depth1"], p["K0"], p["K1"], e["R"], e["t"])
            O.scale_ransac(sc, 0.1); n_ok += 1
    O.pnp_solve(p["pts0"], p["pts1"], p["depth0"], p["K0"], p["K1"], 300, 3.0, 0.9999, 0, seed)
    O.procrustes_solve(p["pts0"], p["pts1"], p["depth0"], p["depth1"], p["K0"], p["K1"], 0.05, 0.999, 512, 0, seed)
rng = np.random.default_rng(0)
d0 = rng.integers(0, 120, (200, 128)).astype(np.float32); d1 = rng.integers(0, 120, (180, 128)).astype(np.float32)
O.sift_ratio_match(d0, d1, rng.random((200, 2)).astype(np.float32) * 500, rng.random((180, 2)).astype(np.float32) * 500, 0.8)
assert n_ok >= 4
print("SANITIZED_OK", n_ok)
'''


def test_oracle_entry_points_under_asan_ubsan():
    so = os.path.join(ROOT, "oracle", "_build", "libmfr_oracle_asan.so")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(so):
        pytest.skip("sanitizer build unavailable here: " + (r.stderr or "")[-200:])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               MFR_ORACLE_SO=so)
    p = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "SANITIZED_OK" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])
