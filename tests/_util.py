"""Shared helpers for the tests: seeded inputs (same recipe as tests/golden/make_golden.py)
and golden-digest comparison."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {  # tag -> (phi, size, batch, seed, input kind)
    "phi0_s256_b2_seed0": (0, 256, 2, 0, "normal"),
    "phi0_s256_b1_seed1": (0, 256, 1, 1, "uniform"),
    "phi3_s512_b1_seed0": (3, 512, 1, 0, "normal"),
}
CAMS = np.array([[480, 480, 128, 128, 1000, 1.0],
                 [572.4114, 573.57043, 325.2611, 242.04899, 1000, 0.8]], dtype=np.float32)


def seeded_input(shape, seed, kind="normal"):
    rng = np.random.Generator(np.random.PCG64([seed, 0x1234]))
    a = rng.standard_normal(shape) if kind == "normal" else rng.random(shape)
    return a.astype(np.float32)


def golden_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


def golden_case(tag):
    return golden_meta()[tag], np.load(os.path.join(GOLDEN, f"net_{tag}.npz"))


def strides_for(size, key):
    if key.startswith("trace_"):
        return 1009 if size == 256 else 8191
    return 97 if size == 256 else 397


def check_digest(name, arr, info, ref_slice, stride, atol, rtol=0.0):
    """Compare a full tensor against the committed strided slice (elementwise) and the
    float64 sum / abs-sum (global)."""
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    assert list(np.shape(arr)) == info["shape"], (name, np.shape(arr), info["shape"])
    got = a[::stride]
    err = np.abs(got.astype(np.float64) - ref_slice.astype(np.float64))
    tol = atol + rtol * np.abs(ref_slice.astype(np.float64))
    assert np.all(err <= tol), f"{name}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|) at {int(err.argmax())}"
    n = a.size
    s = float(a.astype(np.float64).sum())
    sa = float(np.abs(a.astype(np.float64)).sum())
    assert abs(sa - info["abssum"]) <= n * atol + rtol * info["abssum"] + 1e-6, (name, sa, info["abssum"])
    assert abs(s - info["sum"]) <= n * atol + rtol * info["abssum"] + 1e-6, (name, s, info["sum"])
    return float(err.max())
