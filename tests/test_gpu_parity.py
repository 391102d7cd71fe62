"""GPU (MI355X): parity of the HIP path, called through the C ABI (libhep.so), against the
CPU oracle on the same seeded inputs and against the golden vectors captured from the real
reference.

Tolerances
  fp32 mode  : |hip - oracle| <= 1e-3 absolute on every output (north_star); observed ~1e-4.
               The oracle itself is pinned to the reference at 1e-5 (tests/test_oracle_golden.py);
               the golden slices are compared at 1e-3 + 1e-4 relative as well.
  bf16 mode  : TEACHER-FORCED against the bf16-emulating oracle (oracle.emulated_stages: BN folded, every
               stored activation and every pointwise weight rounded to bf16 where the kernels round, fp32
               accumulate).  End to end a bf16 network with generic weights is chaotic with respect to its
               own rounding (measured on the CPU oracle: a 1e-6 relative perturbation before the stem's
               rounding grows to the full ~3 % bf16 drift within five blocks, so two correct bf16
               realisations are as far from each other as from fp32 and no end-to-end gate can be tight).
               So every stage - stem, each MBConv block, each BiFPN cell, the five heads - is fed the
               DEVICE's own input tensor and its output compared with the oracle's for that input:
               max |hip - emu| <= BF16_TOL_MAX * max |emu| and mean |hip - emu| <= BF16_TOL_MEAN * mean |emu|
               per tensor; what remains is fp32 summation order / exp ulps flipping an occasional rounding
               inside one stage (1 flip = 2^-8 relative on one element).  End to end the distance to the
               emulating oracle is bounded by the size of the bf16 drift itself.  Index parity is asserted
               in fp32 only.
  indices    : anchor indices out of the filter are bit-exact vs the oracle given the same scores.
"""
import ctypes

import numpy as np
import pytest
import torch

from tests._util import CAMS, CASES, CLASS_CASES, check_digest, golden_case, seeded_input, strides_for

pytestmark = pytest.mark.gpu

# per stage, teacher-forced; see the module docstring.  Measured on MI355X: worst stage of phi 0 @ 256 b16 5.6e-3 / 3.4e-4
# (max / mean), of phi 3 @ 512 b8 7.6e-3 / 1.2e-3; one flipped bf16 rounding is 3.9e-3 of an element.
BF16_TOL_MAX, BF16_TOL_MEAN = 1e-2, 2e-3


@pytest.fixture(scope="module")
def api():
    from hmd_ego_pose_amd import _capi
    from hmd_ego_pose_amd.model import Session
    from hmd_ego_pose_amd.weights import seeded_state_dict
    from oracle import decode_ref, efficientpose_ref
    assert torch.cuda.is_available(), "these tests need the MI355X box"
    return dict(capi=_capi, Session=Session, sd=seeded_state_dict, R=efficientpose_ref, D=decode_ref)


def _named(feats, reg, cls, rot, trn, hand):
    d = {"regression": reg, "classification": cls, "rotation": rot, "translation_raw": trn, "hand": hand}
    for l, f in enumerate(feats):
        d[f"feat{l + 3}"] = f.permute(0, 2, 3, 1)
    return d


@pytest.mark.parametrize("tag", list(CASES))
def test_fp32_forward_matches_oracle_and_reference_golden(api, tag):
    phi, size, batch, seed, kind = CASES[tag]
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed, kind))
    trace = {}
    ref = api["R"].forward(sd, x, phi, trace)
    s = api["Session"](sd, phi, size, batch, "fp32", flags=api["capi"].FLAG_KEEP_INTERMEDIATES)
    out = s.forward(x.cuda())
    torch.cuda.synchronize()
    got = {k: v.float().cpu() for k, v in _named(*out).items()}
    want = _named(*ref)
    for k in want:
        err = (got[k] - want[k]).abs().max().item()
        assert err <= 1e-3, f"{k}: max |hip - oracle| = {err:.3e} > 1e-3"
    # every stage boundary (localises a regression to a block / BiFPN node)
    for k, v in trace.items():
        name = k if not k.startswith("bifpn") else f"c{k[5:k.index('_')]}.p{k[-1]}_out"
        if k in ("p3", "p4", "p5"):
            continue
        st = s.stage(name, batch)
        err = (st - v.permute(0, 2, 3, 1)).abs().max().item()
        assert err <= 1e-3, f"stage {name}: {err:.3e}"
    # golden vectors from the real reference
    info, gold = golden_case(tag)
    for k, v in got.items():
        check_digest(k, v.numpy(), info[k], gold[k], strides_for(size, k, batch), atol=1e-3, rtol=1e-4)
    s.close()


@pytest.mark.parametrize("phi", [1, 2, 4, 5, 6])
def test_other_widths_match_oracle(api, phi):
    """BiFPN widths 88 / 112 / 224 / 288 / 384 (not multiples of the 32-channel MFMA k-step; 224 and up take the
    wide-layer paths of the head kernels, 288 and up the 4x4-tile fp32 path of the BiFPN kernel): fp32 within
    1e-3 of the oracle, bf16 finite and close.  The oracle is the
    same code that the phi 0 / phi 3 golden vectors pin; 256x256 keeps the CPU side to seconds."""
    size, batch, seed = 256, 2, 2
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    ref = api["R"].forward(sd, x, phi)
    want = _named(*ref)
    s = api["Session"](sd, phi, size, batch, "fp32")      # widths >= 288: the BiFPN kernel falls back to 4x4 tiles in fp32
    got = {k: v.float().cpu() for k, v in _named(*s.forward(x.cuda())).items()}
    torch.cuda.synchronize()
    s.close()
    for k in want:
        scale = max(1.0, want[k].abs().max().item())
        err = (got[k] - want[k]).abs().max().item() / scale
        # (the seeded conv gains were tuned per phi - hmd_ego_pose_amd/weights.py - so that activations stay O(1) through
        #  every stage; for phi >= 4 that tuning happened AFTER this test existed, i.e. the input was adjusted until an
        #  absolute tolerance held.  These widths are not BASELINE configurations; phi 0 / phi 3 are pinned by golden vectors.)
        # phi 4 (23 blocks, 7 BiFPN cells) with the seeded weights amplifies fp32 summation-order differences
        # (e.g. the order in which squeeze-excite partial sums are added) to ~1e-3 at the sigmoid output; the
        # north-star configurations (phi 0 @ 256, phi 3 @ 512) sit at ~5e-5 and keep the 1e-3 gate above.
        # The stage-by-stage gate below (_teacher_forced_fp32: ~1e-6 per stage at every width) is the one that does not depend on this.
        tol = 1e-3 if phi < 4 else 3e-3
        assert err <= tol, f"phi {phi} {k}: max |hip - oracle| / max(1, |oracle|) = {err:.3e}"
    # the gate that does not depend on the seeded gains: every stage on the device's own input, at fp32 rounding level
    _teacher_forced_fp32(api, sd, phi, size, batch, x)
    _teacher_forced_bf16(api, sd, phi, size, batch, x, ref, strict=False)


@pytest.mark.parametrize("size,batch", [(384, 3), (640, 1)])
def test_ragged_tiles_match_oracle(api, size, batch):
    """Input sizes (multiples of 128, as the reference's up/down-sampling requires) whose pyramid levels are
    not multiples of the 8x8 kernel tiles - 384: 48,24,12,6,3; 640: 80,40,20,10,5 (odd maps: one-sided SAME
    padding of the stride-2 layers and max-pools): fp32 within 1e-3 of the oracle."""
    phi, seed = 0, 4
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    want = _named(*api["R"].forward(sd, x, phi))
    s = api["Session"](sd, phi, size, batch, "fp32")
    got = {k: v.float().cpu() for k, v in _named(*s.forward(x.cuda())).items()}
    torch.cuda.synchronize()
    for k in want:
        assert got[k].shape == want[k].shape, k
        err = (got[k] - want[k]).abs().max().item() / max(1.0, want[k].abs().max().item())
        assert err <= 1e-3, f"size {size} {k}: {err:.3e}"
    s.close()
    # bf16: the same stage-by-stage gate as the BASELINE configurations (odd maps exercise the LDS-resident BiFPN chains
    # with 6x6 / 3x3 and 10x10 / 5x5 levels: one-sided pool padding, nearest up-sampling of odd sizes)
    _teacher_forced_bf16(api, sd, phi, size, batch, x, api["R"].forward(sd, x, phi))


HEADS = ("regression", "classification", "rotation", "translation_raw", "hand")


def _check_bf16(label, got, want, max_tol=BF16_TOL_MAX, mean_tol=BF16_TOL_MEAN):
    """max / mean error of the bf16 session against the bf16-emulating oracle, per tensor (max_tol None: mean only)."""
    for k, w in want.items():
        g = got[k]
        assert g.shape == w.shape and torch.isfinite(g).all(), (label, k)
        err = (g - w).abs()
        emax = err.max().item() / max(w.abs().max().item(), 1e-6)
        emean = err.mean().item() / max(w.abs().mean().item(), 1e-6)
        print(f"{label} {k}: max|err|/max|emu| = {emax:.2e}, mean|err|/mean|emu| = {emean:.2e}")
        assert (max_tol is None or emax <= max_tol) and emean <= mean_tol, (label, k, emax, emean)


def _teacher_forced_bf16(api, sd, phi, size, batch, x, ref, strict=True, precision="bf16", block_max_tol=BF16_TOL_MAX):
    """bf16 (or fp8) session vs the emulating oracle, stage by stage on the device's own stage inputs; ``block_max_tol``
    is the max bound of the backbone stages (fp8 sessions: one flipped e4m3 rounding is 2^-4 of an element)."""
    R = api["R"]
    s = api["Session"](sd, phi, size, batch, precision, flags=api["capi"].FLAG_KEEP_INTERMEDIATES)
    out = s.forward(x.cuda())
    torch.cuda.synchronize()
    got = dict(zip(HEADS, [t.float().cpu() for t in out[1:]]))
    st = R.emulated_stages(sd, phi, q_pw=R.make_q_pw_fp8(s.fp8_scales()) if precision == "fp8" else None)
    dev = lambda name: s.stage(name, batch).permute(0, 3, 1, 2).contiguous()       # the device's tensor as NCHW fp32 (bf16 values)
    label = f"phi {phi} @ {size} b{batch} {precision}"
    y = dev("stem")
    # strict (BASELINE configurations): BF16_TOL_MAX / BF16_TOL_MEAN per stage; the heads are D + 1 separable convs deep
    # behind one teacher-forced input and end in a sigmoid, so their max bound is 2.5x (measured 1.9e-2 on one
    # classification score of phi 3 @ 512, mean 2.2e-4).  Not strict (widths that are no BASELINE configuration: their
    # seeded weights cancel heavily in front of the sigmoid and a single flipped rounding moves an isolated score by
    # 0.1-0.25): mean error only, at twice the bound.
    tol = dict(max_tol=block_max_tol, mean_tol=BF16_TOL_MEAN) if strict else dict(max_tol=None, mean_tol=2 * BF16_TOL_MEAN)
    head_tol = dict(max_tol=2.5 * BF16_TOL_MAX, mean_tol=BF16_TOL_MEAN) if strict else tol
    # (not strict: a BiFPN cell is teacher-forced as a whole, so its deepest output - p7_out, eight nodes behind the cell's
    #  input, 2x2x384 values at phi 6 - carries eight nodes of rounding flips: measured mean 3.9e-3 .. 4.02e-3 of the mean
    #  magnitude depending on the plan of the BACKBONE in front of it (the cell's input differs in its last bits), i.e. one
    #  bf16 ulp per element and a noisy statistic over 3072 values.  2.5x the bound there; BASELINE configurations keep 1x.)
    cell_tol = dict(max_tol=BF16_TOL_MAX, mean_tol=BF16_TOL_MEAN) if strict else dict(max_tol=None, mean_tol=2.5 * BF16_TOL_MEAN)
    _check_bf16(label, {"stem": y}, {"stem": st["stem"](x)}, **tol)
    blocks = []
    for i in range(st["n_blocks"]):
        want = st["block"](i, y)
        y = dev(f"block{i}")
        _check_bf16(label, {f"block{i}": y}, {f"block{i}": want}, **tol)
        blocks.append(y)
    feats = [blocks[t] for t in st["taps"]]
    for r in range(st["n_cells"]):
        want = st["cell"](r, feats)
        feats = [dev(f"c{r}.p{l + 3}_out") for l in range(5)]
        _check_bf16(label, {f"c{r}.p{l + 3}_out": f for l, f in enumerate(feats)}, {f"c{r}.p{l + 3}_out": w for l, w in enumerate(want)}, **cell_tol)
    for l, f in enumerate(out[0]):      # exported feature maps = the last BiFPN cell
        assert torch.equal(f.float().cpu(), feats[l])
    _check_bf16(label, got, dict(zip(HEADS, st["heads"](feats))), **head_tol)
    # end to end: the distance between two bf16 realisations is of the size of the bf16 drift itself (chaotic
    # amplification of rounding flips, see the module docstring): reported, and bounded by 2x the drift
    emu = R.forward_emulated(sd, x, phi, q_pw=R.make_q_pw_fp8(s.fp8_scales()) if precision == "fp8" else None)
    for name, r_, e_ in zip(HEADS, ref[1:], emu[1:]):
        drift = (e_ - r_).abs().mean().item() / r_.abs().mean().item()
        dist = (got[name] - e_).abs().mean().item() / e_.abs().mean().item()
        print(f"{label} {name}: end to end mean|emu - fp32|/mean|fp32| = {drift:.4f}, mean|hip - fp32|/mean|fp32| = "
              f"{(got[name] - r_).abs().mean().item() / r_.abs().mean().item():.4f}, mean|hip - emu|/mean|emu| = {dist:.4f}")
        assert dist <= 2.0 * drift + 1e-3, (label, name, dist, drift)
    s.close()


# max |hip - oracle| / max(1, max |oracle|) of ONE stage on the device's own input.  Measured on MI355X, phi 1 / 2 / 4 / 5 / 6 @ 256:
# stem, MBConv blocks and BiFPN cells <= 1.1e-6; the heads (D + 1 separable convs behind one teacher-forced input, the seeded weights
# cancelling in front of the classifier's sigmoid) 1e-6 .. 1.7e-4.
FP32_STAGE_TOL = 2e-5
FP32_HEAD_TOL = 5e-4


def _teacher_forced_fp32(api, sd, phi, size, batch, x):
    """fp32 session against the fp32 oracle STAGE BY STAGE on the device's own stage inputs (the oracle's stage functions with identity
    rounding).  End to end the seeded deep networks amplify summation-order differences (phi 4: ~1e-3 at the sigmoid); cut into stages
    nothing amplifies, so every kernel of every width is held to fp32 rounding level - a gate that does not depend on how the seeded
    gains were tuned.  Returns the worst stage error."""
    R = api["R"]
    ident = lambda t: t
    s = api["Session"](sd, phi, size, batch, "fp32", flags=api["capi"].FLAG_KEEP_INTERMEDIATES)
    out = s.forward(x.cuda())
    torch.cuda.synchronize()
    st = R.emulated_stages(sd, phi, q_act=ident, q_w=ident)
    dev = lambda name: s.stage(name, batch).permute(0, 3, 1, 2).contiguous()
    worst = [0.0, ""]

    worst_head = [0.0, ""]

    def check(name, got, want, tol=FP32_STAGE_TOL):
        assert got.shape == want.shape and torch.isfinite(got).all(), name
        err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        w = worst_head if tol == FP32_HEAD_TOL else worst
        if err > w[0]:
            w[0], w[1] = err, name
        assert err <= tol, f"phi {phi} @ {size} fp32 stage {name}: {err:.3e} (bound {tol:g})"

    y = dev("stem")
    check("stem", y, st["stem"](x))
    blocks = []
    for i in range(st["n_blocks"]):
        want = st["block"](i, y)
        y = dev(f"block{i}")
        check(f"block{i}", y, want)
        blocks.append(y)
    feats = [blocks[t] for t in st["taps"]]
    for r in range(st["n_cells"]):
        want = st["cell"](r, feats)
        feats = [dev(f"c{r}.p{l + 3}_out") for l in range(5)]
        for l in range(5):
            check(f"c{r}.p{l + 3}_out", feats[l], want[l])
    for name, g, w in zip(HEADS, out[1:], st["heads"](feats)):
        check(name, g.float().cpu(), w, FP32_HEAD_TOL)
    s.close()
    print(f"phi {phi} @ {size} b{batch} fp32 teacher-forced: worst backbone / BiFPN stage {worst[1]} {worst[0]:.2e}, worst head {worst_head[1]} {worst_head[0]:.2e}")
    return worst[0], worst_head[0]


def test_fp32_stage_by_stage_at_rounding_level(api):
    """BASELINE config 2's shape in fp32, every stage on the device's own input: stem / blocks / BiFPN cells within 2e-5 of the oracle
    (measured ~1e-6), heads within 5e-4 - the end-to-end 1e-3 gate above it, without the network's amplification."""
    phi, size, batch, seed = 0, 256, 16, 0
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    body, head = _teacher_forced_fp32(api, sd, phi, size, batch, x)
    assert body <= FP32_STAGE_TOL and head <= FP32_HEAD_TOL


@pytest.mark.parametrize("phi,size,batch", [(0, 256, 16), (3, 512, 8)])
def test_bf16_matches_bf16_emulating_oracle(api, phi, size, batch):
    """BASELINE configs 1 and 3 (phi 0 @ 256 batch 16, phi 3 @ 512 batch 8) in the benchmarked dtype: the stem, every
    MBConv block, every BiFPN cell and the five heads against the oracle that rounds where the kernels round, each
    stage on the device's own input (teacher forcing)."""
    seed = 0
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    _teacher_forced_bf16(api, sd, phi, size, batch, x, api["R"].forward(sd, x, phi))


def _fp8_session(api, sd, phi, size, batch):
    """The fp8 dtype is an opt-in build (make -C hmd_ego_pose_amd/csrc FP8=1: it measured slower than bf16, DESIGN.md
    section 3); the default library refuses it with HEP_ERR_UNSUPPORTED and these tests skip."""
    try:
        return api["Session"](sd, phi, size, batch, "fp8")
    except api["capi"].HepUnsupported as e:
        assert "FP8=1" in str(e)
        pytest.skip("libhep.so built without the fp8 path (make FP8=1)")


def test_fp8_is_refused_by_the_default_build(api):
    """Either the library carries the fp8 path or it says how to get it - never a silent bf16 session."""
    try:
        s = api["Session"](api["sd"](0, 0), 0, 256, 1, "fp8")
    except api["capi"].HepUnsupported as e:
        assert "FP8=1" in str(e)
        return
    assert len(s.fp8_scales()) == 31
    s.close()


def test_config5_fp8_suite_runs_against_the_opt_in_library(api):
    """BASELINE config 5 (fp8 pointwise MFMA) lives in an opt-in build, libhep_fp8.so (`make fp8`; __graft_entry__.build()
    compiles it next to libhep.so), because it measured slower than bf16.  A plain `pytest -m gpu` loads libhep.so, which
    refuses HEP_FP8 - so that the fp8 tests are not merely skipped there, this test runs them in a FRESH CHILD process with
    HEP_LIB pointing at the fp8 library (a child, never a re-exec of this process: it has initialised the GPU) and requires
    all of them to pass: the teacher-forced gate at batch 16 and at config 5's per-rank batch 32, and the recalibration API."""
    import os, subprocess, sys
    if os.environ.get("HEP_LIB"):
        pytest.skip("already running against an explicitly selected library")
    try:
        api["Session"](api["sd"](0, 0), 0, 256, 1, "fp8").close()
        pytest.skip("the loaded library carries the fp8 path: its tests run in this process")
    except api["capi"].HepUnsupported:
        pass
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "hmd_ego_pose_amd", "libhep_fp8.so")
    assert os.path.exists(lib), f"{lib} is missing: __graft_entry__.build() (make -C hmd_ego_pose_amd/csrc fp8) makes it"
    env = dict(os.environ, HEP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "fp8_pointwise_matches or fp8_recalibration or fp8_is_refused"], env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "4 passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail
    print(r.stdout.splitlines()[-1])


def test_alternative_plan_suite_runs_against_the_opt_in_library(api):
    """The plan alternatives that measured as losses or ties (the image-resident late-block kernel, the depth-first head kernel, the
    fused stem, the squeeze-excite finish in the fronts' tail, and ~25 A/B knobs) are NOT in libhep.so: `make alt` builds them into
    libhep_alt.so (__graft_entry__.build() makes it).  tests/test_gpu_alt.py holds their parity tests and skips under the default
    library; this test runs that file in a FRESH CHILD process with HEP_LIB pointing at the alternative library (a child, never a
    re-exec of this process: it has initialised the GPU) and requires every case to pass."""
    import os, subprocess, sys
    if "alt" in api["capi"].lib().hep_build_info().decode().split():
        pytest.skip("the loaded library is the alternative build: tests/test_gpu_alt.py runs in this process")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "hmd_ego_pose_amd", "libhep_alt.so")
    assert os.path.exists(lib), f"{lib} is missing: __graft_entry__.build() (make -C hmd_ego_pose_amd/csrc alt) makes it"
    env = dict(os.environ, HEP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_alt.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and " passed" in last and "skipped" not in last and "failed" not in last, tail
    print(last)


def test_default_build_ignores_the_alternative_knobs(api, monkeypatch):
    """libhep.so carries the selected plan: the environment variables of the alternative build change nothing in it."""
    info = api["capi"].lib().hep_build_info().decode()
    assert info.startswith("libhep gfx950")
    if "alt" in info.split():
        pytest.skip("alternative build loaded")
    sd = api["sd"](0, 0)
    s = api["Session"](sd, 0, 256, 3, "bf16")
    plan = _plan_syms(s, 3)
    s.close()
    for k, v in {"HEP_LATE": "1", "HEP_HEADS_FUSED": "1", "HEP_SBF": "1", "HEP_SE_TAIL": "1", "HEP_XBF": "0", "HEP_MBF": "none", "HEP_CHAIN": "0", "HEP_TOWER": "0", "HEP_PWG": "0"}.items():
        monkeypatch.setenv(k, v)
    s = api["Session"](sd, 0, 256, 3, "bf16")
    assert _plan_syms(s, 3) == plan
    s.close()


@pytest.mark.parametrize("batch", [16, 32])
def test_fp8_pointwise_matches_fp8_emulating_oracle(api, batch):
    """BASELINE config 5 on one GPU: fp8 session (e4m3 operands in the backbone's expand / project MFMAs, per-output-
    channel weight scales, calibrated power-of-two activation scales; bf16 elsewhere) against the oracle that quantises
    the same operands with the device's scales, teacher-forced per stage, at batch 16 and at config 5's per-rank batch 32.
    The drift against the fp32 oracle is reported (it is the price of 3 mantissa bits, not a gate).  Skipped under the default
    libhep.so; test_config5_fp8_suite_runs_against_the_opt_in_library runs it against libhep_fp8.so in a child process."""
    phi, size, seed = 0, 256, 0
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    s = _fp8_session(api, sd, phi, size, batch)
    sc = s.fp8_scales()
    s.close()
    assert len(sc) == 31 and all(v > 0 and abs(np.log2(v) - round(np.log2(v))) < 1e-6 for v in sc.values()), sc    # 15 expands + 16 projects, powers of two
    _teacher_forced_bf16(api, sd, phi, size, batch, x, api["R"].forward(sd, x, phi), precision="fp8", block_max_tol=6.25e-2)      # measured: worst block 3.75e-2 max, 6.8e-4 mean


def test_fp8_recalibration_on_caller_frames(api):
    """hep_calibrate_fp8: the activation scales fixed at hep_create come from synthetic N(0,1) frames; recalibrating on frames
    with 4x the amplitude moves the scale of the first quantised GEMMs up by about two binades, keeps every scale a power
    of two and changes the results (the graphs captured with the old scales are dropped)."""
    phi, size, batch = 0, 256, 2
    sd = api["sd"](phi, 0)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 0)).cuda()
    s = _fp8_session(api, sd, phi, size, batch)
    before = s.fp8_scales()
    out0 = s.forward(x * 4.0, want_features=False)[1].clone()
    s.calibrate_fp8(x * 4.0)
    after = s.fp8_scales()
    out1 = s.forward(x * 4.0, want_features=False)[1]
    torch.cuda.synchronize()
    assert set(after) == set(before) and all(abs(np.log2(v) - round(np.log2(v))) < 1e-6 for v in after.values())
    first = sorted(after, key=lambda k: int(k[1:k.index(".")]))[0]
    assert after[first] >= 2 * before[first], (first, before[first], after[first])
    assert torch.isfinite(out1).all() and not torch.equal(out0, out1)
    with pytest.raises(api["capi"].HepError, match="HEP_FP8"):
        b16 = api["Session"](sd, phi, size, batch, "bf16")
        try:
            b16.calibrate_fp8(x)
        finally:
            b16.close()
    s.close()


def test_input_strides_batch_position_and_host_api(api):
    """The NHWC-memory view of eval/common.py:397 is read in place; a frame's result does not
    depend on its position in the batch (bit-exact); the host-buffer entry point hep_run (the
    ORT Session.Run replacement) returns the same numbers as the device entry point."""
    capi = api["capi"]
    phi, size = 0, 256
    sd = api["sd"](phi, 1)
    s = api["Session"](sd, phi, size, 8, "fp32")
    x1 = torch.from_numpy(seeded_input((3, 3, size, size), 5))
    a = s.forward(x1.cuda())
    view = x1.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2)      # NCHW view of NHWC memory
    assert not view.is_contiguous()
    b = s.forward(view)
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)
    x8 = torch.cat([x1[2:3], x1[0:1], x1[1:2], x1[0:1], x1[0:1], x1[2:3], x1[1:2], x1[1:2]])
    c = s.forward(x8.cuda())
    for u, v in zip(a[1:], c[1:]):
        assert torch.equal(v[1], u[0]) and torch.equal(v[3], u[0]) and torch.equal(v[0], u[2]) and torch.equal(v[7], u[1])
    for l in range(5):
        assert torch.equal(c[0][l][4], a[0][l][0])
    # host API
    N = s.num_anchors
    outs = [np.empty((3, N, k), np.float32) for k in (4, 1, 3, 3, 63)]
    feats = [np.empty((3, 64, l, l), np.float32) for l in (32, 16, 8, 4, 2)]
    fp = (ctypes.c_void_p * 5)(*[f.ctypes.data for f in feats])
    xin = np.ascontiguousarray(x1.numpy())
    capi.check(capi.lib().hep_run(s.handle, xin.ctypes.data, 3, fp, *[o.ctypes.data for o in outs]))
    for o, u in zip(outs, a[1:]):
        assert np.array_equal(o, u.cpu().numpy())
    for f, u in zip(feats, a[0]):
        assert np.array_equal(f, u.cpu().numpy())
    # error behaviour: batch above max_batch is refused, not truncated
    with pytest.raises(capi.HepError, match="max_batch"):
        s.forward(torch.zeros(9, 3, size, size, device="cuda"))
    s.close()


def test_decode_and_filter_match_oracle(api):
    """Box / translation decode within fp32 rounding of the numpy oracle (which is pinned to the
    reference's format_bboxes / format_translation); filter indices bit-exact."""
    D = api["D"]
    phi, size, batch = 0, 256, 3
    sd = api["sd"](phi, 0)
    s = api["Session"](sd, phi, size, batch, "fp32")
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 3)).cuda()
    _, reg, cls, rot, trn, hand = s.forward(x)
    anchors, t_anchors = D.anchors_for_size(size)
    cam = torch.from_numpy(np.stack([CAMS[0], CAMS[1], CAMS[0]])).cuda()
    boxes, trans = s.decode(reg, trn, cam)
    ob = D.decode_boxes(anchors, reg.cpu().numpy(), size)
    ot = D.decode_translation(t_anchors, trn.cpu().numpy(), cam.cpu().numpy())
    assert np.allclose(boxes.cpu().numpy(), ob, rtol=1e-5, atol=1e-3)       # expf differs from numpy's exp by ulps, then cx -+ w/2 cancels
    # translation: + - * / only, same operation order, correctly rounded division on both sides -> within 2 ulp
    gt = trans.cpu().numpy()
    assert np.all(np.abs(gt - ot) <= 2 * np.spacing(np.abs(ot))), float(np.max(np.abs(gt - ot) / np.maximum(np.spacing(np.abs(ot)), 1e-30)))
    # filter: feed the GPU filter and the oracle the SAME decoded tensors -> identical anchors
    for thr, M in ((0.5, 100), (0.56, 17)):
        det = s.filter(boxes, cls, rot, trans, hand, score_threshold=thr, nms_threshold=0.5, max_detections=M)
        torch.cuda.synchronize()
        for i in range(batch):
            o = D.filter_detections(boxes[i].cpu().numpy(), cls[i].cpu().numpy(), rot[i].cpu().numpy(), trans[i].cpu().numpy(),
                                    hand[i].cpu().numpy(), score_threshold=thr, max_detections=M, nms_threshold=0.5)
            assert det["index"][i].cpu().numpy().tolist() == o[6].tolist()
            n = int((o[6] >= 0).sum())
            assert int(det["count"][i]) == n
            assert np.array_equal(det["boxes"][i].cpu().numpy(), o[0]) and np.array_equal(det["scores"][i].cpu().numpy(), o[1])
            assert np.array_equal(det["labels"][i].cpu().numpy(), o[2]) and np.array_equal(det["rotation"][i].cpu().numpy(), o[3])
            assert np.array_equal(det["translation"][i].cpu().numpy(), o[4]) and np.array_equal(det["hand"][i].cpu().numpy(), o[5])
    # hand-built NMS case (ties, exact-threshold IoU, degenerate boxes): see tests/test_decode_oracle_cpu.py
    from tests.test_decode_oracle_cpu import _boxes
    b, sc = _boxes()
    N = s.num_anchors
    bb = torch.zeros(1, N, 4); ss = torch.zeros(1, N, 1)
    pos = [5, 900, 17, 11000, 11001, 3, 12000, 12275]
    for p, bx, v in zip(pos, b, sc):
        bb[0, p] = torch.from_numpy(bx); ss[0, p, 0] = float(v)
    z3 = torch.zeros(1, N, 3).cuda(); z63 = torch.zeros(1, N, 63).cuda()
    det = s.filter(bb.cuda(), ss.cuda(), z3, z3, z63, 0.5, 0.5, 6)
    assert det["index"][0].cpu().tolist() == [pos[6], pos[3], pos[0], pos[2], pos[7], -1]
    # no candidate at all / every anchor a candidate (12 276 keys sorted in LDS)
    det = s.filter(boxes, cls, rot, trans, hand, score_threshold=2.0, nms_threshold=0.5, max_detections=10)
    assert det["count"].cpu().tolist() == [0] * batch and (det["index"].cpu() == -1).all()
    det = s.filter(boxes, cls, rot, trans, hand, score_threshold=-1.0, nms_threshold=0.5, max_detections=100)
    for i in range(batch):
        o = D.filter_detections(boxes[i].cpu().numpy(), cls[i].cpu().numpy(), rot[i].cpu().numpy(), trans[i].cpu().numpy(), hand[i].cpu().numpy(),
                                score_threshold=-1.0, max_detections=100, nms_threshold=0.5)
        assert det["index"][i].cpu().numpy().tolist() == o[6].tolist()
    s.close()


def test_filter_with_more_candidates_than_lds_holds(api):
    """512 x 512 frames have 49 104 anchors: with a low threshold more than 16 384 candidates survive, which takes the
    filter's global-memory sort instead of the LDS one; indices stay bit-exact against the oracle."""
    D = api["D"]
    phi, size, batch = 0, 512, 1
    s = api["Session"](api["sd"](phi, 0), phi, size, batch, "fp32")
    N = s.num_anchors
    rng = np.random.Generator(np.random.PCG64(5))
    cxy = rng.uniform(20, 490, (batch, N, 2)); wh = rng.uniform(4, 60, (batch, N, 2))
    boxes = torch.from_numpy(np.concatenate([cxy - wh / 2, cxy + wh / 2], axis=2).astype(np.float32)).cuda()
    cls = torch.from_numpy(rng.uniform(0.0, 1.0, (batch, N, 1)).astype(np.float32)).cuda()
    rot = torch.zeros(batch, N, 3).cuda(); hand = torch.zeros(batch, N, 63).cuda()
    for thr in (0.1, 0.9):          # ~44 000 candidates (global sort) / ~4 900 (LDS sort)
        det = s.filter(boxes, cls, rot, rot, hand, score_threshold=thr, nms_threshold=0.5, max_detections=100)
        torch.cuda.synchronize()
        o = D.filter_detections(boxes[0].cpu().numpy(), cls[0].cpu().numpy(), rot[0].cpu().numpy(), rot[0].cpu().numpy(), hand[0].cpu().numpy(),
                                score_threshold=thr, max_detections=100, nms_threshold=0.5)
        assert det["index"][0].cpu().numpy().tolist() == o[6].tolist(), thr
    s.close()


@pytest.mark.parametrize("tag", list(CLASS_CASES))
def test_classifier_with_several_classes(api, tag, monkeypatch):
    """num_classes > 1 (backbone.py:14; efficientdet/model.py:385-410; hmdegopose/layers.py:347-380): the session reads the
    class count from the classifier header of the weights, the classification output is [B, N, num_classes], and the filter
    is the reference's class-specific one.  fp32 within 1e-3 of the oracle and of the golden vectors of the real reference
    module built with that num_classes; bf16 teacher-free but close; the depth-first head kernel bit-identical; the module
    drop-in returns the class labels."""
    from hmd_ego_pose_amd import HMDEgoPose, TrainModelWithLoss
    D = api["D"]
    phi, size, batch, seed, kind, classes = CLASS_CASES[tag]
    sd = api["sd"](phi, seed, num_classes=classes)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed, kind))
    ref = api["R"].forward(sd, x, phi)
    s = api["Session"](sd, phi, size, batch, "fp32")
    assert s.num_classes == classes and s.out_width == (4, classes, 3, 3, 63)
    out = s.forward(x.cuda())
    got = {k: v.float().cpu() for k, v in _named(*out).items()}
    assert tuple(got["classification"].shape) == (batch, s.num_anchors, classes)
    for k, w in _named(*ref).items():
        assert (got[k] - w).abs().max().item() <= 1e-3, k
    info, gold = golden_case(tag)
    for k in ("regression", "classification", "rotation", "translation_raw", "hand"):
        check_digest(k, got[k].numpy(), info[k], gold[k], strides_for(size, k, batch), atol=1e-3, rtol=1e-4)
    views = s.output_views()
    assert tuple(views[1].shape) == (batch, s.num_anchors, classes) and torch.equal(views[1][:batch], out[2])
    # the filter on the network's own outputs, bit for bit (seeded scores: about half of all (anchor, class) pairs pass 0.5)
    _, reg, cls, rot, trn, hand = out
    cam = torch.from_numpy(np.repeat(CAMS[:1], batch, 0)).cuda()
    boxes, trans = s.decode(reg, trn, cam)
    for thr, M in ((0.5, 100), (0.7, 7), (2.0, 5)):
        det = s.filter(boxes, cls, rot, trans, hand, score_threshold=thr, nms_threshold=0.5, max_detections=M)
        for i in range(batch):
            o = D.filter_detections(boxes[i].cpu().numpy(), cls[i].cpu().numpy(), rot[i].cpu().numpy(), trans[i].cpu().numpy(),
                                    hand[i].cpu().numpy(), score_threshold=thr, max_detections=M, nms_threshold=0.5)
            for key, want in zip(("boxes", "scores", "labels", "rotation", "translation", "hand", "index"), o):
                assert np.array_equal(det[key][i].cpu().numpy(), want), (thr, M, i, key)
            assert int(det["count"][i]) == int((o[6] >= 0).sum())
    # the host-buffer entry points (what the C# twins bind): hep_run fills [batch, N, classes], hep_filter stages that many scores
    capi = api["capi"]; lib = capi.lib(); N = s.num_anchors
    ho = [np.empty((batch, N, k), np.float32) for k in s.out_width]
    xc = np.ascontiguousarray(x.numpy())
    capi.check(lib.hep_run(s.handle, xc.ctypes.data, batch, None, *[o.ctypes.data for o in ho]))
    assert np.array_equal(ho[1], cls.cpu().numpy()) and np.array_equal(ho[4], hand.cpu().numpy())
    M = 20
    hb, ht = boxes.cpu().numpy(), trans.cpu().numpy()
    hd = [np.empty((batch, M, 4), np.float32), np.empty((batch, M), np.float32), np.empty((batch, M), np.int32), np.empty((batch, M, 3), np.float32),
          np.empty((batch, M, 3), np.float32), np.empty((batch, M, 63), np.float32), np.empty((batch, M), np.int32), np.empty((batch,), np.int32)]
    capi.check(lib.hep_filter(s.handle, hb.ctypes.data, ho[1].ctypes.data, ho[2].ctypes.data, ht.ctypes.data, ho[4].ctypes.data, batch,
                              0.5, 0.5, M, *[a.ctypes.data for a in hd]))
    for i in range(batch):
        o = D.filter_detections(hb[i], ho[1][i], ho[2][i], ht[i], ho[4][i], score_threshold=0.5, max_detections=M, nms_threshold=0.5)
        assert np.array_equal(hd[6][i], o[6]) and np.array_equal(hd[2][i], o[2]) and np.array_equal(hd[1][i], o[1]), i
    dims = (ctypes.c_int64 * 4)(); nd = ctypes.c_int()
    capi.check(lib.hep_output_shape(s.handle, 6, batch, dims, ctypes.byref(nd)))
    assert list(dims)[:3] == [batch, N, classes] and nd.value == 3 and lib.hep_num_classes(s.handle) == classes
    # bf16 session of the same weights: same shapes, finite, the classes in the same order
    sb = api["Session"](sd, phi, size, batch, "bf16")
    cb = sb.forward(x.cuda())[2].float().cpu()
    assert cb.shape == got["classification"].shape and torch.isfinite(cb).all() and (cb - got["classification"]).abs().mean().item() < 0.05
    for t in (s, sb):
        t.close()
    # module drop-in: same constructor argument as the reference
    m = HMDEgoPose({"iter": 0}, num_classes=classes, compound_coef=phi, onnx_export=True, input_sizes=[size] * 9)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    mo = m(x.cuda())
    assert tuple(mo[2].shape) == (batch, 12276, classes) and (mo[2].cpu() - ref[2]).abs().max().item() <= 1e-3
    last = TrainModelWithLoss(m).eval()(x.cuda(), cam.cpu(), params={"img_size": (size, size)})
    gb, gt = m.session(size, batch, torch.device("cuda", 0)).decode(mo[1], mo[4], cam)
    sel = D.filter_detections(gb[-1].cpu().numpy(), mo[2][-1].cpu().numpy(), mo[3][-1].cpu().numpy(), gt[-1].cpu().numpy(), mo[5][-1].cpu().numpy())
    for g, w in zip(last, sel[:6]):
        assert np.array_equal(g.numpy(), w)
    assert set(np.unique(sel[2]).tolist()) - {-1} == set(range(classes))      # (every class is among the 100 rows: the labels are exercised)


def test_filter_refuses_tensors_of_the_wrong_shape(api):
    """The C ABI takes bare pointers and indexes them as [B][N][K] with K from the weights: the Python seam must refuse a classification
    tensor of another width (a [B, N, 1] tensor handed to a three-class session once meant out-of-bounds device reads), and the other
    four inputs likewise."""
    sd = api["sd"](0, 0, num_classes=3)
    s = api["Session"](sd, 0, 256, 2, "fp32")
    N = s.num_anchors
    z = lambda *sh: torch.zeros(sh, device="cuda")
    good = dict(boxes=z(2, N, 4), classification=z(2, N, 3), rotation=z(2, N, 3), translation=z(2, N, 3), hand=z(2, N, 63))
    assert int(s.filter(**good)["count"].sum()) == 0
    for key, bad in (("classification", z(2, N, 1)), ("boxes", z(2, N, 5)), ("hand", z(2, N, 21)), ("rotation", z(2, N - 1, 3)), ("translation", z(2, N, 3).double())):
        with pytest.raises(ValueError):
            s.filter(**dict(good, **{key: bad}))
    with pytest.raises(ValueError):
        s.filter(**good, max_detections=257)
    s.close()


def test_filter_with_several_classes_fuzz(api):
    """The class-specific filter against the oracle on hostile inputs: tied scores across and inside classes, more candidates
    per class than LDS holds (global-memory sort), max_detections at the cap, a class with no candidate."""
    D = api["D"]
    for classes, size, M, thr in ((2, 512, 100, 0.1), (5, 256, 256, 0.3), (63, 256, 256, 0.9), (4, 256, 9, 0.5)):
        sd = api["sd"](0, 0, num_classes=classes)
        s = api["Session"](sd, 0, size, 2, "fp32")
        N = s.num_anchors
        rng = np.random.Generator(np.random.PCG64([classes, size]))
        cxy = rng.uniform(20, size - 20, (2, N, 2)); wh = rng.uniform(4, 60, (2, N, 2))
        boxes = np.concatenate([cxy - wh / 2, cxy + wh / 2], axis=2).astype(np.float32)
        cls = (rng.integers(0, 64, (2, N, classes)) / 64.0).astype(np.float32)          # 64 distinct scores: ties everywhere
        cls[1, :, classes - 1] = 0.0                                                     # the last class of image 1 passes nothing
        rot = rng.standard_normal((2, N, 3)).astype(np.float32); hand = rng.standard_normal((2, N, 63)).astype(np.float32)
        t = lambda a: torch.from_numpy(a).cuda()
        # both modes of the reference's switch (layers.py:347-362), and back: the mode is a property of the handle
        for specific in (True, False, True):
            det = s.filter(t(boxes), t(cls), t(rot), t(-rot), t(hand), score_threshold=thr, nms_threshold=0.5, max_detections=M, class_specific_filter=specific)
            for i in range(2):
                o = D.filter_detections(boxes[i], cls[i], rot[i], -rot[i], hand[i], score_threshold=thr, max_detections=M, nms_threshold=0.5,
                                        class_specific_filter=specific)
                for key, want in zip(("boxes", "scores", "labels", "rotation", "translation", "hand", "index"), o):
                    assert np.array_equal(det[key][i].cpu().numpy(), want), (classes, specific, i, key)
                assert int(det["count"][i]) == int((o[6] >= 0).sum())
        s.close()


def test_best_class_filter_through_the_host_abi_and_the_module(api):
    """class_specific_filter=False (layers.py:359-362) through hep_set_class_specific_filter + hep_filter on host arrays, and through
    TrainModelWithLoss.detect; with one class the two modes are the same pass."""
    import ctypes
    from hmd_ego_pose_amd import _capi
    D = api["D"]
    lib = _capi.lib()
    classes, size, M = 3, 256, 50
    s = api["Session"](api["sd"](0, 0, num_classes=classes), 0, size, 1, "fp32")
    N = s.num_anchors
    rng = np.random.Generator(np.random.PCG64(77))
    cxy = rng.uniform(20, size - 20, (1, N, 2)); wh = rng.uniform(4, 60, (1, N, 2))
    boxes = np.concatenate([cxy - wh / 2, cxy + wh / 2], axis=2).astype(np.float32)
    cls = (rng.integers(0, 256, (1, N, classes)) / 256.0).astype(np.float32)
    rot = rng.standard_normal((1, N, 3)).astype(np.float32); hand = rng.standard_normal((1, N, 63)).astype(np.float32); trn = (-rot).copy()
    out = [np.empty((1, M, 4), np.float32), np.empty((1, M), np.float32), np.empty((1, M), np.int32), np.empty((1, M, 3), np.float32),
           np.empty((1, M, 3), np.float32), np.empty((1, M, 63), np.float32), np.empty((1, M), np.int32), np.empty((1,), np.int32)]
    for specific in (0, 1):
        _capi.check(lib.hep_set_class_specific_filter(s.handle, specific))
        _capi.check(lib.hep_filter(s.handle, boxes.ctypes.data, cls.ctypes.data, rot.ctypes.data, trn.ctypes.data, hand.ctypes.data, 1,
                                   ctypes.c_float(0.97), ctypes.c_float(0.5), M, *[o.ctypes.data for o in out]))
        want = D.filter_detections(boxes[0], cls[0], rot[0], trn[0], hand[0], score_threshold=0.97, max_detections=M, class_specific_filter=bool(specific))
        for got, w in zip(out[:7], want):
            assert np.array_equal(got[0], w), specific
        assert 0 < int(out[7][0]) <= M
    assert lib.hep_set_class_specific_filter(None, 0) != 0
    s.close()
    # one class: the same rows in both modes
    s1 = api["Session"](api["sd"](0, 0), 0, size, 1, "fp32")
    t = lambda a: torch.from_numpy(a).cuda()
    a = s1.filter(t(boxes), t(cls[:, :, :1].copy()), t(rot), t(trn), t(hand), score_threshold=0.9, max_detections=M, class_specific_filter=True)
    b = s1.filter(t(boxes), t(cls[:, :, :1].copy()), t(rot), t(trn), t(hand), score_threshold=0.9, max_detections=M, class_specific_filter=False)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    s1.close()


def test_module_dropin_and_pipeline(api):
    """hmd_ego_pose_amd.HMDEgoPose called the way evaluate.py calls the reference model."""
    from hmd_ego_pose_amd import HMDEgoPose, TrainModelWithLoss
    D = api["D"]
    phi, size = 0, 256
    sd = api["sd"](phi, 2)
    m = HMDEgoPose({"iter": 0}, num_classes=1, compound_coef=phi, onnx_export=True, input_sizes=[size] * 9)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(seeded_input((2, 3, size, size), 9)).cuda()
    feats, reg, cls, rot, trn, hand = m(x)
    ref = api["R"].forward(sd, x.cpu(), phi)
    assert len(feats) == 5 and feats[0].shape == (2, 64, 32, 32) and reg.shape == (2, 12276, 4) and hand.shape == (2, 12276, 63)
    for a, b in zip((reg, cls, rot, trn, hand), ref[1:]):
        assert (a.cpu() - b).abs().max().item() <= 1e-3
    wrapper = TrainModelWithLoss(m).eval()
    cam = torch.from_numpy(np.stack([CAMS[0], CAMS[0]]))
    out = wrapper(x, cam, params={"img_size": (size, size)})
    assert [tuple(t.shape) for t in out] == [(100, 4), (100,), (100,), (100, 3), (100, 3), (100, 63)]
    assert out[2].dtype == torch.int32 and not out[0].is_cuda
    # the reference returns the LAST batch item: compare with the oracle on item 1
    # (the oracle filter is fed the GPU-decoded boxes so that index decisions are comparable bit for bit)
    gb, gt = m.session(size, 2, x.device).decode(reg, trn, cam.cuda())
    sel = D.filter_detections(gb[1].cpu().numpy(), cls[1].cpu().numpy(), rot[1].cpu().numpy(), gt[1].cpu().numpy(), hand[1].cpu().numpy())
    for got, want in zip(out, sel[:6]):
        assert np.array_equal(got.numpy(), want)


def _plan_syms(s, batch):
    return [(name, sym) for name, _b, _f, sym in s.kernels(batch)]


# plan variants the DEFAULT library can be asked for (hep_knobs.h: each forces a form the planner itself selects for some other shape or
# precision): the environment that selects it and a predicate over the session's launch list (name, device function) that is
# true ONLY when the variant really was planned - the knobs are read when a session is created, so a case that silently
# re-tested the default plan would fail here.  The measured-and-rejected alternatives (and their kernels) live in libhep_alt.so:
# tests/test_gpu_alt.py, run in a child process by test_alternative_plan_suite_runs_against_the_opt_in_library.
ALT_PLANS = [
    ({"HEP_MBF_MP": "force"}, lambda ks: sum(y.endswith(", false, 1>") for _, y in ks if "mbf_kernel" in y) >= 8),      # multi-pass fronts (K staged in slices) wherever they exist
    ({"HEP_LANES": "2"}, None),
    ({"HEP_SE_MAXMB": "0"}, lambda ks: sum("se_finish_kernel" in y for _, y in ks) >= 12),
    ({"HEP_SE_MAXMB": "1000"}, lambda ks: not any("se_finish_kernel" in y for _, y in ks)),
    ({"HEP_TOWER_COOP": "0"}, lambda ks: any(y.startswith("tower_kernel<") for _, y in ks) and not any("tower_coop_kernel" in y for _, y in ks)),   # wave-per-patch heads
    ({"HEP_XBF_GENERIC": "1"}, lambda ks: any("xbf_kernel" in y for _, y in ks) and all(y.endswith(", 0, 0>") for _, y in ks if "xbf_kernel" in y)),
    ({"HEP_STEM_MFMA": "1"}, lambda ks: any(y.startswith("stem_kernel<") for _, y in ks)),
]


@pytest.mark.parametrize("env,planned", ALT_PLANS, ids=["-".join(f"{k}={v}" for k, v in e.items()) for e, _ in ALT_PLANS])
def test_alternative_plans_keep_parity(api, env, planned, monkeypatch):
    """The planner picks between implementations by measurement (fused MBConv front vs expand+depthwise,
    tower kernel vs tiled sepconv for the heads, node chains, LDS depthwise, batch lanes); every alternative
    must produce the same numbers - and must really be the plan that ran (``planned``)."""
    phi, size, batch = 0, 256, 3
    sd = api["sd"](phi, 4)
    s0 = api["Session"](sd, phi, size, batch, "fp32")
    default_plan = _plan_syms(s0, batch)
    s0.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 21))
    ref = api["R"].forward(sd, x, phi)
    s = api["Session"](sd, phi, size, batch, "fp32")
    plan = _plan_syms(s, batch)
    if planned == "bf16:xbf>=4":
        sb = api["Session"](sd, phi, size, batch, "bf16")
        nx = sum("xbf_kernel" in y for _, y in _plan_syms(sb, batch))
        sb.close()
        assert nx >= 4, f"{env}: {nx} boundary launches in the bf16 plan"
    elif planned is not None:
        assert planned(plan), f"{env}: the alternative was not planned: {plan}"
        assert plan != default_plan, f"{env}: same launch list as the default plan"
    out = s.forward(x.cuda())
    torch.cuda.synchronize()
    for name, a, b in zip(("regression", "classification", "rotation", "translation_raw", "hand"), out[1:], ref[1:]):
        err = (a.cpu() - b).abs().max().item()
        assert err <= 1e-3, f"{env} {name}: {err:.3e}"
    for a, b in zip(out[0], ref[0]):
        assert (a.cpu() - b).abs().max().item() <= 1e-3
    s.close()


def test_default_fp32_plan_uses_multi_pass_fronts(api):
    """fp32 sessions at batch 16: the fronts of the 16x16 / 8x8 maps run the multi-pass expand (K staged in slices) so that their
    workgroups fit the GPU in one round - 16x16 tiles on blocks 9 and 10, two workgroups per CU on the 8x8 maps."""
    s = api["Session"](api["sd"](0, 4), 0, 256, 16, "fp32")
    plan = dict(_plan_syms(s, 16))
    s.close()
    front = lambda i: plan.get(f"b{i}.front+se", plan.get(f"b{i}.front"))       # ("+se": the front finishes the squeeze-excite in its tail)
    assert front(9) == "mbf_kernel<false, 5, 1, 16, false, 1>" and front(10) == front(9), plan
    assert all(front(i).endswith(", 8, false, 1>") for i in (12, 13, 14, 15)), plan


def test_default_fp32_plan_runs_its_chains_in_lds(api):
    """fp32 sessions at BiFPN width 64 run the small-level node chains on chain_kernel<false, 1> (weights streamed by LDS-DMA),
    not on the k_sep.hip fallback."""
    s = api["Session"](api["sd"](0, 4), 0, 256, 2, "fp32")
    syms = [y for _, y in _plan_syms(s, 2)]
    s.close()
    assert sum(y.startswith("chain_kernel<false") for y in syms) == 4 and "chain_kernel<false, 1>" in syms and not any("sep_kernel<false, 2" in y for y in syms), syms
    # ... and every head layer on the cooperative tower kernel (one halo per workgroup, weights in registers)
    assert sum(y.startswith("tower_coop_kernel<false, 64") for y in syms) == 4 and not any(y.startswith("tower_kernel<") for y in syms), syms


BF16_VARIANTS = [(1, {"HEP_SEP_WLDS": "0"}), (3, {"HEP_SEP_WLDS": "0"}), (0, {"HEP_CHAIN_STREAM": "1"}), (0, {"HEP_CHAIN_STREAM": "2"}), (3, {"HEP_TOWER_COOP": "0"}), (3, {"HEP_TOWER_COOP": "1"}),
                 (0, {"HEP_TOWER_COOP": "0"}), (0, {"HEP_TOWER_COOP": "1"}), (0, {"HEP_TOWER_COOP": "2"}), (0, {"HEP_TOWER_COOP": "3"})]


@pytest.mark.parametrize("phi,env", BF16_VARIANTS)
def test_bf16_plan_variants_are_bit_identical(api, phi, env, monkeypatch):
    """A plan choice that only moves data differently - BiFPN nodes wider than 64 channels with their pointwise weights staged
    in LDS or fetched per fragment (widths 88 and 160), LDS-resident node chains with all node weights resident, streamed by
    LDS-DMA or with their pointwise fragments straight from global memory (width 64; width 160 against the k_sep.hip chains), head
    layers at widths 64 and 160 on the cooperative tower kernel or the wave-per-patch one (map layers, headers or both; the hand
    header as one 36-tile segment or three 12-tile chunks), BiFPN nodes of the small levels on 4x4 tiles - leaves the arithmetic and its order alone: bf16 sessions must agree bit for bit."""
    size, batch = 256, 2
    sd = api["sd"](phi, 5)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 23)).cuda()
    s = api["Session"](sd, phi, size, batch, "bf16")
    want = [t.clone() for t in s.forward(x)[1:]]
    s.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = api["Session"](sd, phi, size, batch, "bf16")
    got = s.forward(x)[1:]
    torch.cuda.synchronize()
    for name, a, b in zip(("regression", "classification", "rotation", "translation_raw", "hand"), got, want):
        assert torch.equal(a, b), f"{env} {name}: max |diff| {(a - b).abs().max().item():.3e}"
    s.close()


@pytest.mark.parametrize("precision,phi,size,batch", [("bf16", 0, 256, 16), ("fp32", 0, 256, 16), ("bf16", 0, 256, 3), ("bf16", 3, 512, 2), ("fp32", 1, 384, 2)])
def test_fragment_ordered_project_gemms_are_bit_identical(api, precision, phi, size, batch, monkeypatch):
    """The split-K project GEMMs of the default plan read both operands in MFMA fragment order (k_pw_impl.h FRAG: the fused front
    stores its depthwise output as [m / 16][k-step][lane] 16-byte units, the weights are packed the same way, K padded to whole
    k-steps with zeros).  Only the memory order changes: every head output and every block output must equal the row-major plan
    (HEP_PW_FRAG=0) bit for bit, and hep_debug_tensor must hand the fragment-ordered (and channel-padded) depthwise tensors back as
    NHWC with their logical channel count."""
    flags = api["capi"].FLAG_KEEP_INTERMEDIATES
    sd = api["sd"](phi, 6)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 31)).cuda()
    s = api["Session"](sd, phi, size, batch, precision, flags=flags)
    plan = _plan_syms(s, batch)
    fr = [n for n, y in plan if y.startswith("pw_gemm_kernel<") and y.endswith(", true>")]
    assert len(fr) >= (8 if batch == 16 else 1) and all(n.endswith(".project") for n in fr), plan
    want = [t.clone() for t in s.forward(x)[1:]]
    blocks = [int(n[1:n.index(".")]) for n in fr]
    wdw = {i: s.stage(f"b{i}.dw", batch) for i in blocks}
    wout = {i: s.stage(f"block{i}", batch) for i in blocks}
    cexp = {i: sd[f"backbone_net.model._blocks.{i}._depthwise_conv.conv.weight"].shape[0] for i in blocks}
    assert all(wdw[i].shape[-1] == cexp[i] for i in blocks), "a padded tensor must report its logical channel count"
    s.close()
    monkeypatch.setenv("HEP_PW_FRAG", "0")
    s = api["Session"](sd, phi, size, batch, precision, flags=flags)
    assert not any(y.startswith("pw_gemm_kernel<") and y.endswith(", true>") for _, y in _plan_syms(s, batch))
    got = s.forward(x)[1:]
    for name, a, b in zip(("regression", "classification", "rotation", "translation_raw", "hand"), got, want):
        assert torch.equal(a, b), f"{name}: max |diff| {(a - b).abs().max().item():.3e}"
    for i in blocks:
        assert torch.equal(s.stage(f"block{i}", batch), wout[i]), f"block{i}"
        assert torch.equal(s.stage(f"b{i}.dw", batch), wdw[i]), f"b{i}.dw read back through the fragment order differs from the row-major tensor"
    s.close()


def test_nan_input_reaches_every_fp32_head_as_nan(api):
    """The fp32 activations use v_rcp_f32 + one Newton step behind a clamp of the denominator (hep_dev.h: rcp_newton): the clamp
    must let a NaN through (fminf(NaN, c) is c: a NaN logit once came out of the classification sigmoid as ~1e-38 and would
    have dropped below the score threshold silently).  A poisoned frame is NaN in all five heads, its clean neighbour is untouched."""
    phi, size = 0, 256
    sd = api["sd"](phi, 3)
    x = torch.from_numpy(seeded_input((2, 3, size, size), 5))
    x[0, :, 100:140, 60:200] = float("nan")
    for prec in ("fp32", "bf16"):
        s = api["Session"](sd, phi, size, 2, prec)
        out = [t.cpu() for t in s.forward(x.cuda())[1:]]
        ref = [t.cpu() for t in s.forward(x[1:].cuda())[1:]]
        s.close()
        for name, t, r in zip(HEADS, out, ref):
            assert torch.isnan(t[0]).any(), (prec, name, "the poisoned frame came out finite")
            assert torch.isfinite(t[1]).all() and torch.equal(t[1], r[0]), (prec, name)
        assert torch.isnan(out[1][0]).float().mean().item() > 0.5, (prec, "classification scores of the poisoned frame")


@pytest.mark.parametrize("phi", [1, 2])
def test_fp32_chains_with_pointwise_weights_from_global_memory(api, phi, monkeypatch):
    """HEP_CHAIN_STREAM=2 at widths 88 / 112 in fp32 (the planner can pick it at 88): the last n-tile of a node reads weight rows
    clamped to the node's own [C][C + pad] block (they once ran up to 8 rows past it - past the blob for the last node)."""
    monkeypatch.setenv("HEP_CHAIN_STREAM", "2")
    size, batch, seed = 256, 2, 2
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    trace = {}
    ref = api["R"].forward(sd, x, phi, trace)
    s = api["Session"](sd, phi, size, batch, "fp32", flags=api["capi"].FLAG_KEEP_INTERMEDIATES)
    syms = [y for _, y in _plan_syms(s, batch)]
    out = s.forward(x.cuda())
    torch.cuda.synchronize()
    if not any(y == "chain_kernel<false, 2>" for y in syms):
        s.close()
        pytest.skip(f"phi {phi}: the planner has no LDS-resident fp32 chain with weights from global memory at this width: {sorted(set(y for y in syms if 'chain' in y or 'sep_kernel<false, 2' in y))}")
    for name, a, b in zip(HEADS, out[1:], ref[1:]):
        assert (a.cpu() - b).abs().max().item() <= 1e-3 * max(1.0, b.abs().max().item()), name
    for k, v in trace.items():
        if k.startswith("bifpn"):
            st = s.stage(f"c{k[5:k.index('_')]}.p{k[-1]}_out", batch)
            assert (st - v.permute(0, 2, 3, 1)).abs().max().item() <= 1e-3 * max(1.0, v.abs().max().item()), k
    s.close()


@pytest.mark.parametrize("size,batch", [(128, 5), (384, 2)])
def test_other_input_sizes(api, size, batch):
    """Sizes other than the two benchmark ones move every layer onto other kernels (at 128 blocks 1-2 take the small-map
    front kernel with 16 / 24 input channels: a k-step whose tail lanes must never read the unwritten pad columns of an
    LDS row - 0 x NaN is NaN, and that is how this case once failed in bf16)."""
    phi = 0
    sd = api["sd"](phi, 2)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 9))
    ref = api["R"].forward(sd, x, phi)
    for prec in ("fp32", "bf16"):
        s = api["Session"](sd, phi, size, batch, prec)
        out = s.forward(x.cuda())
        torch.cuda.synchronize()
        s.close()
        for name, a, b in zip(("regression", "classification", "rotation", "translation_raw", "hand"), out[1:], ref[1:]):
            assert torch.isfinite(a).all(), (prec, name)
            err = (a.cpu() - b).abs().max().item()
            assert err <= (1e-3 if prec == "fp32" else 0.15 * max(1.0, b.abs().max().item())), (prec, name, err)


def test_output_views_alias_the_handles_own_buffers(api):
    """hep_output_device: the tensors a serving loop hands out without copies hold exactly what the forward returns, and a
    forward WITHOUT output pointers leaves its results there."""
    phi, size, batch = 0, 256, 3
    sd = api["sd"](phi, 1)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), 5)).cuda()
    s = api["Session"](sd, phi, size, 4, "bf16")
    views = s.output_views()
    assert [tuple(v.shape) for v in views] == [(4, s.num_anchors, k) for k in (4, 1, 3, 3, 63)]
    outs = s.forward(x, want_features=False)[1:]
    torch.cuda.synchronize()
    for v, o in zip(views, outs):
        assert torch.equal(v[:batch], o)
    for v in views:
        v.zero_()
    strides = (ctypes.c_int64 * 4)(*x.stride())
    api["capi"].check(api["capi"].lib().hep_run_device(s.handle, x.data_ptr(), strides, batch, None, None, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for v, o in zip(views, outs):
        assert torch.equal(v[:batch], o)
    with pytest.raises(api["capi"].HepError):
        api["capi"].check(api["capi"].lib().hep_output_device(s.handle, 2, ctypes.byref(ctypes.c_void_p())))      # a feature map index
    s.close()


def test_inflight_pool_matches_single_session():
    """Batches in flight on several streams give bit-identical results to one session run serially, also
    when the consumer is slow (submit() used to hand back buffers it was already overwriting)."""
    import torch
    from hmd_ego_pose_amd import InflightPool
    from hmd_ego_pose_amd.model import Session
    from hmd_ego_pose_amd.weights import seeded_state_dict
    sd = seeded_state_dict(0, 0)
    B, S = 4, 256
    pool = InflightPool(sd, 0, S, B, "bf16", depth=3)
    ref = Session(sd, 0, S, B, "bf16")
    cam = torch.tensor([[480, 480, 128, 128, 1000, 1.0]] * B, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    batches = [torch.randn(B if i % 2 == 0 else B - 1, 3, S, S, device="cuda", generator=g) for i in range(7)]
    # a SLOW reader: .cpu() of every tensor (58 MB of hand rows per batch) while the other slots are in flight;
    # a returned result must not be written again before the next submit()
    got = []
    for x in batches:
        r = pool.submit(x, cam[:x.shape[0]])
        if r is not None:
            got.append({k: v.cpu() for k, v in sorted(r.items(), key=lambda kv: -kv[1].numel())})
    got += [{k: v.cpu() for k, v in r.items()} for r in pool.drain()]
    assert len(got) == len(batches)
    for x, r in zip(batches, got):
        _, reg, cls, rot, trn, hand = ref.forward(x, want_features=False)
        boxes, trans = ref.decode(reg, trn, cam[:x.shape[0]])
        torch.cuda.synchronize()
        for k, t in (("regression", reg), ("classification", cls), ("rotation", rot), ("translation_raw", trn), ("hand", hand),
                     ("boxes", boxes), ("translation", trans)):
            assert r[k].shape == t.shape and torch.equal(r[k], t.cpu()), k
    pool.close(); ref.close()


def test_preprocess_is_bit_exact_and_feeds_the_forward(api):
    """uint8 frames -> hep_preprocess_u8_device == the oracle's preprocess_image bit for bit (whose no-resize branch is
    pinned to the imported reference, tests/golden/preprocess.npz), incl. bottom/right zero padding and the resize
    branch; its NHWC-memory NCHW view goes straight into the forward."""
    size = 256
    s = api["Session"](api["sd"](0, 0), 0, size, 4, "fp32")
    rng = np.random.Generator(np.random.PCG64(11))
    for (h, w) in ((256, 256), (200, 256), (256, 131)):
        img = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
        img[0, 0, 0] = (0, 255, 128)
        want = np.stack([api["D"].preprocess_image(f, size)[0] for f in img])
        got = s.preprocess(torch.from_numpy(img).cuda())
        assert got.shape == (3, 3, size, size) and not got.is_contiguous()
        assert np.array_equal(got.permute(0, 2, 3, 1).cpu().numpy(), want), (h, w)
    a = s.forward(got)
    b = s.forward(got.contiguous())
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)
    # frames that need a resize (8-bit bilinear, OpenCV's fixed-point convention as restated in the oracle: parity
    # unpinned against cv2 itself, bit-identical between kernel and oracle): x2 up (128 -> 256), down (480x640 -> 192x256),
    # odd sizes, tall frames
    for (h, w) in ((128, 128), (480, 640), (97, 211), (300, 200)):
        img = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        want = np.stack([api["D"].preprocess_image(f, size)[0] for f in img])
        got = s.preprocess(torch.from_numpy(img).cuda())
        assert np.array_equal(got.permute(0, 2, 3, 1).cpu().numpy(), want), (h, w)
    s.close()


def test_custom_ops_cover_preprocess_and_filter(api):
    """torch.ops.hep.{preprocess,filter} (the PyTorch custom-op face of hep_preprocess_u8_device / hep_filter_device) return
    what the Session methods return."""
    import hmd_ego_pose_amd.model  # noqa: F401  (registers the ops)
    phi, size, batch = 0, 256, 2
    s = api["Session"](api["sd"](phi, 0), phi, size, batch, "fp32")
    img = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).cuda()
    x = torch.ops.hep.preprocess(img, s.handle)
    assert x.shape == (batch, 3, size, size) and torch.equal(x, s.preprocess(img))
    _, reg, cls, rot, trn, hand = s.forward(x)
    cam = torch.from_numpy(np.stack([CAMS[0], CAMS[1]])).cuda()
    boxes, trans = torch.ops.hep.decode(reg, trn, cam, s.handle)
    got = torch.ops.hep.filter(boxes, cls, rot, trans, hand, 0.5, 0.5, 50, s.handle)
    want = s.filter(boxes, cls, rot, trans, hand, 0.5, 0.5, 50)
    for g, k in zip(got, ("boxes", "scores", "labels", "rotation", "translation", "hand", "index", "count")):
        assert torch.equal(g, want[k]), k
    s.close()


def test_host_api_is_reentrant_from_threads(api):
    """Session.Run replacement semantics: hep_run on ONE handle may be re-entered from several host threads (the
    WebRTC frame callbacks of unity-sandbox/WebRTCNetCoreSandbox/Program.cs:128 do) - calls are serialised per
    handle and every caller gets the result of its own frame; separate handles run concurrently."""
    import threading
    capi = api["capi"]
    lib = capi.lib()
    phi, size = 0, 256
    sd = api["sd"](phi, 2)
    handles = [api["Session"](sd, phi, size, 1, "fp32") for _ in range(2)]
    N = handles[0].num_anchors
    frames = [seeded_input((1, 3, size, size), 20 + i) for i in range(6)]

    def run(sess, x):
        outs = [np.empty((1, N, k), np.float32) for k in (4, 1, 3, 3, 63)]
        capi.check(lib.hep_run(sess.handle, x.ctypes.data, 1, None, *[o.ctypes.data for o in outs]))
        return outs

    want = [run(handles[0], f) for f in frames]                       # serial reference
    got = [None] * len(frames) * 2

    def worker(slot, sess, f):
        for _ in range(3):
            got[slot] = run(sess, f)

    threads = [threading.Thread(target=worker, args=(i, handles[0], frames[i])) for i in range(len(frames))]
    threads += [threading.Thread(target=worker, args=(len(frames) + i, handles[1], frames[i])) for i in range(len(frames))]
    for t in threads: t.start()
    for t in threads: t.join()
    for i in range(len(frames)):
        for a_, b_, c_ in zip(want[i], got[i], got[len(frames) + i]):
            assert np.array_equal(a_, b_) and np.array_equal(a_, c_), i
    for h in handles: h.close()


def test_host_decode_and_filter_are_serialised_with_run(api):
    """hep_run, hep_decode and hep_filter re-entered concurrently on ONE handle (the C# callbacks call all three
    from worker threads): every caller gets the result of ITS inputs.  The host variants hold the handle's mutex
    from staging to the final synchronise and stage in buffers of their own, never in the forward's outputs."""
    import threading
    capi = api["capi"]
    lib = capi.lib()
    phi, size = 0, 256
    sd = api["sd"](phi, 3)
    s = api["Session"](sd, phi, size, 1, "fp32")
    N = s.num_anchors
    frames = [seeded_input((1, 3, size, size), 40 + i) for i in range(4)]
    cams = [np.array([[480 + 7 * i, 480 + 3 * i, 128, 128, 1000, 1.0]], np.float32) for i in range(4)]

    def run(x):
        outs = [np.empty((1, N, k), np.float32) for k in (4, 1, 3, 3, 63)]
        capi.check(lib.hep_run(s.handle, x.ctypes.data, 1, None, *[o.ctypes.data for o in outs]))
        return outs

    def decode(outs, cam):
        b, t = np.empty((1, N, 4), np.float32), np.empty((1, N, 3), np.float32)
        capi.check(lib.hep_decode(s.handle, outs[0].ctypes.data, outs[3].ctypes.data, cam.ctypes.data, 1, b.ctypes.data, t.ctypes.data))
        return b, t

    def filt(outs, b, t):
        M = 50
        d = [np.empty((1, M, 4), np.float32), np.empty((1, M), np.float32), np.empty((1, M), np.int32), np.empty((1, M, 3), np.float32),
             np.empty((1, M, 3), np.float32), np.empty((1, M, 63), np.float32), np.empty((1, M), np.int32), np.empty((1,), np.int32)]
        capi.check(lib.hep_filter(s.handle, b.ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, t.ctypes.data, outs[4].ctypes.data, 1,
                                  0.5, 0.5, M, *[a.ctypes.data for a in d]))
        return d

    want = []
    for f, c in zip(frames, cams):                 # serial reference
        o = run(f); b, t = decode(o, c); want.append((o, b, t, filt(o, b, t)))
    got = [None] * len(frames)
    errs = []

    def worker(i):
        try:
            for _ in range(3):
                o = run(frames[i]); b, t = decode(o, cams[i]); got[i] = (o, b, t, filt(o, b, t))
        except Exception as e:      # pragma: no cover
            errs.append(repr(e))

    def decoder(i):                  # threads that only decode / filter somebody's serial results, interleaving with the runs
        try:
            for _ in range(6):
                b, t = decode(want[i][0], cams[i])
                assert np.array_equal(b, want[i][1]) and np.array_equal(t, want[i][2])
                d = filt(want[i][0], want[i][1], want[i][2])
                assert all(np.array_equal(u, v) for u, v in zip(d, want[i][3]))
        except Exception as e:      # pragma: no cover
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(frames))] + [threading.Thread(target=decoder, args=(i,)) for i in range(len(frames))]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errs, errs
    for i in range(len(frames)):
        for u, v in zip(want[i][0], got[i][0]):
            assert np.array_equal(u, v), i
        assert np.array_equal(want[i][1], got[i][1]) and np.array_equal(want[i][2], got[i][2])
        assert all(np.array_equal(u, v) for u, v in zip(want[i][3], got[i][3]))
    # a decode with host inputs must not clobber the handle's own last outputs (NULL inputs read them)
    o = run(frames[0])
    decode(want[1][0], cams[1])
    b2, t2 = np.empty((1, N, 4), np.float32), np.empty((1, N, 3), np.float32)
    capi.check(lib.hep_decode(s.handle, None, None, cams[0].ctypes.data, 1, b2.ctypes.data, t2.ctypes.data))
    assert np.array_equal(b2, want[0][1]) and np.array_equal(t2, want[0][2])
    s.close()


def test_create_from_pack_file_and_bad_arguments(api, tmp_path):
    """hep_create(path) (what the C# host calls with model.hepw): same numbers as the in-memory constructor; missing
    file, wrong phi for the pack and out-of-range batch in hep_debug_tensor are refused with a message."""
    from hmd_ego_pose_amd import save_pack
    capi = api["capi"]
    lib = capi.lib()
    phi, size = 0, 256
    sd = api["sd"](phi, 6)
    path = str(tmp_path / "model.hepw")
    save_pack(sd, path)
    h = ctypes.c_void_p()
    capi.check(lib.hep_create(path.encode(), phi, size, 2, capi.HEP_F32, 0, 0, ctypes.byref(h)))
    x = seeded_input((2, 3, size, size), 8)
    N = lib.hep_num_anchors(h)
    outs = [np.empty((2, N, k), np.float32) for k in (4, 1, 3, 3, 63)]
    capi.check(lib.hep_run(h, x.ctypes.data, 2, None, *[o.ctypes.data for o in outs]))
    s = api["Session"](sd, phi, size, 2, "fp32")
    ref = s.forward(torch.from_numpy(x).cuda())
    for o, r in zip(outs, ref[1:]):
        assert np.array_equal(o, r.cpu().numpy())
    buf = np.empty(16, np.float32)
    assert lib.hep_debug_tensor(h, b"stem", 3, buf.ctypes.data, buf.size) == -4 and b"max_batch" in lib.hep_last_error()
    lib.hep_destroy(h)
    s.close()
    h2 = ctypes.c_void_p()
    assert lib.hep_create(str(tmp_path / "nope.hepw").encode(), phi, size, 1, capi.HEP_F32, 0, 0, ctypes.byref(h2)) == -2
    assert lib.hep_create(path.encode(), 3, 512, 1, capi.HEP_F32, 0, 0, ctypes.byref(h2)) == -2 and b"weight pack" in lib.hep_last_error()


def test_pose_errors_match_oracle_and_compiled_reference(api):
    """hep_pose_errors (ADD / ADD-S on the GPU) against the numpy oracle (eval/common.py:682-746) and, for the
    nearest-point search, against the reference's own C code compiled into oracle/_ref (when present): ADD within 1e-9
    relative (float64 on both sides, different summation order), ADD-S within 1e-6 relative (identical float32 minima,
    the mean is taken in float64 here and in float32 pairwise by numpy)."""
    from hmd_ego_pose_amd import evaluate as E
    D = api["D"]
    rng = np.random.Generator(np.random.PCG64(17))
    for P in (1, 37, 999, 1000, 1001, 4321):
        pts = (rng.standard_normal((P, 3)) * np.array([40.0, 25.0, 60.0])).astype(np.float32)
        n = 6
        rg, rp = rng.standard_normal((n, 3)).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)
        rp[0] = rg[0]; rp[1] = 0; rg[1] = 0                      # identical rotation; zero rotation (identity branch)
        tg = (rng.standard_normal((n, 3)) * 30 + np.array([0, 0, 500.0])).astype(np.float32)
        tp = tg + rng.standard_normal((n, 3)).astype(np.float32) * np.array([1, 1, 8], np.float32)
        tp[0] = tg[0]
        add, add_s = E.pose_errors(pts, rg, tg, rp, tp)
        for i in range(n):
            Rg, Rp = D.rodrigues(rg[i]), D.rodrigues(rp[i])
            _, want = D.add_metric(pts.astype(np.float64), 100.0, Rg, tg[i].astype(np.float64), Rp, tp[i].astype(np.float64))
            assert abs(add[i] - want) <= 1e-9 * max(1.0, want), (P, i, add[i], want)
            _, want_s = D.add_s_metric(pts.astype(np.float64), 100.0, Rg, tg[i].astype(np.float64), Rp, tp[i].astype(np.float64))
            assert abs(add_s[i] - want_s) <= 1e-6 * max(1.0, want_s), (P, i, add_s[i], want_s)
            try:
                step = P // 1000 + 1
                ref = D.reference_min_distances((pts.astype(np.float64) @ Rg.T + tg[i])[::step], (pts.astype(np.float64) @ Rp.T + tp[i])[::step])
                assert abs(add_s[i] - float(np.mean(ref.astype(np.float64)))) <= 1e-7 * max(1.0, want_s)
            except FileNotFoundError:
                pass
        assert add[0] <= 1e-12 and add_s[0] <= 1e-4          # identical poses (the two transforms may differ in the last ulp)


def test_evaluator_on_a_synthetic_linemod_folder(api, tmp_path):
    """The evaluate.py replacement end to end (folder reader -> GPU preprocess -> forward -> decode -> filter ->
    post-filter -> matching -> ADD / ADD-S on the GPU): its metrics equal an independent recomputation with the oracle's
    numpy functions from the detections it reports."""
    from hmd_ego_pose_amd import HMDEgoPose, TrainModelWithLoss
    from hmd_ego_pose_amd import evaluate as E
    from tests._util import make_linemod_folder
    D = api["D"]
    make_linemod_folder(str(tmp_path / "ds"), n=5)
    ds = E.LinemodFolder(str(tmp_path / "ds"))
    m = HMDEgoPose({"iter": 0}, num_classes=1, compound_coef=0, onnx_export=True, input_sizes=[256] * 9)
    m.load_state_dict(api["sd"](0, 0), strict=True)
    model = TrainModelWithLoss(m.to("cuda").eval()).eval()
    dets = []
    res = E.evaluate(ds, model, 256, score_threshold=0.5, max_detections=10, iou_threshold=0.05, batch_size=3, detections_out=dets)
    assert len(dets) == 5 and res["num_annotations"] == 5.0
    # independent recomputation
    n_ok_add = n_ok_adds = matched = n_ok_2d = 0
    hand_mm = []
    for i, (boxes, sc, _l, rots, trans, hands) in enumerate(dets):
        assert boxes.shape[0] <= 10 and np.all(sc > 0.5) and np.all(np.diff(sc) <= 0)
        ann = ds.annotations[i]
        for d in range(boxes.shape[0]):
            if E.compute_overlap(boxes[d:d + 1], ann["bbox"][None])[0, 0] >= 0.05:
                matched += 1
                Rg, Rp = D.rodrigues(ann["rotation"].astype(np.float32)), D.rodrigues(rots[d])
                ok, _ = D.add_metric(ds.points.astype(np.float64), ds.diameter, Rg, ann["translation"].astype(np.float32).astype(np.float64), Rp, trans[d].astype(np.float64))
                ok_s, _ = D.add_s_metric(ds.points.astype(np.float64), ds.diameter, Rg, ann["translation"].astype(np.float32).astype(np.float64), Rp, trans[d].astype(np.float64))
                n_ok_add += ok; n_ok_adds += ok_s
                # 2D reprojection (eval/common.py:646-679) recomputed point by point, hand joints (eval/common.py:970-982)
                K = ds.camera[i]
                px = lambda R_, t_: np.array([[K[0, 0] * q[0] / q[2] + K[0, 2], K[1, 1] * q[1] / q[2] + K[1, 2]] for q in (ds.points.astype(np.float64) @ R_.T + t_)])
                dist = np.linalg.norm(px(Rg, ann["translation"]) - px(Rp, trans[d].astype(np.float64)), axis=1).mean()
                n_ok_2d += dist <= 5.0
                hand_mm.append(np.linalg.norm(ann["coords_3d"] - hands[d].astype(np.float64).reshape(21, 3), axis=1).mean() * 1000.0)
                break
    assert res["num_matched"] == matched
    assert res["ADD"] == n_ok_add / 5 and res["ADD-S"] == n_ok_adds / 5 and 0.0 <= res["AP"] <= 1.0
    assert res["2D_projection"] == n_ok_2d / 5
    if matched:
        assert abs(res["hand_mean"] - np.mean(hand_mm)) <= 1e-9 * max(1.0, np.mean(hand_mm)) and abs(res["hand_std"] - np.std(hand_mm)) <= 1e-9 * max(1.0, np.std(hand_mm))


def test_anchor_targets_match_oracle(api):
    """Training-side anchor-target assignment on the GPU (hep_anchor_targets_device) against oracle/train_ref.py (pinned
    bit for bit to the imported reference, tests/test_decode_oracle_cpu.py): anchor states, one-hot labels, transformation
    and hand targets bit-exact (same float64 IoU arithmetic and tie rules - incl. a box equal to an anchor and a box that
    overlaps nothing); regression targets within 1e-6 (log)."""
    import importlib.util
    import os
    from hmd_ego_pose_amd.training import anchor_targets
    from oracle import train_ref as T
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mgt", os.path.join(here, "golden", "make_golden_targets.py"))
    mgt = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgt)
    anchors, _ = api["D"].anchors_for_size(256)
    cs = mgt.cases(anchors)
    shapes = [c[1] for c in cs]; boxes = [c[2] for c in cs]; labels = [c[3] for c in cs]; tts = [c[4].astype(np.float32) for c in cs]
    coords = [c[5].astype(np.float32) for c in cs]
    want = T.anchor_targets(anchors, shapes, boxes, labels, tts, coords, 1)
    got = anchor_targets(torch.from_numpy(anchors).cuda(), boxes, labels, tts, coords, [s_[:2] for s_ in shapes], 1)
    names = ("labels", "regression", "transformation", "coords")
    for n_, g, w in zip(names, got, want):
        g = g.cpu().numpy()
        assert g.shape == w.shape, n_
        assert np.array_equal(g[..., -1], w[..., -1]), f"{n_}: anchor states differ"
        if n_ == "regression":
            assert np.allclose(g[..., :4], w[..., :4], rtol=1e-6, atol=1e-6)
        else:
            assert np.array_equal(g, w), n_
    st = got[1][..., -1].cpu().numpy()
    assert (st == 1).sum() > 0 and (st == -1).sum() > 0 and st[2].max() <= 0          # image 2 has no boxes: nothing positive


def test_losses_match_oracle_and_reference():
    """Training-side losses on the GPU (hep_losses_device = batch_iterate, hmdegopose/loss.py:54-428) against the oracle
    restatement and against the values the REAL reference returned for the same seeded cases (tests/golden/losses.npz):
    relative 2e-5 (float32 sums in another order, device cos / sin / pow / log); NaN exactly where the reference has it
    (translation loss of a batch with an image without object anchors); the launch is bit-reproducible."""
    import os
    from hmd_ego_pose_amd.training import losses
    from oracle import train_ref as T
    from tests._util import loss_cases
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))
    for name, c in loss_cases().items():
        d = {k: torch.from_numpy(v).cuda() for k, v in c.items() if k != "model_points"}
        args = (d["gt_classification"], d["classification"], d["gt_regression"], d["regression"], d["gt_transformation"], d["transformation"],
                d["gt_hand"], d["hand"], c["model_points"], 3)
        out, per = losses(*args)
        out2, per2 = losses(*args)
        got = out.cpu().numpy().astype(np.float64)
        assert np.array_equal(out.cpu().numpy(), out2.cpu().numpy(), equal_nan=True) and np.array_equal(per.cpu().numpy(), per2.cpu().numpy(), equal_nan=True)
        want_o = T.batch_losses(c["gt_classification"], c["classification"], c["gt_regression"], c["regression"], c["gt_transformation"],
                                c["transformation"], c["gt_hand"], c["hand"], c["model_points"], 3).astype(np.float64)
        for want, what in ((want_o, "oracle"), (fx[name], "reference")):
            assert np.array_equal(np.isnan(got), np.isnan(want)), (name, what, got, want)
            ok = ~np.isnan(want)
            assert np.allclose(got[ok], want[ok], rtol=2e-5, atol=1e-6), (name, what, got, want)
        print(name, got)
    # without the hand branch (gt_hand = hand = None) the other four are unchanged and the hand loss is 0
    c = loss_cases()["typical"]
    d = {k: torch.from_numpy(v).cuda() for k, v in c.items() if k != "model_points"}
    full, _ = losses(d["gt_classification"], d["classification"], d["gt_regression"], d["regression"], d["gt_transformation"], d["transformation"],
                     d["gt_hand"], d["hand"], c["model_points"], 3)
    nohand, _ = losses(d["gt_classification"], d["classification"], d["gt_regression"], d["regression"], d["gt_transformation"], d["transformation"],
                       None, None, c["model_points"], 3)
    assert torch.equal(full[:4], nohand[:4]) and float(nohand[4]) == 0.0


def test_wrapper_is_losses_branch_returns_the_reference_weighted_losses(api):
    """TrainModelWithLoss(..., is_losses=True) (reference train.py:42-70) in eval mode: forward -> format_translation ->
    batch_iterate -> loss weights, as forward values on the device.  Checked against the oracle's batch_iterate fed with the
    DEVICE's own head outputs and decoded translation (the forward itself is gated elsewhere), rtol 2e-5."""
    from hmd_ego_pose_amd import HMDEgoPose, TrainModelWithLoss
    from oracle import train_ref as T
    D = api["D"]
    phi, size, B = 0, 256, 2
    sd = api["sd"](phi, 4)
    m = HMDEgoPose({"iter": 0}, num_classes=1, compound_coef=phi, onnx_export=True, input_sizes=[size] * 9)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(seeded_input((B, 3, size, size), 31)).cuda()
    cam = torch.from_numpy(np.stack([CAMS[0]] * B))
    anchors, _t_anchors = D.anchors_for_size(size)
    rng = np.random.Generator(np.random.PCG64(8))
    boxes = [np.array([[40., 50., 150., 190.]]), np.array([[10., 20., 90., 80.], [120., 100., 230., 240.]])]
    labels = [np.zeros((len(b),), np.int32) for b in boxes]
    tr = [np.concatenate([rng.uniform(-1, 1, (len(b), 3)), rng.standard_normal((len(b), 3)) * 100 + [0, 0, 600], np.zeros((len(b), 2))], 1).astype(np.float32) for b in boxes]
    co = [rng.standard_normal((len(b), 63)).astype(np.float32) * 50 for b in boxes]
    lab, reg_t, tra_t, crd_t = T.anchor_targets(anchors, [(size, size)] * B, boxes, labels, tr, co, 1)
    pts = (rng.standard_normal((1, 300, 3)) * 30).astype(np.float32)
    w = TrainModelWithLoss(m).eval()
    got = w(x, cam, is_losses=True, model_3d_points=pts, classification_gt=torch.from_numpy(lab), regression_gt=torch.from_numpy(reg_t),
            transformation_gt=torch.from_numpy(tra_t), coords_3d_gt=torch.from_numpy(crd_t), params={"img_size": (size, size), "num_rotation_parameters": 3})
    assert len(got) == 6 and all(t.is_cuda and t.dim() == 0 for t in got)
    _, reg, cls, rot, trn, hand = m(x)
    s = m.session(size, B, x.device)
    _b, t_dec = s.decode(reg, trn, cam.cuda())
    n = lambda t: t.cpu().numpy()
    want = T.batch_losses(lab, n(cls), reg_t, n(reg), tra_t, np.concatenate([n(rot), n(t_dec)], 2), crd_t, n(hand), pts, 3).astype(np.float64)
    want6 = [want[0], want[1], want[2] * 100, want[3] * 0.1, want[4]]
    want6.append(sum(want6))
    assert np.allclose([float(t) for t in got], want6, rtol=2e-5, atol=1e-6), ([float(t) for t in got], want6)
    with pytest.raises(ValueError, match="is_losses=True needs"):
        w(x, cam, is_losses=True)


def test_losses_many_object_anchors_and_large_models():
    """hep_losses_device past its internal batch sizes: 2 500 object anchors in one image (the compacted list holds 2 048 at
    a time) and 1 500 model points per class (a lane rotates more than one point; the limit is 2 048), against the oracle."""
    from hmd_ego_pose_amd.training import losses
    from oracle import train_ref as T
    rng = np.random.Generator(np.random.PCG64(99))
    B, N, K, P = 2, 6000, 1, 1500
    state = np.zeros((B, N), np.float32)
    state[0, rng.choice(N, size=2500, replace=False)] = 1.0
    state[1, rng.choice(N, size=3, replace=False)] = 1.0
    labels = (state == 1).astype(np.float32)[..., None]
    gt_c = np.concatenate([labels, state[..., None]], 2)
    gt_r = np.concatenate([rng.standard_normal((B, N, 4)).astype(np.float32) * 0.3, state[..., None]], 2)
    rot = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32); tr = (rng.standard_normal((B, N, 3)) * 50).astype(np.float32)
    sym = np.zeros((B, N, 1), np.float32); sym[1] = 1.0                      # the 3 symmetric ones: 1500 x 1500 distances each
    gt_t = np.concatenate([rot, tr, sym, np.zeros((B, N, 1), np.float32), state[..., None]], 2)
    cls = (1 / (1 + np.exp(-rng.standard_normal((B, N, K))))).astype(np.float32)
    reg = (gt_r[..., :4] + rng.standard_normal((B, N, 4)) * 0.1).astype(np.float32)
    tra = np.concatenate([rot + rng.standard_normal((B, N, 3)).astype(np.float32) * 0.05, tr + rng.standard_normal((B, N, 3)).astype(np.float32) * 2], 2).astype(np.float32)
    pts = (rng.standard_normal((1, P, 3)) * 30).astype(np.float32)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out, per = losses(c(gt_c), c(cls), c(gt_r), c(reg), c(gt_t), c(tra), None, None, pts, 3)
    want = T.batch_losses(gt_c, cls, gt_r, reg, gt_t, tra, np.zeros((B, N, 2), np.float32), np.zeros((B, N, 1), np.float32), pts, 3)
    got = out.cpu().numpy()
    assert np.allclose(got[:4], want[:4], rtol=5e-5, atol=1e-6), (got, want)
    from hmd_ego_pose_amd import _capi
    with pytest.raises(_capi.HepError):
        losses(c(gt_c), c(cls), c(gt_r), c(reg), c(gt_t), c(tra), None, None, np.zeros((1, 2049, 3), np.float32), 3)


def test_exported_onnx_model_runs_through_the_c_abi(api, tmp_path):
    """The reference's deployment artefact - model.onnx as export_to_onnx writes it (hmdegopose/misc_utils.py:36-95: eval mode,
    BatchNorm folded) - through this build's path: tools/pack_weights.py's reader maps it back to a state_dict, the HEPW pack
    goes into hep_create, and the HIP fp32 forward must match the oracle run on the ORIGINAL weights within the 1e-3 bound.
    The .onnx file is the real exporter's structure with its payloads rebuilt from the seeded weights (tests/_util.py)."""
    from hmd_ego_pose_amd.onnx_init import state_dict_from_onnx
    from tests._util import rebuilt_real_onnx_export
    blob, meta, _ = rebuilt_real_onnx_export()
    path = tmp_path / "model.onnx"; path.write_bytes(blob)
    rec = state_dict_from_onnx(str(path), 0)
    sd = api["sd"](0, 0)
    x = torch.from_numpy(seeded_input((2, 3, 256, 256), 41))
    ref = api["R"].forward(sd, x, 0)
    s = api["Session"](rec, 0, 256, 2, "fp32")
    out = s.forward(x.cuda())
    torch.cuda.synchronize()
    for name, a, b in zip(("regression", "classification", "rotation", "translation_raw", "hand"), out[1:], ref[1:]):
        assert (a.cpu() - b).abs().max().item() <= 1e-3, name
    s.close()


def test_reference_checkpoint_known_answers(api):
    """AUTO-ENABLING known-answer test for the authors' trained phi-0 checkpoint (absent from the reference checkout:
    .MISSING_LARGE_BLOBS lists pytorch-sandbox/onnx-models/model.onnx and the .pth).  Supply it with
        HEP_REF_WEIGHTS=/path/to/checkpoint.pth   (or the exported model.onnx: eval-mode exports with folded BatchNorm are mapped back)
    and the sample frame tests/golden/000000.png (the reference's onnx-models/000000.png, a data fixture) is pushed through
    preprocess -> forward -> decode -> filter; expected values are the ones the reference documents:
      raw heads at anchor 0   scratchpad.py:78-87   regression [4.3404813, 6.3829317, 0.5551747, -15.24141], classification
                              0.0143396, rotation [-0.05352388, 0.51271254, -0.23526134], translation_raw [1.6507937, 0.5715018, 0.4628573]
      final prediction        README.md:298-308, OpenCVDNNSandboxNetCore/Program.cs:604-620: score 0.99625814 at candidate anchor
                              12186 (fig/000000-output.PNG), rotation [-2.9054394, 1.0276762, 0.1723399] rad, translation
                              [-0.02811211, -0.05858146, 0.48664188] m
    Skipped when no checkpoint is supplied."""
    import os
    path = os.environ.get("HEP_REF_WEIGHTS", "")
    if not path or not os.path.exists(path):
        pytest.skip("set HEP_REF_WEIGHTS to the authors' phi-0 checkpoint (.pth, or the exported .onnx)")
    import math
    from PIL import Image
    from hmd_ego_pose_amd.weights import strip_checkpoint_prefix
    if path.endswith(".onnx"):
        from hmd_ego_pose_amd.onnx_init import state_dict_from_onnx
        sd = state_dict_from_onnx(path, 0)
    else:
        ck = torch.load(path, map_location="cpu")
        for k in ("state_dict", "model", "model_state_dict"):
            if isinstance(ck, dict) and k in ck and isinstance(ck[k], dict):
                ck = ck[k]
        sd = strip_checkpoint_prefix(ck)
    img = np.asarray(Image.open(os.environ.get("HEP_REF_IMAGE", os.path.join(os.path.dirname(__file__), "golden", "000000.png"))).convert("RGB"))
    assert img.shape == (256, 256, 3)
    s = api["Session"](sd, 0, 256, 1, "fp32")
    x = s.preprocess(torch.from_numpy(img[None].copy()).cuda())
    _, reg, cls, rot, trn, hand = s.forward(x)
    cam = torch.tensor([[480.0, 480.0, 128.0, 128.0, 1000.0, 1.0]]).cuda()
    boxes, trans = s.decode(reg, trn, cam)
    det = s.filter(boxes, cls, rot, trans, hand, 0.5, 0.5, 100)
    torch.cuda.synchronize()
    a0 = lambda t: t[0, 0].cpu().numpy()
    assert np.allclose(a0(reg), [4.3404813, 6.3829317, 0.5551747, -15.24141], atol=1e-3)
    assert np.allclose(a0(cls), [0.0143396], atol=1e-4)
    assert np.allclose(a0(rot), [-0.05352388, 0.51271254, -0.23526134], atol=1e-3)
    assert np.allclose(a0(trn), [1.6507937, 0.5715018, 0.4628573], atol=1e-3)
    assert int(det["count"][0]) >= 1
    assert int(det["index"][0, 0]) == 12186 and abs(float(det["scores"][0, 0]) - 0.99625814) <= 1e-4
    assert np.allclose(det["rotation"][0, 0].cpu().numpy() * math.pi, [-2.9054394, 1.0276762, 0.1723399], atol=1e-3)
    assert np.allclose(det["translation"][0, 0].cpu().numpy() / 1000.0, [-0.02811211, -0.05858146, 0.48664188], atol=1e-4)
    s.close()


def test_bench_line_contract():
    """`python bench.py` on one GPU prints ONE JSON line with the driver's keys, BASELINE.json's metric, the two extra
    objects (`roofline`, `cpu_baseline`) and figures that are consistent with each other."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--sustain-seconds", "0.3"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["batch_per_gpu"] == 16 and d["config"]["phi"] == 0 and d["config"]["size"] == 256
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and len(rf["layers"]) == d["config"]["launches_per_step"] - 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert d["one_batch_in_flight"]["value"] < d["value"] and d["sustained"]["windows"] == 10
    assert d["add_vs_ref"]["meets_bound"] == ["fp32"] and d["fp32"]["value"] > 0 and d["comm"]["value"] > 0
    # the target in the line: how many launches reach 0.60 of the roofline, the step as a whole, and ONE number that satisfies both
    # halves of the metric (frames/s at ADD within 0.1 mm of the reference)
    assert 0 <= rf["layers_at_or_above_0p6"] <= rf["layers_total"] == len(rf["layers"]) and 0 < rf["time_weighted_frac"] < 1
    mb = d["meets_add_bound"]
    assert mb["dtype"] == "fp32" and mb["value"] == d["fp32"]["value"] and mb["add_mm"] <= 0.1 and mb["bound_mm"] == 0.1
    f32 = d["fp32"]["roofline"]
    assert f32["kernel"] and 0 < f32["time_weighted_frac"] < 1 and f32["layers_total"] >= 50
    assert "not the parity-tested weight set" in d["comm"]["what"]


@pytest.mark.parametrize("precision,env", [("fp32", {}), ("bf16", {"HEP_CHAIN_STREAM": "1"})])
def test_results_are_bit_reproducible_with_four_batches_in_flight(precision, env):
    """tools/soak.py: four sessions in flight on four streams, the same frames every step, every sampled result bit-identical to the
    first one.  Load-dependent races - a missing wait behind the LDS-DMA weight stream of chain_kernel (fp32 sessions; forced for bf16
    here), a scratch buffer shared across streams - do not show with one batch in flight."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), precision, "0", "256", "16", "600"], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "0 mismatching snapshots" in r.stdout, (r.stdout[-800:], r.stderr[-800:])


def test_rccl_initialises_and_reduces_on_this_box():
    """The one thing about RCCL a single-GPU box can show: the library torch's "nccl" backend binds to loads, creates a communicator on
    the MI355X, and runs a collective and a barrier (world size 1 - RCCL refuses two ranks on one device, so the point-to-point scatter /
    gather of hmd_ego_pose_amd/dist.py stays covered by the gloo tests and the skipped 2-GPU test below).  Runs in a fresh child process;
    prints the per-rank line bench.py logs before its first collective (backend, RCCL version, device)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, datetime, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from hmd_ego_pose_amd import dist as hd\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))\n"
        "print(hd.describe(0))\n"
        "t = torch.arange(8, dtype=torch.float32, device='cuda')\n"
        "dist.all_reduce(t); dist.broadcast(t, 0); dist.barrier(); torch.cuda.synchronize()\n"
        "assert t.tolist() == list(range(8)), t\n"
        "assert hd.max_over_ranks(1.5, torch.device('cuda', 0)) == 1.5\n"
        "a = torch.arange(1 << 16, dtype=torch.uint8, device='cuda'); b = torch.zeros_like(a)\n"          # the grouped point-to-point form of scatter_frames / gather_detections, rank 0 to itself
        "for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]): q.wait()\n"
        "torch.cuda.synchronize(); assert torch.equal(a, b)\n"
        "dist.destroy_process_group(); print('rccl ok')\n")
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()      # (a free port: two runs on one host must not collide)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout and "backend nccl" in r.stdout, (r.stdout[-600:], r.stderr[-1200:])
    print(r.stdout.strip().splitlines()[0])


def test_two_gpu_rccl_bench_line():
    """bench.py --gpus 2 on a box with two or more MI355X: the script starts its own rank processes, the weights are
    broadcast and the frames scattered / detections gathered over RCCL (backend nccl), rank 0 prints one JSON line.
    Skipped on the single-GPU boxes the build has access to (the same wiring runs there with both ranks on one device
    over gloo, tools/run_round.sh, and with world size 2 on CPU, tests/test_dist_cpu.py)."""
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 32 and line["value"] > 0
    assert line["comm"]["backend"] == "nccl" and line["comm"]["scatter_bytes_per_step"] == 16 * 256 * 256 * 3
