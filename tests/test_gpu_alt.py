"""GPU (MI355X), ALTERNATIVE library only: parity of the plan alternatives that were measured and rejected (NOTEBOOK.md) - the
image-resident late-block kernel (k_late.hip), the depth-first head kernel (k_heads.hip), the fused stem (k_sbf.hip), the
squeeze-excite finish in the fronts' tail, and the A/B knobs of hep_knobs.h's second half.  None of this is in libhep.so.

Every test here skips unless the loaded library reports "alt" in hep_build_info(); the default `pytest -m gpu` reaches them through
tests/test_gpu_parity.py::test_alternative_plan_suite_runs_against_the_opt_in_library, which runs this file in a child process with
HEP_LIB=hmd_ego_pose_amd/libhep_alt.so (`make -C hmd_ego_pose_amd/csrc alt`; __graft_entry__.build() makes it).
"""
import numpy as np
import pytest
import torch

from tests._util import CLASS_CASES, seeded_input
from tests.test_gpu_parity import HEADS, _plan_syms, _teacher_forced_bf16, api  # noqa: F401  (api: the module-scoped fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_alternative_build(api):
    if "alt" not in api["capi"].lib().hep_build_info().decode().split():
        pytest.skip("libhep.so does not carry the rejected alternatives: run with HEP_LIB=hmd_ego_pose_amd/libhep_alt.so")


# the environment that selects an alternative and a predicate over the session's launch list (name, device function) that is true
# ONLY when the alternative really was planned
ALT_PLANS = [
    ({"HEP_MBF": "all"}, lambda ks: any(n == "b3.front" for n, _ in ks) and any(n == "b0.front" for n, _ in ks)),
    ({"HEP_MBF": "none", "HEP_DWLDS": "0"}, lambda ks: not any("mbf_kernel" in y for _, y in ks)),
    ({"HEP_DWLDS": "1", "HEP_MBF": "none"}, lambda ks: all(n.endswith(".dw") for n, y in ks if "mbf_kernel" in y) and any("mbf_kernel" in y for _, y in ks)),
    ({"HEP_MBF_MP": "force", "HEP_MBF_MP_RES": "1"}, lambda ks: sum(y.endswith(", false, 2>") for _, y in ks if "mbf_kernel" in y) >= 8),      # multi-pass fronts with the whole tile requested at kernel start and held in registers
    ({"HEP_CHAIN": "0"}, lambda ks: not any("chain_kernel" in y or "sep_kernel<false, 2" in y for _, y in ks)),
    ({"HEP_CHAIN": "1"}, lambda ks: any("sep_kernel<false, 2" in y for _, y in ks) and not any("chain_kernel" in y for _, y in ks)),
    ({"HEP_SE_TAIL": "1", "HEP_SE_MAXMB": "0"}, lambda ks: sum(n.endswith(".front+se") for n, _ in ks) >= 11 and sum("se_finish_kernel" in y for _, y in ks) <= 1),
    ({"HEP_PWG": "0", "HEP_PW_MT2": "0"}, lambda ks: not any("pw_group_kernel" in y for _, y in ks)),
    ({"HEP_TOWER": "0"}, lambda ks: not any("tower_" in y for _, y in ks)),
    ({"HEP_XBF": "0"}, lambda ks: not any("xbf_kernel" in y for _, y in ks)),
    ({"HEP_XBF_MINH": "32"}, "bf16:xbf>=4"),          # (fp32 tiles of the 32x32 boundary do not fit LDS: the plan change is checked on a bf16 session)
    ({"HEP_SBF": "1"}, lambda ks: any("sbf_kernel" in y for _, y in ks)),
    ({"HEP_SEP_TS4_MAXHW": "16"}, None),               # (same kernels on 4x4 tiles: the launch list does not change)
]


@pytest.mark.parametrize("env,planned", ALT_PLANS, ids=["-".join(f"{k}={v}" for k, v in e.items()) for e, _ in ALT_PLANS])
def test_rejected_alternatives_keep_parity(api, env, planned, monkeypatch):
    """Every rejected alternative must produce the same numbers as the oracle (fp32, 1e-3) - and must really be the plan that ran."""
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
    for name, a, b in zip(HEADS, out[1:], ref[1:]):
        err = (a.cpu() - b).abs().max().item()
        assert err <= 1e-3, f"{env} {name}: {err:.3e}"
    for a, b in zip(out[0], ref[0]):
        assert (a.cpu() - b).abs().max().item() <= 1e-3
    s.close()


@pytest.mark.parametrize("phi,env", [(3, {"HEP_CHAIN_WGLOBAL": "0"}), (3, {"HEP_SEP_TS4_MAXHW": "32"}), (0, {"HEP_SEP_TS4_MAXHW": "16"})])
def test_bf16_alternatives_that_only_move_data_are_bit_identical(api, phi, env, monkeypatch):
    """Width-160 chains on k_sep.hip instead of chain_kernel's global-weights form; BiFPN nodes of the small levels on 4x4 tiles."""
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
    for name, a, b in zip(HEADS, got, want):
        assert torch.equal(a, b), f"{env} {name}: max |diff| {(a - b).abs().max().item():.3e}"
    s.close()


def test_depth_first_head_kernel_with_several_classes(api, monkeypatch):
    """The class count reaches the depth-first head kernel's classifier header as well (HEP_HEADS_FUSED=1)."""
    phi, size, batch, seed, _kind, classes = CLASS_CASES[next(iter(CLASS_CASES))]
    sd = api["sd"](phi, seed, num_classes=classes)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed)).cuda()
    sb = api["Session"](sd, phi, size, batch, "bf16")
    cb = sb.forward(x)[2].float().cpu()
    sb.close()
    monkeypatch.setenv("HEP_HEADS_FUSED", "1")
    sf = api["Session"](sd, phi, size, batch, "bf16")
    assert any(y == "heads_kernel" for _, y in _plan_syms(sf, batch))
    assert torch.equal(sf.forward(x)[2].float().cpu(), cb)
    sf.close()


@pytest.mark.parametrize("size,batch", [(256, 16), (384, 3)])
def test_depth_first_head_kernel_is_bit_identical(api, size, batch, monkeypatch):
    """HEP_HEADS_FUSED=1 (not the default: 71 us against 70 us stand-alone, -1.8 % frames/s with four batches in flight - DESIGN.md
    section 2): the tower layers and headers of all five nets on all five levels as ONE launch (k_heads.hip: a 16x16 output tile per
    workgroup, the layers in place in LDS, only pixels inside the image computed).  Same arithmetic in the same order as k_tower.hip:
    the five head outputs must agree bit for bit - on the benchmark shape, on ragged levels (384: 48, 24, 12, 6, 3) and on levels
    smaller than the tile's halo (128: 16, 8, 4, 2, 1)."""
    phi, seed = 0, 6
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed)).cuda()
    s0 = api["Session"](sd, phi, size, batch, "bf16")
    want = [t.clone() for t in s0.forward(x)[1:]]
    n0 = len(s0.kernels(batch))
    s0.close()
    monkeypatch.setenv("HEP_HEADS_FUSED", "1")
    s = api["Session"](sd, phi, size, batch, "bf16")
    plan = _plan_syms(s, batch)
    assert [n for n, y in plan if y == "heads_kernel"] == ["heads.fused"] and len(plan) == n0 - 3 and not any("tower" in y for _, y in plan), plan
    for _ in range(2):
        got = s.forward(x)[1:]
        torch.cuda.synchronize()
        for name, a, b in zip(HEADS, got, want):
            assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} elements differ, max {float((a - b).abs().max()):.3e}"
    s.close()
    sf = api["Session"](sd, phi, size, batch, "fp32")      # fp32 sessions keep the launch-by-launch towers
    assert not any(y == "heads_kernel" for _, y in _plan_syms(sf, batch))
    sf.close()


@pytest.mark.parametrize("batch,group", [(16, 1), (3, 3)])
def test_late_block_kernel_alternative_plan(api, batch, group, monkeypatch):
    """HEP_LATE=1 (not the default: measured, +1.7 % frames/s with four batches in flight, -4.5 % with one - NOTEBOOK.md section 2):
    blocks 12-15 of phi 0 @ 256 as ONE image-resident launch (k_late.hip): one workgroup per image (HEP_LATE_G=1), or a group of three
    that split the expanded channels and meet once per block at a counter in global memory (the default of the alternative; batch 3:
    the group placement for batches that are no multiple of eight).  The rounding points are those of the launch-by-launch plan,
    fp32 summation orders differ, so the gate is the teacher-forced one (every block on the device's own input against the
    bf16-emulating oracle) - and block 12, the first fused block, must sit within a few flipped bf16 roundings of the
    launch-by-launch plan's.  Two forwards must agree bit for bit (the group's sums meet in a fixed order)."""
    monkeypatch.setenv("HEP_LATE_G", str(group))
    phi, size, seed = 0, 256, 0
    sd = api["sd"](phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed))
    s0 = api["Session"](sd, phi, size, batch, "bf16", flags=api["capi"].FLAG_KEEP_INTERMEDIATES)
    s0.forward(x.cuda())
    want12 = s0.stage("block12", batch).float().cpu()
    n0 = len(s0.kernels(batch))
    replaced = sum(n.split(".")[0] in ("b12", "b13", "b14", "b15") for n, *_ in s0.kernels(batch))      # fronts (+ their squeeze-excite launches) + projects
    s0.close()
    monkeypatch.setenv("HEP_LATE", "1")
    s = api["Session"](sd, phi, size, batch, "bf16", flags=api["capi"].FLAG_KEEP_INTERMEDIATES)
    plan = _plan_syms(s, batch)
    assert [n for n, y in plan if y == "late_kernel"] == ["b12-b15.blocks"] and len(plan) == n0 - replaced + 1 and replaced in (8, 12), plan
    first = [t.clone() for t in s.forward(x.cuda())[1:]]
    got12 = s.stage("block12", batch).float().cpu()
    for _ in range(3):
        again = s.forward(x.cuda())[1:]
        assert all(torch.equal(a, b) for a, b in zip(first, again)), ("two forwards of the grouped launch differ", [(int((a != b).sum()), float((a.float() - b.float()).abs().max()), bool(torch.isfinite(b).all())) for a, b in zip(first, again)])
    s.close()
    d = (got12 - want12).abs()
    assert d.mean().item() <= 1e-5 * want12.abs().mean().item() and d.max().item() <= 2 ** -6 * want12.abs().max().item(), (d.mean().item(), d.max().item())
    _teacher_forced_bf16(api, sd, phi, size, batch, x, api["R"].forward(sd, x, phi))
    # fp32 sessions do not take it
    sf = api["Session"](sd, phi, size, batch, "fp32")
    assert not any(y == "late_kernel" for _, y in _plan_syms(sf, batch))
    sf.close()
    if group > 1 and batch == 3:
        # the hand-off between a group's members (write-through stores, cache-bypassing loads, a counter) must not depend on where
        # they run: HEP_LATE_XCD=1 puts them on consecutive workgroup ids = three different XCDs - same bits
        monkeypatch.setenv("HEP_LATE_XCD", "1")
        sx = api["Session"](sd, phi, size, batch, "bf16")
        for _ in range(3):
            outs = sx.forward(x.cuda())[1:]
            assert all(torch.equal(a, b) for a, b in zip(first, outs)), "a group spread over XCDs computes other bits"
        sx.close()
