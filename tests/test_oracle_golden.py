"""CPU: pin the oracle (oracle/) to the golden vectors captured from the real reference
(tests/golden/make_golden.py).  Tolerances: 1e-5 abs + 1e-5 rel for the fp32 network
(the reference and the restatement run the same ATen CPU kernels; differences are
summation-order only), bit-exact for anchors."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import decode_ref as D
from oracle import efficientpose_ref as R
from hmd_ego_pose_amd.arch import param_spec
from hmd_ego_pose_amd.weights import seeded_state_dict
from tests._util import CAMS, CASES, CLASS_CASES, check_digest, golden_case, golden_meta, seeded_input, strides_for


@pytest.mark.parametrize("size", [256, 512])
def test_anchors_bit_exact(size, golden_dir):
    meta = golden_meta()
    a, t = D.anchors_for_size(size)
    assert a.dtype == np.float32 and t.dtype == np.float32
    assert hashlib.sha256(a.tobytes()).hexdigest() == meta[f"anchors_{size}_sha256"]
    assert hashlib.sha256(t.tobytes()).hexdigest() == meta[f"translation_anchors_{size}_sha256"]
    assert hashlib.sha256(t.tobytes()).hexdigest() == meta[f"fixture_translation_anchors_{size}_sha256"]
    fx = np.load(f"{golden_dir}/anchors.npz")
    sl = lambda v: np.concatenate([v[::53], v[-9:]])
    assert np.array_equal(sl(t), fx[f"translation_anchors_{size}"])
    assert np.array_equal(sl(a), fx[f"gen_anchors_{size}"])
    if size == 256:
        assert hashlib.sha256(a.tobytes()).hexdigest() == meta["fixture_anchors_256_sha256"]
        assert np.array_equal(sl(a), fx["anchors_256"])
        # known answers quoted in SURVEY.md section 8c
        assert a[0].tolist() == [-12.0, -12.0, 20.0, 20.0]
        assert a.shape == (12276, 4) and t.shape == (12276, 3)
        assert abs(float(a.astype(np.float64).sum()) - 6285312.004671574) < 1e-6
        assert float(t.astype(np.float64).sum()) == 3285504.0


@pytest.mark.parametrize("tag", list(CASES))
def test_network_forward_matches_reference(tag):
    phi, size, batch, seed, kind = CASES[tag]
    info, gold = golden_case(tag)
    sd = seeded_state_dict(phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed, kind))
    trace = {}
    feats, reg, cls, rot, trn, hand = R.forward(sd, x, phi, trace)
    named = {"regression": reg, "classification": cls, "rotation": rot, "translation_raw": trn, "hand": hand}
    for l, f in enumerate(feats):
        named[f"feat{l + 3}"] = f.permute(0, 2, 3, 1)
    for k, v in trace.items():
        if f"trace_{k}" in info:
            named[f"trace_{k}"] = v.permute(0, 2, 3, 1)
    assert len(named) >= 10 + len([k for k in info if k.startswith("trace_")])
    for k, v in named.items():
        check_digest(k, v.numpy(), info[k], gold[k], strides_for(size, k, batch), atol=1e-5, rtol=1e-5)
    # decode against the reference's own format_bboxes / format_translation
    anchors, t_anchors = D.anchors_for_size(size)
    st = strides_for(size, "boxes", batch)
    for ci, cam in enumerate(CAMS):
        boxes = D.decode_boxes(anchors, reg.numpy(), size)
        trans = D.decode_translation(t_anchors, trn.numpy(), np.repeat(cam[None], batch, 0))
        check_digest(f"boxes_cam{ci}", boxes, info[f"boxes_cam{ci}"], gold[f"boxes_cam{ci}"], st, atol=1e-4, rtol=1e-5)
        check_digest(f"translation_cam{ci}", trans, info[f"translation_cam{ci}"], gold[f"translation_cam{ci}"], st,
                     atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("tag", list(CLASS_CASES))
def test_classifier_with_several_classes_matches_reference(tag):
    """num_classes > 1 (backbone.py:14, efficientdet/model.py:385-410): the header holds 9 * num_classes channels and the
    output is [B, N, num_classes]; pinned to the real reference module built with that num_classes."""
    phi, size, batch, seed, kind, classes = CLASS_CASES[tag]
    info, gold = golden_case(tag)
    assert golden_meta()[f"keys_phi{phi}_k{classes}_sha256"] == hashlib.sha256(repr(param_spec(phi, classes)).encode()).hexdigest()
    sd = seeded_state_dict(phi, seed, num_classes=classes)
    one = seeded_state_dict(phi, seed)
    assert [k for k in sd if sd[k].shape != one[k].shape] == ["classifier.header.pointwise_conv.conv.weight", "classifier.header.pointwise_conv.conv.bias"]
    assert R.num_classes(sd) == classes and R.num_classes(one) == 1
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed, kind))
    feats, reg, cls, rot, trn, hand = R.forward(sd, x, phi)
    assert tuple(cls.shape) == (batch, reg.shape[1], classes)
    for k, v in {"regression": reg, "classification": cls, "rotation": rot, "translation_raw": trn, "hand": hand}.items():
        check_digest(k, v.numpy(), info[k], gold[k], strides_for(size, k, batch), atol=1e-5, rtol=1e-5)
    emu = R.forward_emulated(sd, x, phi, q_act=None, q_w=None)
    assert (emu[2] - cls).abs().max().item() <= 2e-5


@pytest.mark.parametrize("tag", ["phi0_s256_b2_seed0", "phi3_s512_b1_seed0"])
def test_storage_emulating_oracle_is_the_same_network(tag):
    """oracle.forward_emulated restates the network with BN folded and explicit rounding hooks (the gate for the
    bf16 / fp8 device sessions).  With identity hooks it must be the golden-pinned ``forward`` up to fp32
    summation order; with the bf16 hooks it must show the drift bf16 storage causes (2-5 % mean relative on the
    seeded weights) - neither zero (hooks not applied) nor large (a folding bug)."""
    phi, size, batch, seed, kind = CASES[tag]
    sd = seeded_state_dict(phi, seed)
    x = torch.from_numpy(seeded_input((batch, 3, size, size), seed, kind))
    tr_ref, tr_id = {}, {}
    ref = R.forward(sd, x, phi, tr_ref)
    idn = R.forward_emulated(sd, x, phi, tr_id, q_act=None, q_w=None)
    for a, b in zip(ref[1:], idn[1:]):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())
    for k in tr_id:
        assert (tr_ref[k] - tr_id[k]).abs().max().item() <= 2e-5 * max(1.0, tr_ref[k].abs().max().item()), k
    emu = R.forward_emulated(sd, x, phi)
    for name, a, b in zip(("regression", "classification", "rotation", "translation_raw", "hand"), ref[1:], emu[1:]):
        rel = (a - b).abs().mean().item() / a.abs().mean().item()
        assert 1e-3 < rel < 0.08, (name, rel)
    # the hooks round to bf16 exactly: every stored activation is representable
    # (8 significant bits, ties to even: 1 + 2^-8 is a tie -> 1.0, 1 + 2^-7 + 2^-8 is a tie -> 1 + 2^-6)
    t = R.q_bf16(torch.tensor([1.0 + 2 ** -9, 1.0 + 2 ** -8, 1.0 + 2 ** -8 + 2 ** -9, 1.0 + 2 ** -7 + 2 ** -8, -3.14159]))
    assert t.tolist() == [1.0, 1.0, 1.0078125, 1.015625, -3.140625]
