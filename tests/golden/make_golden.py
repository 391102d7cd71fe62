#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, read-only).  It imports the
reference's Python (never copies it: sys.dont_write_bytecode, stubs for the absent
torchvision / tensorflow / Cython modules as SURVEY.md Appendix D), feeds it the seeded
weights of ``hmd_ego_pose_amd.weights.seeded_state_dict`` and seeded inputs, and stores
small slices + float64 sums of what the reference returns.  The fixtures are data; this
script is the committed recipe that made them.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/pytorch-sandbox"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("torchvision"); _stub("torchvision.ops"); _stub("torchvision.ops.boxes", nms=None)
    tf = _stub("tensorflow"); tf.keras = _stub("tensorflow.keras")
    _stub("generators.utils.compute_overlap", compute_overlap=None, wrapper_c_min_distances=None)
    sys.path.insert(0, REF)
    from backbone import HMDEgoPose                      # noqa: E402
    from hmdegopose.loss import create_anchors, format_bboxes, format_translation   # noqa: E402
    return HMDEgoPose, create_anchors, format_bboxes, format_translation


def seeded_input(shape, seed, kind="normal"):
    rng = np.random.Generator(np.random.PCG64([seed, 0x1234]))
    a = rng.standard_normal(shape) if kind == "normal" else rng.random(shape)
    return a.astype(np.float32)


def digest(t, stride):
    """float64 sum / abs-sum + a fixed strided slice of the flattened tensor."""
    a = np.ascontiguousarray(t, dtype=np.float32).reshape(-1)
    return dict(sum=float(a.astype(np.float64).sum()), abssum=float(np.abs(a.astype(np.float64)).sum()),
                shape=list(np.shape(t))), a[::stride].copy()


def main():
    import torch
    from hmd_ego_pose_amd.arch import param_spec
    from hmd_ego_pose_amd.weights import seeded_state_dict
    torch.manual_seed(0)
    HMDEgoPose, create_anchors, format_bboxes, format_translation = import_reference()
    meta = {"torch": torch.__version__, "numpy": np.__version__}

    # ---- 1. anchors: the reference's own fixtures + its generator -------------------------
    fx = {}
    for name in ("anchors_256", "translation_anchors_256", "translation_anchors_512"):
        cols = 4 if name.startswith("anchors") else 3
        arr = np.loadtxt(os.path.join(REF, "onnx-models", name + ".txt"), dtype=np.float64).astype(np.float32).reshape(-1, cols)
        fx[name] = arr
    for size in (256, 512):
        a, t = create_anchors(size)
        assert a.dtype == np.float32 and t.dtype == np.float32
        if size == 256:
            assert np.array_equal(a, fx["anchors_256"]), "reference generator != its own fixture"
        assert np.array_equal(t, fx[f"translation_anchors_{size}"])
        fx[f"gen_anchors_{size}"] = a
        meta[f"fixture_translation_anchors_{size}_sha256"] = hashlib.sha256(fx[f"translation_anchors_{size}"].tobytes()).hexdigest()
        meta[f"anchors_{size}_sha256"] = hashlib.sha256(a.tobytes()).hexdigest()
        meta[f"translation_anchors_{size}_sha256"] = hashlib.sha256(t.tobytes()).hexdigest()
        meta[f"anchors_{size}_sum"] = float(a.astype(np.float64).sum())
        meta[f"translation_anchors_{size}_sum"] = float(t.astype(np.float64).sum())
    meta["camera_params"] = [float(v) for v in open(os.path.join(REF, "onnx-models", "camera_params.txt")).read().split()]
    meta["fixture_anchors_256_sha256"] = hashlib.sha256(fx["anchors_256"].tobytes()).hexdigest()
    # full arrays are ~1 MB: commit the hashes/sums above plus every 53rd row and both ends
    np.savez_compressed(os.path.join(HERE, "anchors.npz"),
                        **{k: np.concatenate([v[::53], v[-9:]]) for k, v in fx.items()})

    # ---- 2. network forward: reference module on seeded weights ---------------------------
    cams = np.array([[480, 480, 128, 128, 1000, 1.0],
                     [572.4114, 573.57043, 325.2611, 242.04899, 1000, 0.8]], dtype=np.float32)
    from tests._util import CASES, CLASS_CASES, strides_for      # one table of cases / slice strides for this script and the tests
    for tag, (phi, size, batch, seed, kind, classes) in [(t, c + (1,)) for t, c in CASES.items()] + list(CLASS_CASES.items()):
        assert tag == f"phi{phi}_s{size}_b{batch}_seed{seed}" + (f"_k{classes}" if classes != 1 else "")
        model = HMDEgoPose({"iter": 0}, num_classes=classes, compound_coef=phi, onnx_export=True, input_sizes=[size] * 9).eval()
        ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        assert ref_keys == param_spec(phi, classes), "arch.param_spec drifted from the reference state_dict"
        meta[f"keys_phi{phi}" + (f"_k{classes}" if classes != 1 else "") + "_sha256"] = hashlib.sha256(repr(ref_keys).encode()).hexdigest()
        sd = seeded_state_dict(phi, seed, num_classes=classes)
        model.load_state_dict(sd, strict=True)
        x = torch.from_numpy(seeded_input((batch, 3, size, size), seed, kind))
        # hooks: stage boundaries to localise bugs (stem, every MBConv, every BiFPN cell)
        trace = {}
        bb = model.backbone_net.model
        hooks = [bb._swish.register_forward_hook(lambda m, i, o: trace.setdefault("stem", o))]
        for i, blk in enumerate(bb._blocks):
            hooks.append(blk.register_forward_hook(lambda m, i_, o, i=i: trace.__setitem__(f"block{i}", o)))
        for r, cell in enumerate(model.bifpn):
            def cell_hook(m, i_, o, r=r):
                for l, t in enumerate(o):
                    trace[f"bifpn{r}_p{l + 3}"] = t
            hooks.append(cell.register_forward_hook(cell_hook))
        with torch.no_grad():
            feats, reg, cls, rot, trn, hand = model(x)
            # eval/common.py:397 feeds an NHWC-memory *view*; ATen then picks channels-last
            # kernels whose summation order differs: record how far that moves the outputs
            xv = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            outs_v = model(xv)[1:]
            meta[f"{tag}_nhwc_view_maxdiff"] = [float((a - b).abs().max()) for a, b in zip((reg, cls, rot, trn, hand), outs_v)]
        for h in hooks:
            h.remove()
        out, info = {}, {}
        named = {"regression": reg, "classification": cls, "rotation": rot, "translation_raw": trn, "hand": hand}
        if classes == 1:     # (a class case shares every weight but the classifier header with its one-class twin: heads only)
            for l, f in enumerate(feats):
                named[f"feat{l + 3}"] = f.permute(0, 2, 3, 1)          # stored NHWC
            for k, v in trace.items():
                named["trace_" + k] = v.permute(0, 2, 3, 1)
        for k, v in named.items():
            info[k], out[k] = digest(v.numpy(), strides_for(size, k, batch))
        # decode through the reference's own format_bboxes / format_translation
        anchors, t_anchors = create_anchors(size)
        for ci, cam in enumerate(cams):
            camb = torch.from_numpy(np.repeat(cam[None], batch, 0))
            boxes = format_bboxes(x, anchors, reg).numpy()
            trans = format_translation(t_anchors, trn, camb).numpy()
            info[f"boxes_cam{ci}"], out[f"boxes_cam{ci}"] = digest(boxes, strides_for(size, "boxes", batch))
            info[f"translation_cam{ci}"], out[f"translation_cam{ci}"] = digest(trans, strides_for(size, "translation", batch))
        np.savez_compressed(os.path.join(HERE, f"net_{tag}.npz"), **out)
        meta[tag] = info
        print(tag, "done:", {k: round(v["abssum"], 3) for k, v in list(info.items())[:5]})
    meta["cams"] = cams.tolist()
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
