#!/usr/bin/env python3
"""Convert a reference checkpoint (.pth state_dict, keys un-prefixed, `model.` or `model.module.`
prefixed - pytorch-sandbox/evaluate.py:102-116) or an exported .onnx - a training-mode export whose initialisers carry the state_dict
names, or the eval-mode export of the reference's export_to_onnx with BatchNorm folded (hmd_ego_pose_amd/onnx_init.py) into the HEPW weight pack that libhep.so reads.

    python tools/pack_weights.py weights/syn_colibri/fold_0/phi_0_....pth model.hepw --phi 0
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("out")
    ap.add_argument("--phi", type=int, default=0)
    a = ap.parse_args()
    import torch
    from hmd_ego_pose_amd import param_spec, save_pack, strip_checkpoint_prefix
    if a.checkpoint.lower().endswith(".onnx"):
        from hmd_ego_pose_amd.onnx_init import state_dict_from_onnx
        state = state_dict_from_onnx(a.checkpoint, a.phi)
    else:
        state = strip_checkpoint_prefix(torch.load(a.checkpoint, map_location="cpu", weights_only=True))
    want = dict(param_spec(a.phi))
    missing = [k for k in want if k not in state and not k.endswith("num_batches_tracked")]
    wrong = [k for k in want if k in state and tuple(state[k].shape) != want[k]]
    if missing or wrong:
        raise SystemExit(f"checkpoint does not match phi={a.phi}: {len(missing)} missing (e.g. {missing[:2]}), {len(wrong)} wrong shape (e.g. {wrong[:2]})")
    save_pack({k: state[k] for k in want if k in state}, a.out)
    print(f"wrote {a.out}: {os.path.getsize(a.out)} bytes, {len(want)} tensors")


if __name__ == "__main__":
    main()
