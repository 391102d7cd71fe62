"""hmd_ego_pose_amd - MI355X-native (gfx950) EfficientPose / HMD-EgoPose inference path.

Only what the hot path needs: ``csrc/`` (HIP kernels + the C ABI of ``libhep.so``),
the ctypes binding, and the host-side mirror of the reference's operator interface
(``HMDEgoPose``, ``TrainModelWithLoss``).  See DESIGN.md.
"""
from .arch import get_arch, level_sizes, num_anchors_total, param_spec  # noqa: F401
from .weights import load_pack, pack_bytes, save_pack, seeded_state_dict, strip_checkpoint_prefix  # noqa: F401


def __getattr__(name):   # torch custom-op registration happens on first use of the model API
    if name in ("HMDEgoPose", "TrainModelWithLoss", "Session"):
        from . import model
        return getattr(model, name)
    if name == "InflightPool":
        from .pipeline import InflightPool
        return InflightPool
    raise AttributeError(name)
