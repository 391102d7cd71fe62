import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["HEP_PLAN_DEBUG"] = "1"
import torch
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
s = Session(seeded_state_dict(3, 0), 3, 512, 8, "bf16")
