"""CPU oracle for the EfficientPose (HMD-EgoPose) forward + decode path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``hmd_ego_pose_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg use it, and there only as the checker /
the timed CPU baseline.  See oracle/README.md for how it is pinned.
"""
