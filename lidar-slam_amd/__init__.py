"""lidar-slam_amd: MI355X-native scan-matching front end (VoxelFilter -> NDT / ICP scan-to-local-map).

The product is the C-ABI HIP library under csrc/ (declared in include/lidar_slam_amd.h); this Python
package is the thin host-side mirror used by tests/bench: ctypes bindings (capi), the mirrors of the
reference's RegistrationInterface / CloudFilterInterface (registration, cloud_filter), the multi-GPU
sharding glue (dist) and the synthetic scan generator (synth).
"""
__version__ = "0.1.0"
