"""hmme -- MI355X-native integer block-matching motion estimation for HM 16.4.

Python side of the engine: ctypes bindings onto the C ABI (include/hmme.h), the frame-shard
driver used by bench.py, and synthetic-frame helpers.  The compute path is the HIP library
hm-opencl_amd/csrc/libhmme.so; nothing here falls back to a CPU implementation.
"""
