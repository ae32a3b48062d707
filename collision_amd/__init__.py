"""collision_amd -- MI355X-native broad-phase sphere collision engine.

Drop-in for the hot path of kwohlfahrt/collision: the same module and class names
(``collision_amd.collision.Collider``, ``.radix.RadixSorter``, ``.scan.PrefixScanner``,
``.reduce.Reducer``, ``.bounds.Bounds``, ``.misc``), but the device side is
``libcollision_hip.so`` -- hand-written HIP for gfx950 behind a C ABI
(``include/collision_hip.h``) bound with ctypes -- instead of PyOpenCL + OpenCL C.

``collision_amd.hip`` holds the small runtime that replaces the PyOpenCL objects the
reference's callers create (Context, CommandQueue, Buffer, Event, enqueue_copy, ...).
There is no CPU fallback: if the shared library is missing, importing the runtime raises.
"""
__version__ = "0.1.0"
