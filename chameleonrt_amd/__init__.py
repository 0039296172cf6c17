"""chameleonrt_amd -- MI355X-native wavefront path-tracing backend for ChameleonRT.

Only what the render hot path needs: the HIP core behind the C-ABI of include/crt_hip.h
(csrc/), its ctypes binding (core.py), the host-side mirror of the reference's
RenderBackend interface (render_hip.py), the scene data model (scene.py) and the
synthetic benchmark scenes (scenes.py).
"""
__version__ = "0.1.0"
