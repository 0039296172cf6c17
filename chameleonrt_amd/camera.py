"""Camera vectors as the reference app hands them to ``RenderBackend::render``.

``ArcballCamera`` (util/arcball_camera.cpp:10-23, 61-72) derives eye/dir/up from a
look-at; ``dir`` and ``up`` are unit vectors and ``up`` is re-orthogonalised against
``dir``. The headless harness needs only that result, not the mouse interaction.
"""
import numpy as np


def _n(v):
    v = np.asarray(v, dtype=np.float32)
    return (v / np.sqrt(np.dot(v, v), dtype=np.float32)).astype(np.float32)


def look_at(eye, center, up):
    """Returns (eye, dir, up) float32 triples equivalent to ArcballCamera(eye, center, up)."""
    eye = np.asarray(eye, dtype=np.float32)
    z = _n(np.asarray(center, dtype=np.float32) - eye)
    x = _n(np.cross(z, _n(up)))
    y = _n(np.cross(x, z))
    return eye, z, y


def camera_of(scene):
    """(eye, dir, up, fovy_deg) of a scene's first camera, as main.cpp:122-125,208-213 hands them to render()."""
    cam = scene.cameras[0]
    e, d, u = look_at(cam.position, cam.center, cam.up)
    return e, d, u, cam.fov_y
