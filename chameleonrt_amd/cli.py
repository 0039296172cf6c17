"""Headless re-statement of the reference application's command line and frame loop
(main.cpp:19-33, 113-345) around the HIP backend:

    python -m chameleonrt_amd.cli hip <scene.obj | scene.crts | scene.gltf | scene.glb | synthetic:NAME | C1..C5> [options]

    -eye x y z  -center x y z  -up x y z  -fov deg  -spp n  -img w h
    -mat-mode default|white_diffuse   -benchmark-frames n   -validation prefix

Without a window there is nothing interactive to do, so the loop always runs like
`-benchmark-frames` (default 16): fixed camera, `camera_changed` only on frame 0, the mean render
time / FPS / Ray/s lines of main.cpp:334-345, and `chameleonrt.png` written from `RenderBackend::img`
at the end. `-validation <prefix>` writes `<prefix>crt_hip-f<frame>.png` every frame
(main.cpp:316-325). Defaults as in main.cpp:35-36,122-126: 1280x720, eye (0,0,5), center 0,
up (0,1,0), fovy 65, 1 spp; a scene's own camera is used unless camera options are given.
"""
import os
import sys

import numpy as np

from . import core, scenes
from .camera import look_at
from .render_hip import RenderHIP

USAGE = __doc__


def _pretty(x: float) -> str:
    """pretty_print_count, util/util.cpp:23-36."""
    for div, suffix in ((1e9, "G"), (1e6, "M"), (1e3, "K")):
        if x >= div:
            return f"{x / div:.6g}{suffix}"
    return f"{x:.6g}"


def main(argv=None) -> int:
    args = list(sys.argv[1:] if argv is None else argv)
    if len(args) < 2 or args[0] in ("-h", "--help"):
        print(USAGE)
        return 1
    if args[0] != "hip":
        raise SystemExit(f"backend '{args[0]}' is not provided by this package (only 'hip')")
    width, height = 1280, 720
    eye, center, up, fovy = [0.0, 0.0, 5.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 65.0
    spp, frames, validation, mat_mode, scene_file, got_camera = 1, 16, "", "default", "", False
    spp_given = False
    i = 1
    while i < len(args):
        a = args[i]
        if a in ("-eye", "-center", "-up"):
            vals = [float(x) for x in args[i + 1:i + 4]]
            {"-eye": eye, "-center": center, "-up": up}[a][:] = vals
            got_camera, i = True, i + 3
        elif a == "-fov":
            fovy, got_camera, i = float(args[i + 1]), True, i + 1
        elif a == "-spp":
            spp, spp_given, i = int(args[i + 1]), True, i + 1
        elif a == "-img":
            width, height, i = int(args[i + 1]), int(args[i + 2]), i + 2
        elif a == "-mat-mode":
            mat_mode, i = args[i + 1], i + 1
        elif a == "-benchmark-frames":
            frames, i = int(args[i + 1]), i + 1
        elif a == "-validation":
            validation, i = args[i + 1], i + 1
        elif a == "-camera":
            i += 1
        elif not a.startswith("-"):
            scene_file = a
        i += 1
    if not scene_file:
        print(USAGE)
        return 1
    if scene_file in scenes.WORKLOADS:
        scene, w0, h0, spp0 = scenes.make_workload(scene_file)
        if "-img" not in args:
            width, height = w0, h0
        if spp_given:
            scene.samples_per_pixel = spp
    elif scene_file.startswith("synthetic:"):
        scene = getattr(scenes, scene_file.split(":", 1)[1])()
        scene.samples_per_pixel = spp
    else:  # dispatch on the extension like Scene::Scene (util/scene.cpp:49-72); the loaders apply the material mode
        scene = scenes.load_scene_file(scene_file, mat_mode, spp)
        mat_mode = "default"
    if mat_mode == "white_diffuse":
        scene = scene.white_diffuse()
    print(f"Scene '{scene_file}':\n# Unique Triangles: {_pretty(scene.unique_tris())}\n"
          f"# Total Triangles: {_pretty(scene.total_tris())}\n# Materials: {len(scene.materials)}\n"
          f"# Textures: {len(scene.textures)}\n# Lights: {len(scene.lights)}")
    if not got_camera and scene.cameras:
        c = scene.cameras[0]
        eye, center, up, fovy = list(c.position), list(c.center), list(c.up), float(c.fov_y)
    cam_eye, cam_dir, cam_up = look_at(eye, center, up)

    # CRT_HIP_ELIDE=1, like the C++ plugin: occlusion rays the reference never looks at are counted, not traced (same image)
    renderer = RenderHIP(flags=core.FLAG_ELIDE_UNUSED_SHADOW_RAYS if os.environ.get("CRT_HIP_ELIDE") == "1" else 0)
    renderer.initialize(width, height)
    renderer.set_scene(scene)
    from PIL import Image as PILImage
    total_ms = total_rps = 0.0
    for f in range(frames):
        last = f + 1 == frames
        st = renderer.render(cam_eye, cam_dir, cam_up, fovy, f == 0, last or bool(validation))
        total_ms += st.render_time_ms
        total_rps += st.rays_per_second
        if validation:
            PILImage.fromarray(renderer.img.view(np.uint8).reshape(height, width, 4), "RGBA").save(
                f"{validation}crt_hip-f{f + 1}.png")
    PILImage.fromarray(renderer.img.view(np.uint8).reshape(height, width, 4), "RGBA").save("chameleonrt.png")
    print("Image saved to chameleonrt.png")
    print(f"{renderer.name()}\nBenchmarked {frames} frames\nRender Time: {total_ms / frames:g}ms/frame "
          f"({1000.0 / (total_ms / frames):g} FPS)")
    print(f"Rays per-second {total_rps / frames:g} Ray/s ({_pretty(total_rps / frames)}Ray/s)")
    renderer.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
