"""Generate tests/golden/ref_f4.npz + tests/golden/blender_sample/ from the reference's importable Python
(SURVEY.md section 8f rank 4).  Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_f4.py

Captured (data only -- inputs written by this script, outputs produced by the reference):
  * scene/dataset_readers.py:179-220 readCamerasFromTransforms on tests/golden/blender_sample/ (a transforms json and
    three small RGBA PNGs written below): R, T, FovX, FovY, width, height, image names, the composited image bytes
  * utils/camera_utils.py:20-53 loadCam's resolution rule for a table of (image size, --resolution, scale) cases
    (Camera replaced by a recorder for the call: the reference's Camera moves tensors to "cuda")
  * render.py:45-73 id2rgb for ids 0..256 and visualize_obj on a small id map
Third-party modules the reference imports but this image lacks are replaced by empty stand-ins for the import only.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = os.path.join(HERE, "blender_sample")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def write_sample():
    from PIL import Image
    os.makedirs(os.path.join(SAMPLE, "train"), exist_ok=True)
    rng = np.random.default_rng(7)
    frames = []
    sizes = [(40, 30), (40, 30), (40, 30)]
    for i, (w, h) in enumerate(sizes):
        rgba = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        rgba[: h // 3, :, 3] = 0
        rgba[-h // 3:, :, 3] = 255
        Image.fromarray(rgba, "RGBA").save(os.path.join(SAMPLE, "train", f"r_{i}.png"))
        th, ph = 0.7 * i + 0.2, 0.3 + 0.25 * i
        eye = 4.0 * np.array([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)])
        z = eye / np.linalg.norm(eye)                      # OpenGL camera looks down -z: +z points away from the target
        x = np.cross([0.0, 0.0, 1.0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
        frames.append({"file_path": f"./train/r_{i}", "rotation": 0.0125, "transform_matrix": c2w.tolist()})
    with open(os.path.join(SAMPLE, "transforms_train.json"), "w") as f:
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f, indent=1)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    write_sample()
    _stub("plyfile", PlyData=object, PlyElement=object)
    _stub("simple_knn")
    _stub("simple_knn._C", distCUDA2=None)
    _stub("diff_gaussian_rasterization", GaussianRasterizationSettings=object, GaussianRasterizer=object)
    out = {}
    from scene.dataset_readers import readCamerasFromTransforms
    # Pillow >= 10 refuses int8 input to Image.fromarray(arr, "RGB"); the Pillow the reference was written against took
    # the buffer as raw bytes.  Same bytes, for this call only:
    from PIL import Image as _Image
    _orig = _Image.fromarray
    _Image.fromarray = lambda a, mode=None: _orig(a.view(np.uint8) if a.dtype == np.int8 else a, mode)
    for white in (False, True):
        infos = readCamerasFromTransforms(SAMPLE, "transforms_train.json", white, ".png")
        tag = "w" if white else "b"
        out[f"tf_R_{tag}"] = np.stack([c.R for c in infos])
        out[f"tf_T_{tag}"] = np.stack([c.T for c in infos])
        out[f"tf_fov_{tag}"] = np.array([[c.FovX, c.FovY] for c in infos])
        out[f"tf_size_{tag}"] = np.array([[c.width, c.height] for c in infos])
        out[f"tf_image_{tag}"] = np.stack([np.array(c.image) for c in infos])
    out["tf_names"] = np.array([c.image_name for c in infos])

    import utils.camera_utils as cu
    from PIL import Image
    cu.Camera = lambda **kw: kw                            # recorder: the real one needs a CUDA device
    cases = []
    res = []
    for (w, h) in ((800, 800), (1920, 1080), (3840, 2160), (1601, 900), (640, 481)):
        for r in (-1, 1, 2, 4, 8, 1000, 333.0):
            for sc in (1.0, 2.0, 1.5):
                info = types.SimpleNamespace(image=Image.new("RGB", (w, h)), uid=0, R=np.eye(3), T=np.zeros(3), FovX=1.0,
                                             FovY=1.0, image_name="x")
                args = types.SimpleNamespace(resolution=r, data_device="cpu")
                cam = cu.loadCam(args, 0, info, sc)
                cases.append((w, h, r, sc))
                res.append((cam["image"].shape[2], cam["image"].shape[1]))
    out["res_cases"] = np.array(cases, dtype=np.float64)
    out["res_out"] = np.array(res, dtype=np.int64)

    # scene/gaussian_model.py:377-411: the PLY property order and the arrays save_ply hands to plyfile (plyfile itself is
    # absent; a recorder stands in for it and keeps the structured array the reference built)
    import torch
    import scene.gaussian_model as gm
    captured = {}

    class _El:
        @staticmethod
        def describe(arr, name):
            captured["elements"], captured["name"] = arr.copy(), name
            return arr

    class _Pd:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = path
    gm.PlyElement, gm.PlyData = _El, _Pd
    gm.mkdir_p = lambda p: None
    m = gm.GaussianModel(3)
    gen = torch.Generator().manual_seed(11)
    P = 9
    m._xyz = torch.randn(P, 3, generator=gen)
    m._features_dc = torch.randn(P, 1, 3, generator=gen)
    m._features_rest = torch.randn(P, 15, 3, generator=gen)
    m._opacity = torch.randn(P, 1, generator=gen)
    m._scaling = torch.randn(P, 3, generator=gen)
    m._rotation = torch.randn(P, 4, generator=gen)
    m._objects_dc = torch.randn(P, 1, 16, generator=gen)
    out["ply_attribute_names"] = np.array(m.construct_list_of_attributes())
    m.save_ply("/tmp/unused/point_cloud.ply")
    out["ply_elements_bytes"] = np.frombuffer(captured["elements"].tobytes(), dtype=np.uint8)
    out["ply_elements_descr"] = np.array([f"{n}:{t}" for n, t in captured["elements"].dtype.descr])
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_objects_dc"):
        out["ply_in" + n] = getattr(m, n).numpy()

    _stub("gaussian_renderer", render=None, GaussianModel=object)
    _stub("scene", Scene=object)
    _stub("scene.gaussian_model", GaussianModel=object)   # not needed by the captured functions
    _stub("arguments", ModelParams=object, PipelineParams=object, get_combined_args=None)
    _stub("torchvision")
    _stub("cv2")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_render", os.path.join(REF, "render.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
        out["id2rgb"] = np.stack([mod.id2rgb(i) for i in range(257)])
        ids = (np.arange(12 * 9).reshape(9, 12) * 7 % 23).astype(np.uint8)
        out["vis_ids"] = ids
        out["vis_rgb"] = mod.visualize_obj(ids)
    except Exception as e:                                 # an ordinary import error of some viewer dependency
        print("render.py not importable here:", repr(e))
    np.savez_compressed(os.path.join(HERE, "ref_f4.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_f4.npz"), sorted(out))


if __name__ == "__main__":
    main()
