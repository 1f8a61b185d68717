"""Generate tests/golden/ref_twins.npz from the reference's importable Python twins.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is captured (data only -- inputs and the reference's outputs):
  * utils/sh_utils.py: eval_sh for deg 0..3, RGB2SH, SH2RGB  (+ the "+0.5, clamp_min 0" of
    gaussian_renderer/__init__.py:78)
  * utils/graphics_utils.py: getWorld2View2, getProjectionMatrix; full_proj / camera centre formed as
    scene/cameras.py:54-57 does, on CPU tensors
  * utils/general_utils.py: inverse_sigmoid
  * scene/gaussian_model.py: the activation getters (:97-124) of a CPU GaussianModel
  * attack.py:25-173: the ten PGD step functions applied to a small fake model with .grad set
  * scene/colmap_loader.py text readers + qvec2rotmat and the FoV / pose conversion of
    scene/dataset_readers.py:68-143 on tests/golden/colmap_sample/ (a hand-written COLMAP text model; data, not code)
  * scene/colmap_loader.py binary readers on tests/golden/colmap_sample_bin/ (written by this script in COLMAP's
    published binary layout, with 2D points and tracks present)

Third-party modules the reference imports but this image lacks (hydra, omegaconf, plyfile,
simple_knn, diff_gaussian_rasterization, detectors.factory's dependencies) are replaced by empty
stand-in modules *for the import only*; none of their functionality is used by the captured functions.
Functions that hard-code device="cuda" (utils/general_utils.py:64-110, reached through
GaussianModel.get_covariance) run with torch.zeros redirected to the CPU for the duration of the call.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_twins.npz")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    # import-only stand-ins
    _stub("hydra", main=lambda **kw: (lambda f: f))
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    _stub("plyfile", PlyData=object, PlyElement=object)
    _stub("simple_knn")
    _stub("simple_knn._C", distCUDA2=None)
    _stub("diff_gaussian_rasterization", GaussianRasterizationSettings=object, GaussianRasterizer=object)
    _stub("detectors")
    _stub("detectors.factory", load_detector=None)

    from utils.sh_utils import eval_sh, RGB2SH, SH2RGB
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix
    from utils.general_utils import inverse_sigmoid
    from scene.gaussian_model import GaussianModel
    import attack

    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- SH ------------------------------------------------------------
    n = 64
    sh = torch.randn(n, 3, 16, generator=g)                       # [..., C, (deg+1)^2] as eval_sh wants
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = sh.numpy()
    out["sh_dirs"] = dirs.numpy()
    for deg in range(4):
        res = eval_sh(deg, sh, dirs)
        out[f"sh_eval_deg{deg}"] = res.numpy()
        out[f"sh_color_deg{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()
    rgb = torch.rand(16, 3, generator=g)
    out["rgb_in"] = rgb.numpy()
    out["rgb2sh"] = RGB2SH(rgb).numpy()
    out["sh2rgb"] = SH2RGB(rgb).numpy()

    # ---- cameras ---------------------------------------------------------
    Rs, Ts, trs, scs, fovs, w2v, proj, full, cpos = [], [], [], [], [], [], [], [], []
    for i in range(4):
        A = torch.randn(3, 3, generator=g).double().numpy()
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = torch.randn(3, generator=g).double().numpy() * 2.0
        trans = np.array([0.0, 0.0, 0.0]) if i < 2 else torch.randn(3, generator=g).double().numpy()
        scale = 1.0 if i < 3 else 1.7
        fovx, fovy = 0.6 + 0.2 * i, 0.5 + 0.15 * i
        V = torch.tensor(getWorld2View2(Q, T, trans, scale)).transpose(0, 1)           # scene/cameras.py:54
        Pm = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)   # :55
        F = (V.unsqueeze(0).bmm(Pm.unsqueeze(0))).squeeze(0)                          # :56
        C = V.inverse()[3, :3]                                                       # :57
        Rs.append(Q); Ts.append(T); trs.append(trans); scs.append(scale); fovs.append([fovx, fovy])
        w2v.append(V.numpy()); proj.append(Pm.numpy()); full.append(F.numpy()); cpos.append(C.numpy())
    out.update(cam_R=np.stack(Rs), cam_T=np.stack(Ts), cam_trans=np.stack(trs), cam_scale=np.array(scs),
               cam_fov=np.array(fovs), cam_world_view=np.stack(w2v), cam_proj=np.stack(proj),
               cam_full=np.stack(full), cam_center=np.stack(cpos))

    # ---- misc ------------------------------------------------------------
    x = torch.rand(32, generator=g) * 0.98 + 0.01
    out["isig_in"] = x.numpy()
    out["isig_out"] = inverse_sigmoid(x).numpy()

    # ---- GaussianModel getters -------------------------------------------
    P = 40
    gm = GaussianModel(3)
    gm._xyz = torch.randn(P, 3, generator=g)
    gm._features_dc = torch.randn(P, 1, 3, generator=g)
    gm._features_rest = torch.randn(P, 15, 3, generator=g)
    gm._scaling = torch.randn(P, 3, generator=g)
    gm._rotation = torch.randn(P, 4, generator=g)
    gm._opacity = torch.randn(P, 1, generator=g)
    gm._objects_dc = torch.randn(P, 1, 16, generator=g)
    for nme in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_objects_dc"):
        out["gm" + nme] = getattr(gm, nme).numpy()
    out["gm_get_scaling"] = gm.get_scaling.numpy()
    out["gm_get_rotation"] = gm.get_rotation.numpy()
    out["gm_get_opacity"] = gm.get_opacity.numpy()
    out["gm_get_features"] = gm.get_features.numpy()
    out["gm_get_objects"] = gm.get_objects.numpy()
    out["gm_get_xyz"] = gm.get_xyz.numpy()
    # get_covariance -> utils/general_utils.py:64-110, whose three helpers allocate with a hard-coded device="cuda".
    # For this call only, torch.zeros is given a wrapper that sends that device request to the CPU; the arithmetic
    # executed is the reference's own.
    _zeros = torch.zeros

    def _zeros_cpu(*a, **k):
        if str(k.get("device", "")) == "cuda":
            k["device"] = "cpu"
        return _zeros(*a, **k)
    torch.zeros = _zeros_cpu
    try:
        for i, mod in enumerate((1.0, 1.3)):
            out[f"gm_get_covariance_{i}"] = gm.get_covariance(mod).numpy()
    finally:
        torch.zeros = _zeros
    out["gm_covariance_mods"] = np.array([1.0, 1.3])

    # ---- PGD step functions (attack.py:25-173) ----------------------------
    class Fake:
        pass

    def fresh():
        f = Fake()
        gg = torch.Generator().manual_seed(77)
        for nme, shp in (("_xyz", (P, 3)), ("_rotation", (P, 4)), ("_opacity", (P, 1)), ("_scaling", (P, 3)),
                         ("_features_rest", (P, 15, 3)), ("_features_dc", (P, 1, 3))):
            t = torch.randn(*shp, generator=gg)
            t.grad = torch.randn(*shp, generator=gg) * 3.0
            setattr(f, nme, t)
        f._opacity.grad.zero_()            # exercises the "norm == 0 -> zero step" branch of the L2 variants
        return f

    base = fresh()
    orig = {n: (getattr(base, n) + 0.3 * torch.randn(getattr(base, n).shape, generator=g)) for n in
            ("_xyz", "_rotation", "_opacity", "_scaling", "_features_rest", "_features_dc")}
    for n in orig:
        out["pgd_in" + n] = getattr(base, n).numpy()
        out["pgd_grad" + n] = getattr(base, n).grad.numpy()
        out["pgd_orig" + n] = orig[n].numpy()
    alpha, eps = 0.5, 0.8
    out["pgd_alpha_eps"] = np.array([alpha, eps])
    single = {"position": "_xyz", "rotation": "_rotation", "opacity": "_opacity", "scaling": "_scaling"}
    for norm in ("linf", "l2"):
        for what, attr in single.items():
            f = fresh()
            getattr(attack, f"gaussian_{what}_{norm}_attack")(f, alpha, eps, orig[attr].clone())
            out[f"pgd_{what}_{norm}"] = getattr(f, attr).numpy()
        f = fresh()
        getattr(attack, f"gaussian_color_{norm}_attack")(f, alpha, eps, orig["_features_rest"].clone(),
                                                        orig["_features_dc"].clone())
        out[f"pgd_color_{norm}_rest"] = f._features_rest.numpy()
        out[f"pgd_color_{norm}_dc"] = f._features_dc.numpy()

    # ---- COLMAP text model -> camera parameters (scene/colmap_loader.py:156-271, scene/dataset_readers.py:68-143) ----
    from scene.colmap_loader import read_intrinsics_text, read_extrinsics_text, qvec2rotmat
    from utils.graphics_utils import focal2fov
    sample = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colmap_sample", "sparse", "0")
    intr = read_intrinsics_text(os.path.join(sample, "cameras.txt"))
    extr = read_extrinsics_text(os.path.join(sample, "images.txt"))
    rows = []
    for key in extr:
        e = extr[key]
        i = intr[e.camera_id]
        R = np.transpose(qvec2rotmat(e.qvec))
        name = os.path.basename(e.name).split(".")[0]
        rows.append((name, i.id, R, np.array(e.tvec), focal2fov(i.params[0], i.width), focal2fov(i.params[1], i.height),
                     i.width, i.height))
    rows.sort(key=lambda r: r[0])                                  # readColmapSceneInfo sorts by image_name
    out["colmap_names"] = np.array([r[0] for r in rows])
    out["colmap_uid"] = np.array([r[1] for r in rows])
    out["colmap_R"] = np.stack([r[2] for r in rows])
    out["colmap_T"] = np.stack([r[3] for r in rows])
    out["colmap_fov"] = np.array([[r[4], r[5]] for r in rows])
    out["colmap_wh"] = np.array([[r[6], r[7]] for r in rows])
    # scene/dataset_readers.py:40-66 getNerfppNorm on those cameras (its CameraInfo only needs .R and .T here)
    from scene.dataset_readers import getNerfppNorm

    class _CI:
        def __init__(self, R, T):
            self.R, self.T = R, T
    norm = getNerfppNorm([_CI(r[2], r[3]) for r in rows])
    out["colmap_nerfnorm_radius"] = np.array(norm["radius"])
    out["colmap_nerfnorm_translate"] = np.asarray(norm["translate"])

    # ---- COLMAP binary model (scene/colmap_loader.py:125-154, :180-244) ---------------------------------------
    # The fixture files are written here from the text sample in COLMAP's published binary layout, WITH 2D points and
    # tracks so that readers have something to skip, then read back by the reference's binary readers.
    import struct
    from scene.colmap_loader import read_intrinsics_binary, read_extrinsics_binary, read_points3D_binary
    bdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colmap_sample_bin", "sparse", "0")
    os.makedirs(bdir, exist_ok=True)
    model_ids = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1}
    with open(os.path.join(bdir, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(intr)))
        for c in intr.values():
            f.write(struct.pack("<iiQQ", c.id, model_ids[c.model], c.width, c.height))
            f.write(struct.pack("<" + "d" * len(c.params), *c.params))
    with open(os.path.join(bdir, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(extr)))
        for j, e in enumerate(extr.values()):
            f.write(struct.pack("<idddddddi", e.id, *e.qvec, *e.tvec, e.camera_id))
            f.write(e.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", j))                          # j 2D points: 0 for the first image
            for q in range(j):
                f.write(struct.pack("<ddq", 10.5 * q, 3.25 + q, -1 if q % 2 else 7))
    pg = torch.Generator().manual_seed(5)
    pts = torch.randn(5, 3, generator=pg).double().numpy()
    cols = torch.randint(0, 256, (5, 3), generator=pg).numpy()
    errs = torch.rand(5, generator=pg).double().numpy()
    with open(os.path.join(bdir, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", 5))
        for i in range(5):
            f.write(struct.pack("<QdddBBBd", 100 + i, *pts[i], *[int(v) for v in cols[i]], errs[i]))
            f.write(struct.pack("<Q", i % 3))                      # track length 0, 1, 2
            for t in range(i % 3):
                f.write(struct.pack("<ii", 1 + t, 4 * t))
    bi = read_intrinsics_binary(os.path.join(bdir, "cameras.bin"))
    be = read_extrinsics_binary(os.path.join(bdir, "images.bin"))
    ids = sorted(bi)
    out["colmapbin_cam_ids"] = np.array(ids)
    out["colmapbin_cam_wh"] = np.array([[bi[k].width, bi[k].height] for k in ids])
    out["colmapbin_cam_model"] = np.array([bi[k].model for k in ids])
    out["colmapbin_cam_params"] = np.stack([np.pad(np.array(bi[k].params, dtype=np.float64), (0, 4 - len(bi[k].params)))
                                            for k in ids])
    iids = sorted(be)
    out["colmapbin_img_ids"] = np.array(iids)
    out["colmapbin_img_q"] = np.stack([be[k].qvec for k in iids])
    out["colmapbin_img_t"] = np.stack([be[k].tvec for k in iids])
    out["colmapbin_img_cam"] = np.array([be[k].camera_id for k in iids])
    out["colmapbin_img_name"] = np.array([be[k].name for k in iids])
    bx, bc, berr = read_points3D_binary(os.path.join(bdir, "points3D.bin"))
    out["colmapbin_pts_xyz"], out["colmapbin_pts_rgb"], out["colmapbin_pts_err"] = bx, bc, berr.reshape(-1)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
