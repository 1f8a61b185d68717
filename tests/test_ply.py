"""PLY wire format: property names / order / channel-major layout of the reference (scene/gaussian_model.py:377-467),
round trips, combine_splats padding and masks, scene surgery helpers."""
import numpy as np
import pytest
import torch

from gsplat_attack import ply
from gsplat_attack.gaussian_model import GaussianModel
from gsplat_attack.scenes import make_scene


def _model(P=37, seed=0):
    m, _, _ = make_scene("hydrant-1k", P=P, n_views=1)
    return m


def test_header_and_layout_match_the_reference_format(tmp_path):
    m = _model()
    path = str(tmp_path / "sub" / "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert names == ply.gaussian_attribute_names()
    assert len(names) == 6 + 3 + 45 + 1 + 3 + 4 + 16 and all(l.startswith("property float ") for l in lines[3:])
    rec = np.frombuffer(body, dtype=np.dtype([(n, "<f4") for n in names]))
    assert rec.shape[0] == 37
    # channel-major storage: f_rest_j with j = channel*15 + coefficient  (transpose(1,2).flatten in save_ply)
    fr = m._features_rest.detach().numpy()                     # [P,15,3]
    for coeff, ch in ((0, 0), (3, 1), (14, 2), (7, 0)):
        assert np.array_equal(rec[f"f_rest_{ch * 15 + coeff}"], fr[:, coeff, ch])
    assert np.array_equal(rec["f_dc_1"], m._features_dc.detach().numpy()[:, 0, 1])
    assert np.array_equal(rec["opacity"], m._opacity.detach().numpy()[:, 0])           # logit, not sigmoid
    assert np.array_equal(rec["rot_0"], m._rotation.detach().numpy()[:, 0])            # un-normalised w
    assert np.array_equal(rec["obj_dc_5"], m._objects_dc.detach().numpy()[:, 0, 5])
    assert not rec["nx"].any() and not rec["ny"].any() and not rec["nz"].any()


def test_round_trip_is_exact(tmp_path):
    m = _model()
    path = str(tmp_path / "a.ply")
    m.save_ply(path)
    m2 = GaussianModel.load_ply(path)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert a.shape == b.shape and torch.equal(a.detach(), b.detach())
    assert m2.active_sh_degree == 3 and all(p.requires_grad for p in m2.parameters())


def test_load_without_object_features_and_wrong_degree(tmp_path):
    m = _model()
    path = str(tmp_path / "a.ply")
    m.save_ply(path)
    props = ply.read_ply(path)
    for i in range(16):
        props.pop(f"obj_dc_{i}")
    ply.write_ply(path, props)
    m2 = GaussianModel.load_ply(path)
    assert tuple(m2._objects_dc.shape) == (37, 1, 16) and float(m2._objects_dc.abs().max()) == 0.0
    with pytest.raises(AssertionError):
        GaussianModel.load_ply(path, sh_degree=2)               # 45 f_rest values do not fit degree 2 (reference :437)


def test_reader_handles_ascii_doubles_and_extra_elements(tmp_path):
    path = tmp_path / "t.ply"
    path.write_text("ply\nformat ascii 1.0\ncomment hand written\nelement vertex 2\nproperty double x\nproperty float y\n"
                    "property uchar z\nelement face 0\nend_header\n1.5 2.5 3\n-4 5 6\n")
    v = ply.read_ply(str(path))
    assert list(v) == ["x", "y", "z"] and v["x"].tolist() == [1.5, -4.0] and v["z"].tolist() == [3, 6]


def test_combine_splats_pads_truncates_and_masks(tmp_path):
    a, b = _model(10), _model(6)
    pa, pb = str(tmp_path / "target.ply"), str(tmp_path / "background.ply")
    a.save_ply(pa)
    # second file: SH degree 1 only (9 f_rest values) -> padded with zeros up to 45
    props = ply.read_ply(pa)
    keep = [n for n in props if not n.startswith("f_rest_") or int(n.split("_")[-1]) < 9]
    small = {n: props[n][:6] for n in keep}
    from collections import OrderedDict
    ply.write_ply(pb, OrderedDict(small))
    model, masks = ply.combine_splats([pa, pb])
    assert model.get_xyz.shape[0] == 16 and len(masks) == 2
    assert masks[0].sum() == 10 and masks[1].sum() == 6 and not (masks[0] & masks[1]).any()
    assert float(model._objects_dc.abs().max()) == 0.0                       # object features are zeroed (:530)
    rest_b = model._features_rest[masks[1]]                                   # [6,15,3]
    flat = rest_b.transpose(1, 2).reshape(6, 45)
    assert torch.equal(flat[:, :9], torch.tensor(np.stack([props[f"f_rest_{i}"][:6] for i in range(9)], 1)).float())
    assert float(flat[:, 9:].abs().max()) == 0.0
    assert torch.equal(model._xyz[masks[0]].detach(), a._xyz.detach())
    with pytest.raises(ValueError):
        ply.combine_splats([])


def test_removal_and_concat_setup():
    m = _model(12)
    frozen = m.clone()
    mask = torch.zeros(12, dtype=torch.bool)
    mask[:5] = True
    m.removal_setup(mask)                                                     # keeps the 7 unselected
    assert m.get_xyz.shape[0] == 7 and all(p.requires_grad for p in m.parameters())
    assert torch.equal(m._xyz.detach(), frozen._xyz.detach()[5:])
    m.concat_setup("xyz", frozen._xyz[:5].detach(), False)
    assert m._xyz.shape[0] == 12 and not m._xyz.requires_grad
    assert torch.equal(m._xyz[7:].detach(), frozen._xyz[:5].detach())


def test_create_from_pcd_initialisation():
    """Counterpart of scene/gaussian_model.py:130-158: shapes, DC colour, identity rotation, 0.1 opacity and the
    3-NN log-scale (against a brute-force neighbour search)."""
    import numpy as np
    from gsplat_attack.gaussian_model import GaussianModel
    from gsplat_attack.sh import SH2RGB
    rng = np.random.default_rng(4)
    pts = rng.normal(size=(300, 3)).astype(np.float32)
    cols = rng.uniform(size=(300, 3)).astype(np.float32)
    m = GaussianModel.create_from_pcd(pts, cols, generator=torch.Generator().manual_seed(1))
    assert m._xyz.shape == (300, 3) and m._features_dc.shape == (300, 1, 3) and m._features_rest.shape == (300, 15, 3)
    assert m._objects_dc.shape == (300, 1, 16) and m._opacity.shape == (300, 1)
    assert torch.allclose(SH2RGB(m._features_dc[:, 0].detach()), torch.from_numpy(cols), atol=1e-6)
    assert float(m._features_rest.abs().max()) == 0.0
    assert torch.allclose(m.get_opacity.detach(), torch.full((300, 1), 0.1), atol=1e-6)
    assert torch.equal(m.get_rotation.detach(), torch.tensor([[1.0, 0, 0, 0]]).expand(300, 4))
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    d2.sort(axis=1)
    want = np.log(np.sqrt(np.maximum(d2[:, 1:4].mean(axis=1), 1e-7)))
    assert np.allclose(m._scaling.detach().numpy(), np.repeat(want[:, None], 3, 1), atol=1e-4)


def test_written_file_body_equals_the_reference_save_ply_elements(tmp_path):
    """The vertex element of a file written here, byte for byte, against the structured array the reference's own
    save_ply builds for the same tensors (scene/gaussian_model.py:393-411; captured by tests/golden/make_golden_f4.py with
    a recorder in place of the absent plyfile, whose binary_little_endian writer emits exactly that array after the
    header), plus the property order of construct_list_of_attributes (:377-391)."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_f4.npz"))
    t = {n: torch.from_numpy(ref["ply_in" + n]) for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling",
                                                          "_rotation", "_objects_dc")}
    m = GaussianModel.from_tensors(t["_xyz"], t["_features_dc"], t["_features_rest"], t["_scaling"], t["_rotation"],
                                   t["_opacity"], t["_objects_dc"])
    path = str(tmp_path / "p.ply")
    m.save_ply(path)
    head, body = open(path, "rb").read().split(b"end_header\n", 1)
    names = [l.split()[2] for l in head.decode().strip().split("\n")[3:]]
    assert names == [str(n) for n in ref["ply_attribute_names"]] == ply.gaussian_attribute_names()
    assert [d.split(":")[1] for d in ref["ply_elements_descr"]] == ["<f4"] * len(names)
    assert body == ref["ply_elements_bytes"].tobytes()
