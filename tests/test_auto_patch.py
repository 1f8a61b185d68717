"""GSR_PATCH_REFERENCE=1: the reference's ``gaussian_renderer.render`` is rebound to the fused render() of this package
without a line of the reference being edited (diff_gaussian_rasterization._auto_patch_reference).  A stand-in
``gaussian_renderer`` package with the reference's import line (gaussian_renderer/__init__.py:14) and a ``render`` of its
own is imported in a fresh interpreter, in both import orders the reference can produce."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-gaussian-splat-attack_amd")

STANDIN = """
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    return "the reference's own render"
def inside():
    return render
"""

SCRIPT = """
import os, sys
if os.environ["PRE_IMPORT"] == "1":
    import simple_knn._C                      # scene/gaussian_model.py:17 runs before attack.py:20 in the reference
    import diff_gaussian_rasterization
from gaussian_renderer import render          # reference attack.py:20
import gaussian_renderer
print(render.__module__, gaussian_renderer.render.__module__, gaussian_renderer.inside().__module__,
      gaussian_renderer.GaussianRasterizer.__module__)
"""


@pytest.mark.parametrize("pre_import", ["0", "1"])
@pytest.mark.parametrize("flag", ["0", "1", None])
def test_env_flag_rebinds_the_references_render(tmp_path, pre_import, flag):
    pkg = tmp_path / "gaussian_renderer"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(STANDIN)
    env = dict(os.environ, PRE_IMPORT=pre_import, PYTHONPATH=os.pathsep.join([str(tmp_path), PKG]))
    env.pop("GSR_PATCH_REFERENCE", None)
    if flag is not None:
        env["GSR_PATCH_REFERENCE"] = flag
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    got = out.stdout.split()
    want = "gsplat_attack.renderer" if flag == "1" else "gaussian_renderer"
    # what callers bind is the fused function; the module's own global (its internal calls) stays the reference's
    assert got == [want, want, "gaussian_renderer", "diff_gaussian_rasterization"], got
