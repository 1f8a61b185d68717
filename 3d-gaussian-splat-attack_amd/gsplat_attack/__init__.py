"""Counterparts of the reference's callers and data formats either side of the rasteriser."""
import os

# View pipelining (gsplat_attack.streams) deals a batch's views over four HIP streams; the HIP runtime maps a process's
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, read at its first call) and streams sharing a queue
# serialise.  Ask for eight unless the user chose a value; harmless when the runtime is already initialised.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def patch_reference(module_name: str = "gaussian_renderer") -> int:
    """Opt-in for a running reference checkout: make its ``render`` the fused one of this package.

    The reference's ``gaussian_renderer.render`` reaches the rasteriser through the classic surface (activated tensors
    built with PyTorch ops first).  This replaces that function object -- in ``gaussian_renderer`` itself and in every
    already-imported module that did ``from gaussian_renderer import render`` (``attack.py:20``) -- by
    ``gsplat_attack.renderer.render``: same signature, same returned dict, raw parameters straight into the kernels.
    Returns the number of bindings replaced.  Call it once after the reference's modules are imported."""
    import importlib
    import sys
    from .renderer import render as fused
    mod = sys.modules.get(module_name) or importlib.import_module(module_name)
    original = getattr(mod, "render")
    n = 0
    for m in list(sys.modules.values()):
        if m is not None and getattr(m, "render", None) is original:
            setattr(m, "render", fused)
            n += 1
    return n
