"""mygauhuman_amd -- MI355X-native articulated Gaussian-splat hot path.

Hand-written HIP kernels for gfx950 behind a C ABI (include/gsr.h, libgsr.so), with a PyTorch-ROCm host side that
mirrors the reference's operator surface:

  mygauhuman_amd.diff_gaussian_rasterization  <- submodules/diff-gaussian-rasterization/diff_gaussian_rasterization
  mygauhuman_amd.simple_knn._C.distCUDA2      <- submodules/simple-knn
  mygauhuman_amd.gaussian_renderer.render     <- gaussian_renderer/__init__.py

`install_dropin()` registers those modules under the reference's import names so train.py / render.py style
callers work unmodified.
"""
import importlib
import sys

__version__ = "0.1.0"


def install_dropin():
    """Make `import diff_gaussian_rasterization`, `from simple_knn._C import distCUDA2`, `from knn_cuda import KNN` resolve
    to this package."""
    for theirs, ours in (("diff_gaussian_rasterization", "mygauhuman_amd.diff_gaussian_rasterization"),
                         ("simple_knn", "mygauhuman_amd.simple_knn"),
                         ("simple_knn._C", "mygauhuman_amd.simple_knn._C"),
                         ("knn_cuda", "mygauhuman_amd.knn_cuda")):
        sys.modules[theirs] = importlib.import_module(ours)
