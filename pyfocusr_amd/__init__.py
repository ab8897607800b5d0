"""pyfocusr_amd — MI355X-native spectral-embedding hot path of pyfocusr.

Same public names as the reference package (`/root/reference/pyfocusr/__init__.py:1-5`):
`Focusr`, `Graph`, `recursive_eig`, `vtk_functions`; plus `eigsort` and the
device binding.  Importing the package does not touch the GPU; the first device
call loads `csrc/libpyfocusr_hip.so` and fails loudly if it (or an MI355X) is
missing — there is no CPU fallback.
"""
from . import vtk_functions
from .eigsort import eigsort
from .focusr import *  # noqa: F401,F403
from .graph import *  # noqa: F401,F403
from .vtk_functions import PolyMesh, read_vtk_mesh, write_vtk_mesh

__version__ = "0.1.0"
