"""sedifoam_amd -- MI355X-native (gfx950 / HIP) implementation of sediFoam's CFD-DEM particle hot path.

Product code: csrc/ (HIP kernels + the C-ABI of include/sedifoam_amd.h) and thin host-side mirrors
of the reference's plug-in surfaces (lammps.py: interfaceToLammps/library.h; cloud.py:
lammpsFoam/enhancedCloud.H + dragModels).  No CPU fallback exists.
"""
from ._lib import SfError, lib, exported_symbols  # noqa: F401
from .lammps import Lammps  # noqa: F401
from .cloud import enhancedCloud, dragModel, adjustLampTimestep  # noqa: F401
