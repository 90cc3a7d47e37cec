"""tabcorr_amd: TabCorr's predict() path on AMD Instinct MI355X (gfx950).

Same class surface as johannesulf/TabCorr (``TabCorr``, ``Interpolator``);
the arithmetic runs in hand-written HIP kernels behind a C ABI
(``include/tabcorr_amd.h``), loaded with ctypes.  There is no CPU fallback.
"""

from .tabcorr import TabCorr, symmetric_matrix_to_array
from .interpolator import Interpolator
from .models import Zheng07Model, Leauthaud11Model
from .galtable import GalTypeTable
from .pinned import pinned_empty, pinned_array, is_pinned, pin
from . import synthetic
from . import corrfunc

__version__ = '0.1.0'
__all__ = ['TabCorr', 'Interpolator', 'Zheng07Model', 'Leauthaud11Model',
           'GalTypeTable', 'pinned_empty', 'pinned_array', 'is_pinned', 'pin',
           'symmetric_matrix_to_array', 'synthetic', 'corrfunc']
