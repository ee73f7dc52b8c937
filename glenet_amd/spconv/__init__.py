"""Drop-in for `import spconv` / `import spconv.pytorch as spconv`
(pcdet/utils/spconv_utils.py:3-6)."""
from . import conv, core, utils  # noqa: F401
from .core import (SparseConv3d, SparseConvTensor, SparseConvolution, SparseInverseConv3d,  # noqa: F401
                   SparseModule, SparseSequential, SubMConv3d)

__version__ = "2.1.0+glenet_amd"
