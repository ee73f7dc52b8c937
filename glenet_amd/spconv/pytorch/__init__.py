"""`import spconv.pytorch as spconv` namespace (spconv 2.x import path)."""
from .. import conv, core, utils  # noqa: F401
from ..core import (SparseConv3d, SparseConvTensor, SparseConvolution, SparseInverseConv3d,  # noqa: F401
                    SparseModule, SparseSequential, SubMConv3d)
