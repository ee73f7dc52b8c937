"""`spconv.conv.SparseConvolution` is what pcdet/utils/spconv_utils.py:19 isinstance-checks."""
from .core import (SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d)  # noqa: F401
