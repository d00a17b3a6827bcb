"""Stand-in for the third-party ``torch_scatter`` package -- TEST INFRASTRUCTURE ONLY.

Used solely by tests/golden/gen_golden.py so that the reference (which imports torch_scatter at
src/jmac_model.py:7 and modules/helper/message_passing.py:2, a dependency absent from this image)
can be imported in the build container to capture golden vectors.  It exposes the three functions
the reference calls, with torch_scatter's signatures, on top of the oracle's restatement of the
published composite algorithm.
"""
from oracle.jmac_oracle import scatter_sum as _sum, scatter_softmax as _softmax


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0 and out is None
    return _sum(src, index, dim_size)


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None and reduce in ("sum", "add")
    return _sum(src, index, dim_size)


def scatter_softmax(src, index, dim=0, dim_size=None):
    assert dim == 0
    return _softmax(src, index, dim_size)
