"""jmac_amd -- MI355X (gfx950) implementation of JMAC's relation-aware GNN layer and scoring hot path.

Everything that computes lives in ``libjmac_hip.so`` (hand-written HIP, C ABI in include/jmac_hip.h);
this package is the Python host side that mirrors the reference's operator interfaces for that path:

    jmac_amd.layer.RelationAwareLayer      drop-in for src/jmac_model.py:10-109
    jmac_amd.layer.RelationalAwareLayer    drop-in for JMAC_DBPv1/models/jmac_model.py:20-113
    jmac_amd.model.JMAC                    encoder + scoring call sites of src/jmac_model.py:125-380
    jmac_amd.scoring                       l1_scores / filtered_rank / get_neg / alignment_quality
    jmac_amd.scatter                       torch_scatter-compatible scatter_add / scatter / scatter_softmax
    jmac_amd.dist                          destination-sharded multi-GPU layer (RCCL over xGMI)
    jmac_amd.optim                         Adam / AdamW: torch.optim.Adam's update (train.py:406-407) as one launch
"""
from ._lib import JmacError, lib  # noqa: F401

__all__ = ["JmacError", "lib"]
__version__ = "0.1.0"
