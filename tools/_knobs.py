"""Schedule-level tuning knobs of jmac_amd.graph for the probes in this directory: the product module holds them as plain
constants; a probe that wants another value sets the module attribute, from these environment variables."""
import os

KNOBS = {"JMAC_SMALL_ITEMS": "INLINE_EDGES_MAX_ITEMS", "JMAC_SMALL_BWD_ITEMS": "SMALL_BWD_MAX_ITEMS", "JMAC_COOP_MIN": "COOP_MIN",
         "JMAC_COOP_MIN_LARGE": "COOP_MIN_LARGE"}


def apply():
    import jmac_amd.graph as G
    out = {}
    for env, attr in KNOBS.items():
        if os.environ.get(env):
            setattr(G, attr, int(os.environ[env]))
        out[env] = getattr(G, attr)
    return out
