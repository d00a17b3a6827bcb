"""GPU: the selectable forms of the half-wave forward kernel (env knobs of jmac_amd/csrc/aggregate.hip, read once per process, so
each form runs in its own subprocess) give the SAME results on a graph large enough for the persistent grid: the default
(uniform two-deep pipeline), the guarded two-deep form (JMAC_FWD_HW_DEPTH=2), the LDS-staged hot relation rows
(JMAC_FWD_HOT=1) and the streaming [Q|Z] gathers (JMAC_FWD_NT=1) -- bitwise, they run the same arithmetic in the same order --
and the 64-lane kernel (JMAC_FWD_HW=0) to rounding (a different summation tree across lanes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, **env):
    out = str(tmp_path / (tag + ".npz"))
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, os.path.join(HERE, "hw_variant_worker.py"), out], env=e, check=True, timeout=600)
    return np.load(out)


@pytest.mark.timeout(1800)
def test_half_wave_forms_agree(tmp_path):
    base = _run(tmp_path, "default")
    assert np.isfinite(base["o32"]).all() and np.isfinite(base["o16"]).all() and float(np.abs(base["o32"]).max()) > 0.1
    for tag, env in (("guarded", {"JMAC_FWD_HW_DEPTH": "2"}), ("hot", {"JMAC_FWD_HOT": "1"}), ("nt", {"JMAC_FWD_NT": "1"})):
        v = _run(tmp_path, tag, **env)
        assert np.array_equal(v["o32"], base["o32"]), tag
        assert np.array_equal(v["o16"], base["o16"]), tag
    w = _run(tmp_path, "lanes64", JMAC_FWD_HW="0")
    scale = float(np.abs(base["o32"]).max())
    assert float(np.abs(w["o32"] - base["o32"]).max()) <= 2e-5 * scale
    assert float(np.abs(w["o16"] - base["o16"]).max()) <= 2e-5 * scale
