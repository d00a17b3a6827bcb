"""The driver's entry points in ONE fresh process, build() before smoke(): build() maps libjmac_hip.so before anything has used
torch's HIP runtime, which is the load order that once gave "no ROCm-capable device" on the library's first launch
(jmac_amd/_lib.py:lib imports torch first so that both sides share one runtime)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("code", [
    "import __graft_entry__ as g; g.build(); g.smoke()",
    "import __graft_entry__ as g; g.smoke()",
    # the library mapped before torch is even imported by the caller
    "import jmac_amd; jmac_amd.lib(); import __graft_entry__ as g; g.smoke()",
])
def test_entry_points_in_one_process(code):
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "smoke: layer fwd rel err" in res.stdout
