"""CPU: the C-ABI library loads, exports every symbol include/jmac_hip.h declares, validates arguments
before touching a device, and the product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

import jmac_amd
from jmac_amd import _lib


def test_library_exports_every_declared_symbol():
    names = _lib.header_symbols()
    assert len(names) >= 25
    L = _lib.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(_lib._SIGS) == set(names)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(names) <= exported


def test_argument_validation_needs_no_device():
    L = _lib.lib()
    assert L.jmac_version() >= 100
    assert L.jmac_l1_score_f32(None, 4, None, 4, -1, 3, 4, None, 4, 0, None) == -1          # negative size
    assert L.jmac_l1_score_f32(None, 4, None, 4, 2, 3, 4, None, 4, 0, None) == -1           # null pointers
    assert L.jmac_sim_matrix_f32(ctypes.c_void_p(16), 3, ctypes.c_void_p(16), 4, 2, 2, 4, ctypes.c_void_p(16), 2, None) == -2
    assert L.jmac_csr_build(None, None, -5, 3, 2, None, None, None, None, None, 0, None) == -1
    assert b"workspace" in L.jmac_strerror(-3)
    assert L.jmac_items_max(10, 100, 8, 0) == 10 + 12 + 1
    assert L.jmac_items_max(10, 100, 8, 8) == 10 + 12 + 1 + 3 * 11           # cooperative splits: 3 extra items each
    assert L.jmac_rel_attn_fwd_workspace_bytes(4, 300) >= 4 * 300 * 4 + 32


def test_no_cpu_fallback():
    from jmac_amd.layer import RelationAwareLayer
    from util import make_args
    layer = RelationAwareLayer(8, 8, 8, act=torch.tanh, args=make_args())
    x, r = torch.randn(5, 8), torch.randn(3, 8)
    ei = torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(_lib.JmacError):
        layer(x, r, ei, torch.tensor([0, 1]))
    from jmac_amd import scoring
    with pytest.raises(_lib.JmacError):
        scoring.l1_scores(torch.randn(2, 8), torch.randn(3, 8))


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.abspath(jmac_amd.__file__))
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dp, f)).read()
                assert "oracle" not in text.replace("# oracle", ""), os.path.join(dp, f)


def test_state_dict_keys_match_reference():
    from util import load_golden, make_args
    from jmac_amd.model import JMAC
    g = load_golden("model_small")
    ref_keys = sorted(k[len("state."):] for k in g if k.startswith("state."))
    args = make_args(dim=int(g["d"]), dropout=0.0, num_gcn_layer=2, num_negative=5, margin_align=1.0,
                     margin_completion=5.0, batch_size=40, no_name_info=False, device="cpu")
    m = JMAC(args, g["name_emb"], 2 * int(g["nrel"]), int(g["n1"]) + int(g["n2"]))
    assert sorted(m.state_dict().keys()) == ref_keys
    m.load_state_dict({k: torch.from_numpy(g["state." + k]) for k in ref_keys}, strict=True)


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the product path only: on a box without a HIP device it exits with a message instead of
    timing a CPU stand-in (the cpu_baseline leg is the only place the oracle is timed, beside a GPU run)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "MI355X" in (res.stderr + res.stdout) and res.stdout.strip() == ""


def _bench(*args, env=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, env=e, timeout=300)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one rendezvous address shared by all) -- the dry launch stops each rank before it touches a GPU."""
    import json
    res = _bench("--gpus", "2", "--dry-launch")
    assert res.returncode == 0, res.stderr
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world"] == 2 and l["local_rank"] == l["rank"] for l in lines)
    assert len({l["master"] for l in lines}) == 1 and lines[0]["master"].startswith("127.0.0.1:")


def test_bench_fails_when_a_rank_fails_or_the_world_disagrees():
    # a launcher that started a different number of ranks than --gpus says: refuse instead of reporting n_gpus wrongly
    res = _bench("--gpus", "4", "--dry-launch", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr
    # no HIP device here: every child exits non-zero, and so must the parent
    res = _bench("--gpus", "2", "--steps", "1")
    assert res.returncode != 0


def test_index_range_check_on_host_data():
    """Out-of-range ids raise IndexError like torch indexing in the reference (host data is checked on the host)."""
    import numpy as np
    _lib.check_index_range([0, 3, 2], 4)
    _lib.check_index_range(np.array([], dtype=np.int64), 0)
    _lib.check_index_range(torch.tensor([1, 2]), 3)
    with pytest.raises(IndexError):
        _lib.check_index_range([0, 4], 4, "edge_type")
    with pytest.raises(IndexError):
        _lib.check_index_range(torch.tensor([-1, 2]), 3)
    from jmac_amd.model import _idx
    with pytest.raises(IndexError):
        _idx(np.array([5.0, 1.0]), "cpu", 5)
    assert getattr(_idx([1, 2], "cpu", 5), "_jmac_range_ok") == 5
