"""Shared helpers for the parity tests."""
import os
import types

import numpy as np
import torch

from conftest import load_golden  # noqa: F401

LAYER_CASES = ["layer_tiny", "layer_rand200", "layer_d300", "layer_ja_train", "layer_ja_bidir", "layer_noedge",
               "layer_ja_full"]       # the whole real DBP-5L ja graph (bidirectional loader form), d = 8


def layer_params(g, device="cpu"):
    return {k[len("param."):]: torch.from_numpy(v).to(device) for k, v in g.items() if k.startswith("param.")}


def layer_grads(g):
    return {k[len("grad."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("grad.")}


def t(a, device="cpu"):
    return torch.from_numpy(np.asarray(a)).to(device)


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the '1e-4 relative' of BASELINE.json north_star."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = max(b.abs().max().item(), 1e-30)
    return ((a - b).abs().max().item()) / denom


def make_args(slope=0.05, comp_op="sub", **kw):
    return types.SimpleNamespace(leaky_relu_w=slope, comp_op=comp_op, opn=comp_op, **kw)


def random_graph(rng, n, nr, e, hub=None):
    w = 1.0 / np.arange(1, n + 1) ** 0.9
    dst = rng.choice(n, size=e, p=w / w.sum())
    if hub is not None:
        dst[: hub] = 3                      # one destination with `hub` in-edges (forces split segments)
    src = rng.integers(0, n, e)
    typ = rng.integers(0, nr, e)
    p = rng.permutation(e)
    return np.stack([dst[p], src[p]]).astype(np.int64), typ[p].astype(np.int64)


def assert_close(got, ref, rtol, atol=1e-6, what=""):
    """|got-ref|_max <= rtol*|ref|_max + atol.  atol covers quantities that are mathematically zero
    (e.g. d loss/d loop_rel under train-mode BN: a constant row shift cancels in the batch mean)."""
    got = torch.as_tensor(got, dtype=torch.float64).cpu()
    ref = torch.as_tensor(ref, dtype=torch.float64).cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    bound = rtol * (ref.abs().max().item() if ref.numel() else 0.0) + atol
    assert err <= bound, "%s: max|err| %.3e > %.3e" % (what, err, bound)


def write_dbp5l_dir(root, g):
    """Write the arrays of tests/golden/dbp5l_ja_el_data.npz as a directory in the DBP-5L on-disk format (entity/<lang>.tsv,
    kg/<lang>-{train,val,test}.tsv, seed_{train,test}_pairs/<l1>-<l2>.tsv with float-formatted ids, relations.txt).  Entity and
    relation NAME files are placeholders of the right line counts: the loaders only count their lines."""
    import os
    for sub in ("entity", "kg", "seed_train_pairs", "seed_test_pairs"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    with open(os.path.join(root, "relations.txt"), "w") as f:
        f.write("".join("r%d\n" % i for i in range(int(g["n_relation_lines"]))))
    for lang in [str(x) for x in g["langs"]]:
        with open(os.path.join(root, "entity", lang + ".tsv"), "w") as f:
            f.write("".join("%s%d\n" % (lang, i) for i in range(int(g[lang + ".num_entity"]))))
        for part in ("train", "val", "test"):
            np.savetxt(os.path.join(root, "kg", "%s-%s.tsv" % (lang, part)), g["%s.%s" % (lang, part)], fmt="%d", delimiter="\t")
    if "seed_pairs" in g:                                       # all KGs: one file per seed pair
        for pr in [str(x) for x in g["seed_pairs"]]:
            for sub in ("seed_train_pairs", "seed_test_pairs"):
                np.savetxt(os.path.join(root, sub, pr + ".tsv"), g["%s.%s" % (sub, pr)].astype(np.float64), fmt="%.1f", delimiter="\t")
    else:
        pair = "-".join(str(x) for x in g["seed_pair"])
        for sub in ("seed_train_pairs", "seed_test_pairs"):
            np.savetxt(os.path.join(root, sub, pair + ".tsv"), g[sub].astype(np.float64), fmt="%.1f", delimiter="\t")
    return root


def expand_rel_act(act, rel_used, nr, loop=True):
    """Sign pattern (> 0) of a relation-side activation the encoder nodes report on their COMPACT relation rows
    (encoder._RelCompact: rows ``rel_used`` of the table, then the loop row when ``loop``) as a bool [nr (+1), w] pattern over all
    rows -- False on rows no edge names: those rows reach no output and receive a zero gradient, so the side of the kink the
    oracle takes for them changes nothing.  ``rel_used`` None: ``act`` already covers every row."""
    m = (act > 0).cpu()
    if rel_used is None:
        return m
    out = torch.zeros((nr + (1 if loop else 0), m.shape[1]), dtype=torch.bool)
    idx = rel_used.cpu()
    out[idx] = m[: len(idx)]
    if loop:
        out[-1] = m[-1]
    return out


def rel_rows(rel_used, nr, loop=True):
    """Row indices (into the full table, loop row = nr) the compact relation side covers: where kink-side counts are meaningful."""
    if rel_used is None:
        return torch.arange(nr + (1 if loop else 0))
    idx = rel_used.cpu()
    return torch.cat((idx, torch.tensor([nr]))) if loop else idx
