import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle.jmac_oracle as orc
from util import load_golden, layer_params, t, make_args
from jmac_amd import ops
from jmac_amd.graph import RelGraph
from jmac_amd.layer import RelationAwareLayer

for case in ["layer_tiny", "layer_rand200"]:
    g = load_golden(case)
    p = layer_params(g)
    X, R = t(g["X"]), t(g["R"])
    ei, et = t(g["edge_index"]), t(g["edge_type"])
    n, d = X.shape
    slope = float(g["slope"])
    rel = orc.transform_relations(p, R, slope, "leaky_relu")
    wt, wb = p["w_att"][:d], p["w_att"][d:]
    Wcat = torch.cat([wt, wb, p["gcn_weight"]], 1)
    PQZ = X @ Wcat
    RR = rel @ Wcat[:, d:]
    a = p["a_att"].reshape(-1)
    nb_ref, sl_ref, pre_ref = orc.layer_pre_bn(p, X, R, ei, et, slope)
    graph = RelGraph(ei.cuda(), et.cuda(), n, rel.shape[0], 4 if case == "layer_tiny" else 256)
    nb = ops.rel_attn_aggregate(PQZ.cuda(), RR.cuda(), a.cuda(), graph, slope, -1, 1.0)
    print(case, "nb err", (nb.cpu() - nb_ref).abs().max().item(), "ref max", nb_ref.abs().max().item())
    pre = ops.rel_attn_aggregate(PQZ.cuda(), RR.cuda(), a.cuda(), graph, slope, rel.shape[0] - 1, 0.5)
    print(case, "pre err", (pre.cpu() - pre_ref).abs().max().item(), "ref max", pre_ref.abs().max().item())
    if case == "layer_tiny":
        print("nb gpu row0", nb.cpu()[0]); print("nb ref row0", nb_ref[0])
        print("nb gpu row1", nb.cpu()[1]); print("nb ref row1", nb_ref[1])
        print("rowptr", graph.rowptr.cpu().tolist())
    # bn
    y = ops.bn_tanh(pre_ref.cuda(), p["bn.weight"].cuda(), p["bn.bias"].cuda(), torch.zeros(d).cuda(), torch.ones(d).cuda(), True)
    yref = torch.tanh(torch.nn.functional.batch_norm(pre_ref, None, None, p["bn.weight"], p["bn.bias"], True))
    print(case, "bn err", (y.cpu() - yref).abs().max().item())
