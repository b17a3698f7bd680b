"""Generated-label-accuracy classifier, CPU side: the graph decoded from the reference's frozen GraphDef
(scripts/extract_label_classifier.py) is what the oracle interprets and what the product's weight asset holds."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRAPH = os.path.join(ROOT, "tests", "golden", "cifar_label_classifier_graph.json")
ASSET = os.path.join(ROOT, "robust-conditional-gan_amd", "assets", "cifar_label_classifier.npz")


def test_decoded_graph_inventory():
    nodes = json.load(open(GRAPH))
    ops = {}
    for nd in nodes:
        ops[nd["op"]] = ops.get(nd["op"], 0) + 1
    # SURVEY 8f #1 probe: 680 nodes, 31 convs, batch-moment BN (Mean/SquaredDifference/Rsqrt), 2 AvgPool + 2 Pad shortcuts
    assert len(nodes) == 680 and ops["Conv2D"] == 31 and ops["Rsqrt"] == 31 and ops["AvgPool"] == 2 and ops["Pad"] == 2
    assert ops["MatMul"] == 1 and ops["Softmax"] == 1 and ops["Placeholder"] == 1
    assert nodes[0]["name"] == "resnet_test_batch" and nodes[-1]["name"] == "infer_softmax"
    z = np.load(ASSET)
    assert z["conv0|conv"].shape == (3, 3, 3, 16) and z["fc|fc_weights"].shape == (64, 10)
    assert float(z["conv0|batchnorm|add|y"]) == np.float32(1e-3)
    nparams = sum(z[k].size for k in z.files if z[k].dtype.kind == "f" and z[k].ndim >= 1)
    assert 460000 < nparams < 470000          # ~465 k parameters


def test_graph_interpreter_runs_the_reference_graph():
    from oracle import graph_interp as GI
    nodes, consts = GI.load_graph(GRAPH, ASSET)
    rs = np.random.RandomState(0)
    x = rs.randint(0, 256, size=(6, 32, 32, 3))
    p = GI.run(nodes, consts, {"resnet_test_batch": x}, "infer_softmax")
    assert p.shape == (6, 10) and np.allclose(p.sum(1), 1.0) and (p >= 0).all()
    # batch-moment normalisation: the prediction for a sample depends on the batch it is evaluated with
    p2 = GI.run(nodes, consts, {"resnet_test_batch": x[:3]}, "infer_softmax")
    assert not np.allclose(p[:3], p2)
    # intermediate fetch: stage-2 entry halves the resolution and doubles the channels (stride-2 conv + padded shortcut)
    assert GI.run(nodes, consts, {"resnet_test_batch": x}, "conv2_0/add").shape == (6, 16, 16, 32)
