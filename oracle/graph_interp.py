"""TEST INFRASTRUCTURE (oracle): a numpy interpreter for the frozen TensorFlow GraphDef of the reference's
generated-label-accuracy classifier (cifar10/resnet-110/graph_optimized.pb, run by gan_resnet.py:424-455 through
tf.import_graph_def).  It executes the decoded node list (tests/golden/cifar_label_classifier_graph.json + the Const
tensors in the product asset) node by node with the documented TF-1.x semantics of the 15 op types the graph uses, in
float64 -- i.e. it is the reference's own graph, not a re-derivation of its architecture.

Pinning: "parity unpinned" numerically (no TensorFlow in this image to produce golden outputs); structurally pinned --
every node, attribute and constant comes from the reference's file (scripts/extract_label_classifier.py).
"""
import json

import numpy as np

from . import nn


def load_graph(json_path, npz_path):
    with open(json_path) as f:
        nodes = json.load(f)
    z = np.load(npz_path)
    consts = {k.replace("|", "/"): z[k] for k in z.files}
    return nodes, consts


def _avgpool_valid(x, ksize, strides):
    kh, kw, sh, sw = ksize[1], ksize[2], strides[1], strides[2]
    n, h, w, c = x.shape
    oh, ow = (h - kh) // sh + 1, (w - kw) // sw + 1
    out = np.zeros((n, oh, ow, c), x.dtype)
    for i in range(kh):
        for j in range(kw):
            out += x[:, i:i + sh * oh:sh, j:j + sw * ow:sw, :]
    return out / (kh * kw)


def run(nodes, consts, feed, fetch):
    """feed: {placeholder name: array}.  Returns the value of node ``fetch`` (float64)."""
    val = {}

    def get(name):
        name = name.split(":")[0].lstrip("^")
        return val[name]

    for nd in nodes:
        op, name, a = nd["op"], nd["name"], nd["attr"]
        ins = nd["inputs"]
        if op == "Placeholder":
            v = np.asarray(feed[name], np.float64)
        elif op == "Const":
            v = consts[name]
            v = v.astype(np.float64) if v.dtype.kind == "f" else v
        elif op == "Conv2D":
            assert a["data_format"] == "NHWC" and a["padding"] == "SAME" and a["dilations"] == [1, 1, 1, 1]
            assert a["strides"][1] == a["strides"][2]
            v = nn.conv2d_fwd(get(ins[0]), get(ins[1]), a["strides"][1])
        elif op == "Mean":
            v = get(ins[0]).mean(axis=tuple(int(i) for i in np.atleast_1d(get(ins[1]))), keepdims=bool(a.get("keep_dims", False)))
        elif op == "StopGradient":
            v = get(ins[0])
        elif op == "SquaredDifference":
            v = (get(ins[0]) - get(ins[1])) ** 2
        elif op == "Squeeze":
            v = np.squeeze(get(ins[0]), axis=tuple(a["squeeze_dims"]))
        elif op == "Add":
            v = get(ins[0]) + get(ins[1])
        elif op == "Sub":
            v = get(ins[0]) - get(ins[1])
        elif op == "Mul":
            v = get(ins[0]) * get(ins[1])
        elif op == "Rsqrt":
            v = 1.0 / np.sqrt(get(ins[0]))
        elif op == "Relu":
            v = np.maximum(get(ins[0]), 0.0)
        elif op == "AvgPool":
            assert a["padding"] == "VALID" and a["data_format"] == "NHWC"
            v = _avgpool_valid(get(ins[0]), a["ksize"], a["strides"])
        elif op == "Pad":
            v = np.pad(get(ins[0]), [tuple(int(q) for q in r) for r in get(ins[1])], mode="constant")
        elif op == "MatMul":
            assert not a.get("transpose_a") and not a.get("transpose_b")
            v = get(ins[0]) @ get(ins[1])
        elif op == "Softmax":
            z = get(ins[0])
            e = np.exp(z - z.max(axis=-1, keepdims=True))
            v = e / e.sum(axis=-1, keepdims=True)
        else:
            raise NotImplementedError("GraphDef op %s (node %s)" % (op, name))
        val[name] = v
        if name == fetch:
            return v
    raise KeyError(fetch)
