"""Host-side utilities that keep the reference's CLI and on-disk layout drop-in: absl-style flags
(unknown flags ignored, ``--noX`` negation), metric plots, sample grids, script snapshot, checkpoints.

Reference: cifar10/gan_resnet.py:38-79 (flags), cifar10/common/plot.py:20-79, cifar10/common/misc.py:18-26,
215-244; tf.train.Saver(max_to_keep=5) at gan_resnet.py:906-914, 1007-1013.
"""
import collections
import glob
import logging
import os
import pickle
import shutil
import sys

import numpy as np


# ------------------------------------------------------------------------------------------------------
# flags: tf.app.flags (absl, parsed with known_only=True)
# ------------------------------------------------------------------------------------------------------
class Flags:
    def __init__(self):
        self._defs = collections.OrderedDict()

    def DEFINE_string(self, name, default, help=""):
        self._defs[name] = ("string", default, help)

    def DEFINE_integer(self, name, default, help=""):
        self._defs[name] = ("integer", default, help)

    def DEFINE_float(self, name, default, help=""):
        self._defs[name] = ("float", default, help)

    def DEFINE_boolean(self, name, default, help=""):
        self._defs[name] = ("boolean", default, help)

    def DEFINE_list(self, name, default, help=""):
        self._defs[name] = ("list", default, help)

    def parse(self, argv):
        """Returns a namespace.  Unknown flags (and their values) are silently skipped, as absl does with
        known_only=True; booleans accept --x, --nox, --x=true/false."""
        vals = {k: v[1] for k, v in self._defs.items()}
        conv = {"string": str, "integer": int, "float": float}
        i = 0
        while i < len(argv):
            a = argv[i]
            i += 1
            if not a.startswith("-"):
                continue
            name = a.lstrip("-")
            val = None
            if "=" in name:
                name, val = name.split("=", 1)
            if name in self._defs:
                kind = self._defs[name][0]
                if kind == "boolean":
                    vals[name] = True if val is None else val.lower() in ("1", "true", "t", "yes", "y")
                else:
                    if val is None:
                        if i >= len(argv):
                            raise ValueError("flag --%s needs a value" % name)
                        val = argv[i]
                        i += 1
                    vals[name] = val.split(",") if kind == "list" else conv[kind](val)
            elif name.startswith("no") and name[2:] in self._defs and self._defs[name[2:]][0] == "boolean":
                vals[name[2:]] = False
            # else: unknown flag -> ignored
        return type("FlagValues", (), vals)()


# ------------------------------------------------------------------------------------------------------
# metric logger (common/plot.py)
# ------------------------------------------------------------------------------------------------------
class Plot:
    def __init__(self):
        self.since_beginning = collections.defaultdict(dict)
        self.since_last_flush = collections.defaultdict(dict)
        self.iter = 0

    def tick(self):
        self.iter += 1

    def plot(self, name, value):
        self.since_last_flush[name][self.iter] = value

    def plot_at(self, name, iteration, value):
        """plot() for a value that was read back later than the iteration it belongs to (asynchronous loss read-back)."""
        self.since_last_flush[name][iteration] = value

    def dir_flush(self, d, log_pkl=False):
        prints = []
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:          # plotting is cosmetic
            plt = None
        for name, vals in self.since_last_flush.items():
            prints.append("{}: {}".format(name, np.mean(list(vals.values()))))
            self.since_beginning[name].update(vals)
            if plt is not None:
                xs = np.sort(list(self.since_beginning[name].keys()))
                ys = [self.since_beginning[name][x] for x in xs]
                plt.clf()
                plt.plot(xs, ys)
                plt.xlabel("iteration")
                plt.ylabel(name)
                plt.savefig(os.path.join(d, "{}.jpg".format(name.replace(" ", "_"))))
        logging.info("iter {}\n{}".format(self.iter, ", ".join(prints)))
        self.since_last_flush.clear()
        if log_pkl:
            with open(os.path.join(d, "log.pkl"), "wb") as f:
                pickle.dump(dict(self.since_beginning), f, pickle.HIGHEST_PROTOCOL)


# ------------------------------------------------------------------------------------------------------
# sample grid + script snapshot (common/misc.py)
# ------------------------------------------------------------------------------------------------------
def save_images(X, save_path):
    """Tile [n,h,w,3] (or [n,h,w]) images into the squarest grid and write a PNG (misc.py:215-244)."""
    from PIL import Image
    X = np.asarray(X)
    if X.dtype.kind == "f":
        X = (255.99 * X).astype("uint8")
    n = X.shape[0]
    rows = int(np.sqrt(n))
    while n % rows != 0:
        rows -= 1
    nh, nw = rows, n // rows
    h, w = X.shape[1:3]
    img = np.zeros((h * nh, w * nw) + tuple(X.shape[3:]), dtype=np.uint8)
    for k, x in enumerate(X):
        j, i = k // nw, k % nw
        img[j * h:j * h + h, i * w:i * w + w] = np.clip(x, 0, 255).astype(np.uint8)
    Image.fromarray(img).save(save_path)


def record_setting(out, src_dir=None):
    """cp *.py <out>; command.txt (misc.py:18-26)."""
    os.makedirs(out, exist_ok=True)
    src_dir = src_dir or os.getcwd()
    for f in glob.glob(os.path.join(src_dir, "*.py")):
        shutil.copy(f, out)
    with open(os.path.join(out, "command.txt"), "w") as f:
        f.write(" ".join(sys.argv) + "\n")


# ------------------------------------------------------------------------------------------------------
# checkpoints: TensorFlow V2 bundles <dir>/<prefix>-<step>.index + .data-00000-of-00001 (tf_bundle.py) + the
# CheckpointState text file `checkpoint`, max_to_keep newest kept -- what tf.train.Saver writes in the reference
# (cifar10/gan_resnet.py:906-925, mnist/model.py:398-425).  Tensor names are the TF variable names
# plus the Adam slots "<var>/Adam", "<var>/Adam_1" and the optimisers' "beta1_power" / "beta2_power" scalars
# (suffixes _1, _2 in optimiser creation order); "_opt/<group>/step" and "_iteration" are this engine's own exact
# integer counters (TensorFlow ignores names it does not ask for).  ".npz" files of earlier runs still load.
# ------------------------------------------------------------------------------------------------------
def adam_power_tensors(optimisers):
    """optimisers: [(steps taken, beta1, beta2)] in the reference's AdamOptimizer creation order ->
    {"beta1_power": beta1**(t+1), "beta2_power": ..., "beta1_power_1": ...}: the non-slot variables AdamOptimizer keeps
    (initialised to beta, multiplied by beta after every apply_gradients)."""
    out = {}
    for i, (t, b1, b2) in enumerate(optimisers):
        sfx = "" if i == 0 else "_%d" % i
        out["beta1_power" + sfx] = np.float32(float(b1) ** (t + 1))
        out["beta2_power" + sfx] = np.float32(float(b2) ** (t + 1))
    return out


def steps_from_beta_power(power, beta):
    """Inverse of adam_power_tensors for one optimiser (a checkpoint written by TensorFlow has no integer counter)."""
    if not 0.0 < beta < 1.0:
        return 0
    if not power >= float(np.finfo(np.float32).tiny):
        # TensorFlow keeps beta**t in fp32: beta2 = 0.9 underflows after ~830 steps (a zero or subnormal says "many steps",
        # not "none").  The count only feeds Adam's bias correction, which is 1 to fp32 precision long before that, so any
        # large count restores the optimiser exactly where TensorFlow left it.
        return 1 << 20
    return max(int(round(np.log(float(power)) / np.log(float(beta)))) - 1, 0)


class Saver:
    def __init__(self, max_to_keep=5):
        self.max_to_keep = max_to_keep
        self.kept = []

    def save(self, tensors, directory, prefix, global_step):
        from . import tf_bundle
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, "%s-%d" % (prefix, global_step))
        tf_bundle.write_bundle(path, tensors)
        if path in self.kept:
            self.kept.remove(path)
        self.kept.append(path)
        while len(self.kept) > self.max_to_keep:
            old = self.kept.pop(0)
            for ext in (".index", ".data-00000-of-00001", ".npz"):
                if os.path.exists(old + ext):
                    os.remove(old + ext)
        with open(os.path.join(directory, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
            for p in self.kept:
                f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
        return path


def latest_checkpoint(directory):
    from . import tf_bundle
    idx = os.path.join(directory, "checkpoint")
    if not os.path.exists(idx):
        return None
    with open(idx) as f:
        for line in f:
            if line.startswith("model_checkpoint_path:"):
                name = line.split(":", 1)[1].strip().strip('"')
                p = name if os.path.isabs(name) else os.path.join(directory, name)
                return p if tf_bundle.exists(p) or os.path.exists(p + ".npz") else None
    return None


def load_checkpoint(path):
    from . import tf_bundle
    if tf_bundle.exists(path):
        return tf_bundle.read_bundle(path)
    with np.load(path + ".npz") as z:
        return {k.replace("|", "/"): z[k] for k in z.files}
