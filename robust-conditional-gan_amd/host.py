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
# checkpoints: <dir>/<prefix>-<step>.npz + a TF-style `checkpoint` index file, max_to_keep newest kept.
# Tensor names are the TF variable names plus Adam slots "<var>/Adam", "<var>/Adam_1"
# and "<group>/beta_step".  (Genuine TF-bundle bytes are a "next" item, SURVEY 8f #2.)
# ------------------------------------------------------------------------------------------------------
class Saver:
    def __init__(self, max_to_keep=5):
        self.max_to_keep = max_to_keep
        self.kept = []

    def save(self, tensors, directory, prefix, global_step):
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, "%s-%d" % (prefix, global_step))
        np.savez(path + ".npz", **{k.replace("/", "|"): v for k, v in tensors.items()})
        self.kept.append(path)
        while len(self.kept) > self.max_to_keep:
            old = self.kept.pop(0)
            if os.path.exists(old + ".npz"):
                os.remove(old + ".npz")
        with open(os.path.join(directory, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
            for p in self.kept:
                f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
        return path


def latest_checkpoint(directory):
    idx = os.path.join(directory, "checkpoint")
    if not os.path.exists(idx):
        return None
    with open(idx) as f:
        for line in f:
            if line.startswith("model_checkpoint_path:"):
                name = line.split(":", 1)[1].strip().strip('"')
                p = os.path.join(directory, name)
                return p if os.path.exists(p + ".npz") else None
    return None


def load_checkpoint(path):
    with np.load(path + ".npz") as z:
        return {k.replace("|", "/"): z[k] for k in z.files}
