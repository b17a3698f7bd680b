"""The reference's CLI surface and on-disk layout on the MI355X engine: a short synthetic run of
cifar10/gan_resnet.py's flags (incl. an ignored unknown flag), sample grid, checkpoint, restore."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_train_cli_layout_and_restore(tmp_path):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.host import latest_checkpoint, load_checkpoint
    from rcgan_amd.train_cifar import main
    parent = str(tmp_path)
    argv = ["--dataset", "cifar", "--algorithm", "rcgan-u", "--alpha", "0.6", "--run", "0", "--log_file", os.path.join(parent, "log.txt"),
            "--parent_dir", parent, "--expt_dir", "e1", "--ngpus", "1", "--multi_gpu_multi_batch", "--perm_classifier", "--confuse_init",
            "--niters", "3", "--batch_size", "8", "--synthetic", "--sample_every", "2", "--noaux_classifier", "--bogus_flag", "7"]
    d = main(argv)
    assert os.path.isdir(os.path.join(d, "scripts")) and os.path.exists(os.path.join(d, "scripts", "command.txt"))
    assert os.path.exists(os.path.join(d, "samples_1.png"))
    from PIL import Image
    assert Image.open(os.path.join(d, "samples_1.png")).size == (320, 320)
    ck = latest_checkpoint(os.path.join(d, "checkpoint"))
    assert ck is not None and ck.endswith("model.ckpt-2")
    sd = load_checkpoint(ck)
    assert sd["Generator/G.Block.3.Conv2/Filters"].shape == (3, 3, 256, 256)
    assert "Discriminator/D.Block.1.Conv1/filters/spectral_norm/u" in sd and "confusion_logits" in sd
    assert "Generator/G.Input/W/Adam_1" in sd and np.isfinite(sd["Generator/G.Input/W"]).all()
    assert glob.glob(os.path.join(d, "d_cost.jpg"))
    # second launch restores the newest checkpoint (gan_resnet.py:910-914) and keeps training
    d2 = main(argv)
    assert d2 == d
    sd2 = load_checkpoint(latest_checkpoint(os.path.join(d, "checkpoint")))
    assert int(sd2["_opt/Discriminator/step"][0]) == 2 * int(sd["_opt/Discriminator/step"][0])
