"""The reference's CLI surface and on-disk layout on the MI355X engine: a short synthetic run of
cifar10/gan_resnet.py's flags (incl. an ignored unknown flag), sample grid, checkpoint, restore."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_train_cli_layout_and_restore(tmp_path, caplog):
    import rcgan_amd  # noqa: F401
    from rcgan_amd.host import latest_checkpoint, load_checkpoint
    from rcgan_amd.train_cifar import main
    parent = str(tmp_path)
    argv = ["--dataset", "cifar", "--algorithm", "rcgan-u", "--alpha", "0.6", "--run", "0", "--log_file", os.path.join(parent, "log.txt"),
            "--parent_dir", parent, "--expt_dir", "e1", "--ngpus", "1", "--multi_gpu_multi_batch", "--perm_classifier", "--confuse_init",
            "--niters", "3", "--batch_size", "8", "--synthetic", "--sample_freq", "2", "--inception_freq", "2", "--noaux_classifier", "--bogus_flag", "7",
            "--generated_label_accuracy_freq", "2", "--perm_gen_label_acc"]
    import logging
    with caplog.at_level(logging.INFO):
        d = main(argv)
    log = caplog.text
    assert log.count("generated label accuracy: ") == 2 and "min. permuted generated label accuracy" in log
    assert glob.glob(os.path.join(d, "gen_label_acc.jpg"))
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
    assert glob.glob(os.path.join(d, "d_cost.jpg")) and glob.glob(os.path.join(d, "g_cost.jpg"))
    # dev cost over the dev generator at --sample_freq (gan_resnet.py:972-990); the unsupported Inception score is announced
    assert "finished calculating dev cost." in log and glob.glob(os.path.join(d, "dev_cost.jpg"))
    assert "Inception score needs the Inception-v3 graph" in log
    assert ck.endswith("model.ckpt-2") and os.path.exists(os.path.join(d, "checkpoint", "model.ckpt-0.index"))   # every early iteration
    # second launch restores the newest checkpoint (gan_resnet.py:910-914) and keeps training
    # ... this time with a classifier for the Inception score (the reference's comes from a TF-GAN download): the trainer draws
    # --inception_samples random-label samples, scores them and plots inception_50k / _std / _max (gan_resnet.py:836-845,960-967)
    caplog.clear()
    with caplog.at_level(logging.INFO):
        d2 = main(argv + ["--inception_logits_fn", "tests.tools.fake_inception:logits", "--inception_samples", "300"])
    assert d2 == d
    assert "finished inception score computation." in caplog.text and "Inception score needs" not in caplog.text
    assert glob.glob(os.path.join(d, "inception_50k.jpg")) and glob.glob(os.path.join(d, "inception_50k_max.jpg"))
    sd2 = load_checkpoint(latest_checkpoint(os.path.join(d, "checkpoint")))
    assert int(sd2["_opt/Discriminator/step"][0]) == 2 * int(sd["_opt/Discriminator/step"][0])


def test_train_cli_on_class_pattern_images_scores_with_the_stand_in_classifier(tmp_path, caplog):
    """--synthetic --synthetic_kind templates: the class-pattern stand-in data set through the CLI, generated-label accuracy by
    eval_cifar.TemplateClassifier (not the CIFAR ResNet, whose verdict on these images would mean nothing) at
    --generated_label_accuracy_freq and at the end of training (gan_resnet.py:995-1005, 1021-1035)."""
    import logging
    import re
    import rcgan_amd  # noqa: F401
    from rcgan_amd.train_cifar import main
    parent = str(tmp_path)
    argv = ["--algorithm", "rcgan", "--alpha", "0.6", "--log_file", os.path.join(parent, "log.txt"), "--parent_dir", parent, "--expt_dir", "t1",
            "--ngpus", "1", "--multi_gpu_multi_batch", "--niters", "4", "--batch_size", "8", "--synthetic", "--synthetic_kind", "templates",
            "--sample_freq", "0", "--inception_freq", "0", "--generated_label_accuracy_freq", "2", "--early_checkpoint_every", "4"]
    with caplog.at_level(logging.INFO):
        d = main(argv)
    accs = [float(x) for x in re.findall(r"generated label accuracy: ([0-9.]+)", caplog.text)]
    assert len(accs) == 3 and all(0.0 <= a <= 1.0 for a in accs)        # iterations 1, 3 and the final evaluation
    assert glob.glob(os.path.join(d, "gen_label_acc.jpg"))


def test_mnist_cli_layout_restore_and_presets(tmp_path, capsys):
    """mnist/main.py's flag surface (run_rcganu.sh preset + an ignored unknown flag), console lines, output tree
    (script/, samples/train_EE_IIII.png, samples_<epoch>.npy, mnist_<B>_28_28/DCGAN.model-<step>), restore without
    --train, and the vanilla-discriminator CE preset with --add_noise."""
    import rcgan_amd  # noqa: F401
    from rcgan_amd.host import latest_checkpoint, load_checkpoint
    from rcgan_amd.train_mnist import main
    root = str(tmp_path)
    argv = ["--algorithm", "rcgan", "--alpha", "0.3", "--disc_type", "projection", "--estimate_confuse", "--aux_classifier",
            "--noadd_noise", "--noconcat_y", "--spectral_norm", "--max_norm", "--checkpoint_dir", root, "--checkpoint", "e1",
            "--epoch", "2", "--batch_size", "100", "--synthetic", "--synthetic_size", "500", "--save_every", "3",
            "--sample_epochs", "1", "--train", "--recover_epoch", "200", "--recover_batch_size", "6", "--recover_learning_rate", "50"]
    d = main(argv)
    out = capsys.readouterr().out
    assert "Recover Epoch: [99] time:" in out and "Recover Epoch: [199] time:" in out
    rec = glob.glob(os.path.join(d, "recover_bs6_epoch200_lr50", "*", "recover.npz"))
    assert len(rec) == 1
    rz = np.load(rec[0])
    assert rz["y_recover"].shape == (6, 10) and np.allclose(rz["y_recover"].sum(1), 1.0, atol=1e-5)
    assert rz["history"][-1][1] < rz["history"][0][1]         # the mse objective went down
    assert "Epoch: [ 0] [   0/   5]" in out and "d_real:" in out and "[Sample] d_loss:" in out
    assert os.path.exists(os.path.join(d, "script", "command.txt"))
    grids = sorted(glob.glob(os.path.join(d, "samples", "train_*.png")))
    assert grids and os.path.basename(grids[0]) == "train_00_0002.png"
    from PIL import Image
    assert Image.open(grids[0]).size == (280, 280)
    assert os.path.exists(os.path.join(d, "samples", "samples_1.npy")) and not os.path.exists(os.path.join(d, "samples", "samples_0.npy"))
    assert np.load(os.path.join(d, "samples", "samples_1.npy")).shape == (100, 100, 28, 28, 1)
    mdir = os.path.join(d, "mnist_100_28_28")
    ck = latest_checkpoint(mdir)
    assert ck is not None and os.path.basename(ck) == "DCGAN.model-11"
    sd = load_checkpoint(ck)
    assert sd["generator/g_h1_lin/Matrix"].shape == (1034, 6272) and "confusion_logits" in sd
    assert "discriminator/d_h0_conv/w/Adam_1" in sd and np.isfinite(sd["generator/g_h3/w"]).all()
    steps = int(sd["_opt/discriminator/step"][0])
    assert steps == 10 and int(sd["_opt/generator/step"][0]) == 20
    # test mode (no --train): a successful restore skips training and goes straight to recover_labels (main.py:133-140);
    # the checkpoint directory is left exactly as it was
    before = {f: os.path.getmtime(os.path.join(mdir, f)) for f in sorted(os.listdir(mdir))}
    d2 = main([a for a in argv if a != "--train"])
    out2 = capsys.readouterr().out
    assert d2 == d and " [*] Success to read DCGAN.model-11" in out2
    assert "Epoch: [ 0]" not in out2 and "Recover Epoch: [199] time:" in out2
    assert {f: os.path.getmtime(os.path.join(mdir, f)) for f in sorted(os.listdir(mdir))} == before
    sd2 = load_checkpoint(latest_checkpoint(mdir))
    assert int(sd2["_opt/discriminator/step"][0]) == steps
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)
    # test mode without a checkpoint trains first (main.py:137-138)
    d4 = main([("e4" if a == "e1" else a) for a in argv if a != "--train"])
    out4 = capsys.readouterr().out
    assert " [*] Failed to find a checkpoint" in out4 and "Epoch: [ 0] [   0/   5]" in out4
    assert latest_checkpoint(os.path.join(d4, "mnist_100_28_28")) is not None
    assert "mean generated label accuracy=skipped" in out4
    # biased preset: vanilla D, CE loss, real_match, no SN / max-norm; plus the --add_noise schedule
    d3 = main(["--algorithm", "biased", "--alpha", "0.6", "--disc_type", "vanilla", "--loss_fn", "ce", "--real_match",
               "--noestimate_confuse", "--add_noise", "--noise_alpha", "0.3", "--noise_start", "1", "--noise_end", "2",
               "--nospectral_norm", "--nomax_norm", "--checkpoint_dir", root, "--checkpoint", "e2", "--epoch", "2",
               "--batch_size", "64", "--synthetic", "--synthetic_size", "256", "--train", "--recover_epoch", "0",
               "--sample_epochs", "2", "--label_classifier_fn", "tests.tools.fake_mnist_classifier:predict"])
    out3 = capsys.readouterr().out
    # generated-label accuracy through a supplied classifier (utils.py:273-306; the reference's .pb is not in the checkout)
    import re
    mm = re.search(r"######EPOCH=1, mean generated label accuracy=([0-9.]+)", out3)
    assert mm is not None and 0.0 <= float(mm.group(1)) <= 1.0, out3[-400:]
    sd3 = load_checkpoint(latest_checkpoint(os.path.join(d3, "mnist_64_28_28")))
    assert sd3["discriminator/d_h3_lin/Matrix"].shape[1] == 1024 and all(np.isfinite(v).all() for v in sd3.values())
