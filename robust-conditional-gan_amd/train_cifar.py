"""CIFAR-10 training entry point with the reference's flag surface and output layout
(cifar10/gan_resnet.py:38-197 flags/globals, :819-1016 loop, checkpoints and samples).

Launch exactly like the reference (``cifar10/run_rcgan.sh`` ...); for more than one GPU start one process
per GPU with ``python -m torch.distributed.run --nproc-per-node N`` -- ``--ngpus`` then has to equal N and
(with --multi_gpu_multi_batch) keeps its reference meaning: global batch = 64*N, iterations = niters/N.
"""
import logging
import os
import sys
import time
from datetime import datetime

import numpy as np

from . import data as D
from .host import Flags, Plot, Saver, latest_checkpoint, load_checkpoint, record_setting, save_images

DATA_DIR = '../data/cifar10/cifar-10-batches-py/'


def define_flags():
    f = Flags()
    f.DEFINE_string("dataset", 'cifar', "Dataset")
    f.DEFINE_string("algorithm", 'rcgan', "Algorithm [rcgan, rcgan-u, biased, unbiased]")
    f.DEFINE_float("alpha", 0.8, "1 - noise level")
    f.DEFINE_string("run", '0', "run name")
    f.DEFINE_string("log_file", None, "logging file")
    f.DEFINE_string("parent_dir", '.', "parent directory for checkpoints")
    f.DEFINE_string("expt_dir", None, "directory for expts")
    f.DEFINE_integer("inception_freq", 2500, "frequncy of inception score calculation")
    f.DEFINE_integer("inception_samples", 50000, "samples per Inception score (gan_resnet.py:962 hard-codes 50000)")
    f.DEFINE_string("inception_logits_fn", None, "package.module:callable -- the Inception-v3 classifier of the Inception score "
                    "(float32 [128,3,32,32] in [-1,1] -> [128, >= 1000] logits); the reference downloads one through TF-GAN")
    f.DEFINE_integer("sample_freq", 2500, "frequncy of dev cost calc. and sample pics")
    f.DEFINE_integer("generated_label_accuracy_freq", 2500, "frequncy of generated label accruacy")
    f.DEFINE_integer("sample_save_freq", 0, "frequncy of saving samples")
    f.DEFINE_integer("batch_size", 64, "batch size")
    f.DEFINE_integer("niters", 50000, "no. of batches")
    f.DEFINE_float("lr", 2.0e-4, "learning rate")
    f.DEFINE_integer("ngpus", 2, "no. of gpus")
    f.DEFINE_boolean("multi_gpu_multi_batch", True, "multiply batch_size with #gpus and divide #iterations by #gpus")
    f.DEFINE_boolean("confuse_init", False, "whether to initialize confusion matrix with identity")
    f.DEFINE_float("confuse_init_diag", 0.2, "intial confusion matrix with diagonal entry")
    f.DEFINE_float("confuse_multiplier", 1.0, "learning rate multiplier for learnable confusion matrix")
    f.DEFINE_boolean("confuse_lr_decay", False, "whether to decay confusion matrix estimation learning rate")
    f.DEFINE_boolean("perm_classifier", False, "whether to real fake classifier or not.")
    f.DEFINE_float("perm_multiplier", 1.0, "whether to real fake classifier or not.")
    f.DEFINE_string("perm_type", 'linear', "type of real fake classifier to use [linear, 2layer].")
    f.DEFINE_boolean("restore", True, "whether to restore from past checkpoint")
    f.DEFINE_boolean("perm_gen_label_acc", False, "min. over label permutations of the generated label accuracy")
    f.DEFINE_string("log_level", 'info', "logging level [info, debug]")
    # this build's additions (absent flags keep the reference behaviour)
    f.DEFINE_string("dtype", 'bf16', "activation dtype [bf16, f16 (static loss scale 1024), f32]")
    f.DEFINE_boolean("synthetic", False, "train on the SURVEY 8(d) synthetic data instead of ../data/cifar10")
    f.DEFINE_string("synthetic_kind", 'uniform', "with --synthetic: [uniform] SURVEY 8(d) label-free noise images, [templates] "
                    "class-pattern images (data.template_images) scored by eval_cifar.TemplateClassifier instead of the CIFAR ResNet")
    f.DEFINE_integer("seed", 0, "variable-initialisation seed")
    f.DEFINE_string("data_dir", DATA_DIR, "CIFAR-10 python batches")
    f.DEFINE_integer("sample_every", 0, "if > 0: overrides --sample_freq (dev cost + sample grid period)")
    f.DEFINE_integer("early_checkpoint_every", 1, "checkpoint period during the first 500 iterations (the reference: every one)")
    return f


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    FLAGS = define_flags().parse(argv)
    if FLAGS.log_file is None:
        raise ValueError('flag log_file is required')                         # gan_resnet.py:81-82
    if FLAGS.dataset != "cifar":
        raise ValueError("only --dataset cifar is wired in this entry point")
    logging.basicConfig(filename=FLAGS.log_file, level=logging.DEBUG if FLAGS.log_level == 'debug' else logging.INFO,
                        format='%(asctime)s %(levelname)-8s %(message)s')
    ALGORITHM, ALPHA = FLAGS.algorithm, FLAGS.alpha
    logging.info('alpha = {}'.format(ALPHA))
    C_ALPHA = D.C_ALPHA(ALPHA)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    N_GPUS = FLAGS.ngpus
    if world > 1 and N_GPUS != world:
        raise Exception('--ngpus %d does not match the %d launched ranks' % (N_GPUS, world))
    if world == 1 and N_GPUS != 1:
        logging.warning("--ngpus %d requested but one rank launched: running the %d towers' batch on one GPU", N_GPUS, N_GPUS)
    BATCH_SIZE, ITERS = FLAGS.batch_size, FLAGS.niters
    if FLAGS.multi_gpu_multi_batch:                                           # gan_resnet.py:190-192
        BATCH_SIZE, ITERS = BATCH_SIZE * N_GPUS, ITERS // N_GPUS
    per_rank = BATCH_SIZE // world

    DIR = os.path.join(FLAGS.parent_dir, ALGORITHM + '_alpha' + str(ALPHA) + '_run-' + FLAGS.run + '_' +
                       datetime.now().strftime("%Y%m%d-%H%M%S"))                # gan_resnet.py:115
    if FLAGS.expt_dir is not None:
        DIR = '{}/{}'.format(FLAGS.parent_dir, FLAGS.expt_dir)
    if rank == 0:
        os.makedirs(DIR, exist_ok=True)
        record_setting(os.path.join(DIR, 'scripts'))
    CHECKPOINT_DIR = os.path.join(DIR, 'checkpoint')
    # gan_resnet.py:129-132: the flags override the per-dataset constants of :111-114
    INCEPTION_FREQUENCY = FLAGS.inception_freq
    SAMPLE_FREQUENCY = FLAGS.sample_every if FLAGS.sample_every > 0 else FLAGS.sample_freq
    SAMPLE_SAVE_FREQUENCY = FLAGS.sample_save_freq
    inception_fn = None
    if FLAGS.inception_logits_fn:
        from .inception_score import load_logits_fn
        inception_fn = load_logits_fn(FLAGS.inception_logits_fn)
    elif INCEPTION_FREQUENCY and INCEPTION_FREQUENCY <= FLAGS.niters:
        logging.warning("--inception_freq %d: the Inception score needs the Inception-v3 graph the reference downloads at import "
                        "(common/inception/inception_score_.py:31-48); it is not part of the checkout -- pass "
                        "--inception_logits_fn module:callable to score with your own copy; skipped", INCEPTION_FREQUENCY)

    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from .cifar import N_CRITIC, Z_DIM, CifarRCGAN
    m = CifarRCGAN(algorithm=ALGORITHM, alpha=ALPHA, batch_size=per_rank, lr=FLAGS.lr, dtype=FLAGS.dtype, seed=FLAGS.seed,
                   perm_classifier=FLAGS.perm_classifier, perm_multiplier=FLAGS.perm_multiplier, perm_type=FLAGS.perm_type,
                   confuse_init=FLAGS.confuse_init, confuse_init_diag=FLAGS.confuse_init_diag,
                   confuse_multiplier=FLAGS.confuse_multiplier, confuse_lr_decay=FLAGS.confuse_lr_decay,
                   device=local, world_size=world, rank=rank)

    # data: label noise drawn from the global numpy stream exactly as the reference does (unseeded there)
    if FLAGS.synthetic:
        tx, ty = D.synthetic_cifar(50000, 1234, FLAGS.synthetic_kind)
        vx, vy = D.synthetic_cifar(10000, 1235, FLAGS.synthetic_kind)
        train_gen = D.cifar_generator(tx, ty, BATCH_SIZE, C_ALPHA)
        dev_gen = D.cifar_generator(vx, vy, BATCH_SIZE, C_ALPHA)
    else:
        train_gen, dev_gen = D.load(BATCH_SIZE, FLAGS.data_dir, C_ALPHA)
    gen = D.inf_train_gen(train_gen)
    gen_G = D.inf_train_gen_G(train_gen, 2)
    from .dp import shard_rows
    sh = lambda a: shard_rows(a, rank, world)

    saver = Saver(max_to_keep=5)
    if FLAGS.restore:
        ckpt = latest_checkpoint(CHECKPOINT_DIR)
        if ckpt:
            logging.info('restore model from: {}...'.format(ckpt))
            m.load_state_dict(load_checkpoint(ckpt))
    plot = Plot()
    fixed_noise = np.random.normal(size=(100, Z_DIM)).astype('float32')       # gan_resnet.py:822
    fixed_labels = np.array([k for k in range(10) for _ in range(10)], dtype='int32')

    def d_feed(batch):
        images, labels, rnd, bia, inv = batch
        second = rnd if ALGORITHM in ("biased", "unbiased") else bia
        return dict(images=sh(images), labels=sh(labels), labels_random=sh(rnd), labels_biased=sh(bia),
                    inv_weights=sh(inv), labels_all=np.concatenate([sh(labels), sh(second)]))

    # generated-label accuracy (gan_resnet.py:424-455, 847-861): 1000 samples, 100 per class, ONE classifier batch
    label_100_list = [label for label in range(10) for _ in range(10)]
    GEN_ACC_FREQ = FLAGS.generated_label_accuracy_freq
    acc_state = {"clf": None, "max": 0.0}

    def save_samples(n):
        # the reference draws these latents with TensorFlow's generator (Generator(noise=None), gan_resnet.py:847-861): a private
        # stream, so that evaluating does not shift the numpy stream the data and label noise come from
        all_samples = [m.sample(label_100_list, is_rs.normal(size=(100, Z_DIM)).astype('float32')) for _ in range(int(n / 100))]
        all_samples = ((np.concatenate(all_samples, axis=0) + 1.) * (255.99 / 2)).astype('int32')     # gan_resnet.py:858
        return all_samples.reshape((-1, 32, 32, 3)), np.concatenate([label_100_list] * int(n / 100), axis=0)

    def label_accuracy(confusion_matrix=None):
        from .eval_cifar import LabelClassifier, TemplateClassifier, generated_label_accuracy
        if acc_state["clf"] is None:
            templates = FLAGS.synthetic and FLAGS.synthetic_kind == 'templates'
            acc_state["clf"] = TemplateClassifier() if templates else LabelClassifier(local)
        samples, labels = save_samples(1000)
        acc = generated_label_accuracy(samples, labels, confusion_matrix=confusion_matrix, classifier=acc_state["clf"])
        logging.info('generated label accuracy: {}'.format(acc))
        return acc

    def inception_score(n):
        # gan_resnet.py:836-845: 100 samples with uniformly random labels per Generator call, n / 100 calls
        from .inception_score import get_inception_score, samples_as_the_reference_feeds_them
        # the reference draws these labels and latents with TensorFlow's generators (gan_resnet.py:836-838): they must not advance
        # the global numpy stream the data and label-noise draws come from -- a private stream
        all_samples = [m.sample(is_rs.randint(10, size=100).astype('int32'), is_rs.normal(size=(100, Z_DIM)).astype('float32'))
                       for _ in range(int(n / 100))]
        return get_inception_score(samples_as_the_reference_feeds_them(np.concatenate(all_samples, axis=0)), inception_fn)

    is_rs = np.random.RandomState(0x15c0 + rank)
    inception_score_max = 0.0                                                  # gan_resnet.py:917
    from .dp import mean_over_ranks
    pending = []                       # (iteration, ticket) of losses read back asynchronously

    def drain_losses():
        # d_cost / g_cost of EVERY iteration (gan_resnet.py:950-951) without a host-device synchronisation per iteration:
        # the device scalars are copied into a pinned ring behind each iteration and collected here
        for (it_, _), (dc, gc) in zip(pending, m.fetch_losses([t for _, t in pending])):
            plot.plot_at('d_cost', it_, dc)
            plot.plot_at('g_cost', it_, gc)
        del pending[:]

    _random_labels_G, _labels_biased_G = next(gen_G)
    for iteration in range(ITERS):                                             # gan_resnet.py:919-1016
        timed = iteration % 10 == 0 or iteration < 5
        if timed:
            m.ctx.sync()               # launches are asynchronous: drain the queue so sec_per_iter times THIS iteration only
        t0 = time.time()
        if ALGORITHM == 'rcgan-u' and (iteration % 100 == 0 or iteration < 500) and FLAGS.log_level == 'debug':   # :921-925
            np.set_printoptions(precision=3, suppress=True)
            logging.debug('confusion_matrix: ')
            logging.debug('\n{}'.format(m.confusion_matrix_value()))
            np.set_printoptions()
        if 0 < iteration:
            _random_labels_G, _labels_biased_G = next(gen_G)
            m.feed_host("g", labels_random_G=sh(_random_labels_G), labels_biased_G=sh(_labels_biased_G))
            m.g_step(iteration=iteration)
        # the generator is fixed during the critic updates: their N_CRITIC Generator() forwards run as one pass
        # (prepare_critic_fakes), then every critic step consumes its slice
        batches = [next(gen) for _ in range(N_CRITIC)]
        m.feed_host("gf", labels_random_all=np.concatenate([sh(b[2]) for b in batches]))
        m.prepare_critic_fakes()
        # the N_CRITIC critic updates (disc_train_op, gan_resnet.py:928-947): one hand-over of their batches, one captured graph where
        # the engine can (CifarRCGAN.critic_steps), else batch by batch
        m.critic_steps([d_feed(batch) for batch in batches], iteration=iteration)
        m.iteration = iteration + 1
        pending.append((iteration, m.enqueue_losses()))
        if timed:
            drain_losses()
            plot.plot('sec_per_iter', time.time() - t0)
        elif len(pending) >= 512:
            drain_losses()
        if rank == 0 and inception_fn is not None and INCEPTION_FREQUENCY and iteration % INCEPTION_FREQUENCY == INCEPTION_FREQUENCY - 1:
            logging.info('starting inception score computation.')               # gan_resnet.py:960-967
            score = inception_score(FLAGS.inception_samples)
            inception_score_max = max(inception_score_max, score[0])
            plot.plot('inception_50k', score[0])
            plot.plot('inception_50k_std', score[1])
            plot.plot('inception_50k_max', inception_score_max)
            logging.info('finished inception score computation.')
        if rank == 0 and SAMPLE_SAVE_FREQUENCY and iteration % SAMPLE_SAVE_FREQUENCY == SAMPLE_SAVE_FREQUENCY - 1:   # :965-969
            logging.info('starting saving samples.')
            samples_for_save, _ = save_samples(10000)
            np.save(os.path.join(DIR, '_samples_{}'.format(iteration)), samples_for_save)
            logging.info('finished saving samples.')
        if SAMPLE_FREQUENCY and iteration % SAMPLE_FREQUENCY == SAMPLE_FREQUENCY - 1:
            # dev cost (gan_resnet.py:972-990): disc_cost, forward only, over the whole dev set; every rank evaluates its
            # shard of every dev batch (the pass updates the spectral-norm u vectors exactly as the reference's does, so all
            # ranks have to make it), the cost is the mean over towers and batches
            logging.info('starting calculating dev cost.')
            dev_disc_costs = []
            for batch in dev_gen():
                m.feed_host("d", **d_feed(batch))
                dev_disc_costs.append(m.eval_d_cost())
            if dev_disc_costs:
                plot.plot('dev_cost', mean_over_ranks(np.mean(dev_disc_costs), m.ctx.device))
            logging.info('finished calculating dev cost.')
        if rank == 0 and SAMPLE_FREQUENCY and iteration % SAMPLE_FREQUENCY == SAMPLE_FREQUENCY - 1:
            samples = m.sample(fixed_labels, fixed_noise)
            samples = ((samples + 1.) * (255. / 2)).astype('int32')             # gan_resnet.py:831
            save_images(samples.reshape((100, 32, 32, 3)), os.path.join(DIR, 'samples_{}.png'.format(iteration)))
        if rank == 0 and GEN_ACC_FREQ > 0 and iteration % GEN_ACC_FREQ == GEN_ACC_FREQ - 1:            # gan_resnet.py:995-1005
            logging.info('starting calculating generated label accuracy.')
            accuracy = label_accuracy()
            acc_state["max"] = max(acc_state["max"], accuracy)
            plot.plot('gen_label_acc', accuracy)
            plot.plot('gen_label_acc_max', acc_state["max"])
            logging.info('finished calculating generated label accuracy.')
        ECE = max(FLAGS.early_checkpoint_every, 1)
        if rank == 0 and ((iteration < 500 and iteration % ECE == ECE - 1) or (iteration % 1000 == 999)):      # :1007-1014
            drain_losses()
            plot.dir_flush(DIR)
            saver.save(m.state_dict(), CHECKPOINT_DIR, 'model.ckpt', iteration)
        plot.tick()
    if rank == 0 and GEN_ACC_FREQ > 0:                                          # gan_resnet.py:1021-1035
        cm = m.confusion_matrix_value() if FLAGS.perm_gen_label_acc else None
        logging.info('starting calculating %sgenerated label accuracy.' % ('min. permuted ' if cm is not None else ''))
        plot.plot('gen_label_acc', label_accuracy(cm))
        logging.info('finished calculating generated label accuracy.')
    drain_losses()
    if rank == 0 and ITERS:
        plot.dir_flush(DIR)
        saver.save(m.state_dict(), CHECKPOINT_DIR, 'model.ckpt', max(ITERS - 1, 0))
    if acc_state["clf"] is not None:
        acc_state["clf"].close()
    m.ctx.close()
    return DIR


if __name__ == '__main__':
    main()
