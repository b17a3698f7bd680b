#!/usr/bin/env python3
"""Benchmark of the RCGAN G+D training iteration on MI355X (BASELINE.json metric).

Workload (config.workload): CIFAR-10 32x32 SNGAN-projection ResNet RCGAN, per-GPU critic batch 64,
bf16 activations with fp32 master weights / accumulation (BASELINE.json configs[2]).  One "step" is one
reference iteration (cifar10/gan_resnet.py:928-947): 1 generator update on 2*B fakes + N_CRITIC=5
discriminator updates on B real + B fake each; images/sec = 5*B*n_gpus / t_iteration (real images
consumed).  Weak scaling: per-GPU batch fixed, one process per GPU, RCCL all-reduce of the flat gradient
slabs.  Inputs are synthetic and resident in HBM before the timed region; z and the
dequantisation noise are drawn on the device inside the captured step graphs.

Extra objects on the JSON line:
  roofline     dominant kernel (conv_mfma_h8_kernel: fwd + dgrad of the 256-channel 32x32 convs, 256x256 tiles, pixel operand as an
               LDS patch): algorithmic flops of its launches in one iteration / their summed duration measured with HIP events on
               the launch stream, against the dense bf16 MFMA peak; mfma_busy = the matrix pipe's busy fraction by the hardware
               counter (its grids weighted by time), time_share = the kernel's part of the iteration's kernel time,
               conv_mfma_busy_time_weighted = the same counter over EVERY convolution kernel weighted by time (conv2d as a whole) and
               traffic = HBM bytes per launch, all from committed rocprofv3 --pmc / --kernel-trace passes of this same command on
               this same build of the kernels (null / absent otherwise).
  cpu_baseline the PyTorch-CPU restatement of the reference graph (oracle/torch_port.py; kind "port": TensorFlow 1.5 is not
               installable) on all host cores for a bounded sample at the same batch, rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_CRITIC = 5
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 (= fp16) MFMA peak, MI355X_MICROARCH.md
POOL = 30                     # synthetic batches resident on the device (1920 images at B = 64: the critic cannot memorise them in a bench run); a multiple of N_CRITIC: an iteration's batches are consecutive pool slots
BATCH_CRITIC_FAKES = os.environ.get("RCGAN_BATCH_CRITIC_FAKES", "1") == "1"


def synthetic_images(rs, labels, kind=None):
    """Synthetic "real" images, uint8 CHW [n, 3072].  kind "smooth" (default): a fixed low-frequency colour pattern per class
    plus per-image low-pass noise squashed to the pixel range -- natural-image-like second-order statistics that carry the label,
    which the critic cannot separate from the generator's (equally smooth) output within a few hundred updates, so the bench
    runs in the regime training runs in (hinge terms active, d_loss around 1-2, every step back-propagates non-zero
    gradients).  kind "uniform" (round 1/2): U{0..255} per pixel -- the critic saturates the hinge within ~25 iterations and
    5 of 6 steps back-propagate exact zeros (same time per step, unrepresentative values)."""
    kind = kind or os.environ.get("RCGAN_BENCH_IMAGES", "smooth")
    n = len(labels)
    if kind == "uniform":
        return rs.randint(0, 256, size=(n, 3072))
    from rcgan_amd import data as D
    return D.template_images(rs, labels)


def build_pool(m, rank, alpha):
    """Synthetic label / image streams uploaded once; returns device tensors to cycle through."""
    from rcgan_amd import data as D
    B = m.B
    rs = np.random.RandomState(1234 + rank)
    n = POOL * B
    clean = rs.randint(10, size=n)
    images = synthetic_images(rs, clean)
    Cm = D.C_ALPHA(alpha)
    lab, rnd, bia, inv = D.corrupt_labels(clean, Cm, rs)
    dev = m.ctx.device
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    second = rnd[:n] if m.alg in ("biased", "unbiased") else bia[:n]
    pool = dict(images=to(images, torch.int32).reshape(POOL, B, 3072), labels=to(lab, torch.int32).reshape(POOL, B),
                labels_random=to(rnd[:n], torch.int32).reshape(POOL, B), labels_biased=to(bia[:n], torch.int32).reshape(POOL, B),
                inv_weights=to(inv[:n], torch.float32).reshape(POOL, B, 10), second=to(second, torch.int32).reshape(POOL, B))
    rs2 = np.random.RandomState(99 + rank)
    pool["labels_random_G"] = to(rs2.randint(10, size=(POOL, 2 * B)), torch.int32)
    pool["labels_biased_G"] = to(rs2.randint(10, size=(POOL, 2 * B)), torch.int32)
    # packed per-step batches in the engine's feed layout (CifarRCGAN.feed_layout): one copy per step
    host = lambda t: t.cpu().numpy()
    fd, fg = [], []
    for k in range(POOL):
        fd.append(m.pack_feed("d", images=host(pool["images"][k]), labels=host(pool["labels"][k]), labels_random=host(pool["labels_random"][k]),
                              labels_biased=host(pool["labels_biased"][k]), inv_weights=host(pool["inv_weights"][k]),
                              labels_all=np.concatenate([host(pool["labels"][k]), host(pool["second"][k])])))
        fg.append(m.pack_feed("g", labels_random_G=host(pool["labels_random_G"][k]), labels_biased_G=host(pool["labels_biased_G"][k])))
    pool["feed_d"] = torch.from_numpy(np.stack(fd)).to(dev)
    pool["feed_g"] = torch.from_numpy(np.stack(fg)).to(dev)
    # generator labels of N_CRITIC consecutive critic steps starting at pool index s (any s: dcount walks the pool cyclically)
    lr = host(pool["labels_random"])
    pool["feed_gf"] = torch.from_numpy(np.stack([
        m.pack_feed("gf", labels_random_all=np.concatenate([lr[(s0 + t) % POOL] for t in range(N_CRITIC)])) for s0 in range(POOL)])).to(dev)
    torch.cuda.synchronize()
    return pool


def feed_d(m, pool, i):
    m.set_feed("d", pool["feed_d"][i % POOL])         # one device-to-device copy of the packed batch


def feed_g(m, pool, i):
    m.set_feed("g", pool["feed_g"][i % POOL])


def iteration(m, pool, it, dcount):
    """gan_resnet.py:928-947: [G step if it>0] then N_CRITIC D steps (logging-only forward passes excluded)."""
    if it > 0:
        feed_g(m, pool, it)
        m.g_step(iteration=it)
    if BATCH_CRITIC_FAKES:
        # the N_CRITIC generator forwards of this iteration's critic steps as one pass (same G weights, own z / labels /
        # batch-norm statistics each)
        m.set_feed("gf", pool["feed_gf"][dcount[0] % POOL])
        m.prepare_critic_fakes()
    k0 = dcount[0] % POOL
    if BATCH_CRITIC_FAKES and k0 + N_CRITIC <= POOL:
        # the iteration's N_CRITIC packed batches handed over together; where the engine can, the five critic steps are ONE captured
        # graph (CifarRCGAN.critic_steps: same launches, same order), else five d_step() calls
        m.critic_steps(pool["feed_d"][k0:k0 + N_CRITIC], iteration=it)
        dcount[0] += N_CRITIC
        return
    for _ in range(N_CRITIC):
        feed_d(m, pool, dcount[0])
        dcount[0] += 1
        m.d_step(iteration=it)


def kernel_roofline(m, pool, default_workload=True):
    """One eager (un-captured) iteration with every launch of the dominant kernel (conv_mfma_h8_kernel: forward and data
    gradient of the 256-channel 3x3 convolutions at 32x32, 256 x 256 tiles; a layer it does not take runs on conv_mfma_p8_kernel
    under the same profiling id) bracketed by HIP events on the launch stream."""
    from rcgan_amd import _lib as L
    ctx = m.ctx
    saved = m.use_graphs
    m.use_graphs = False
    # (round 6) the critic steps' generator forwards run on a second context (CifarRCGAN.overlap_gf): the kernel's launches are
    # counted on both, each bracketed by events on the stream it is launched on
    ctxs = [ctx] + ([m.ctx2] if getattr(m, "ctx2", None) is not None else [])
    for c in ctxs:
        c.check(c.lib.rcgan_prof_begin(c.h, 4))       # RCGAN_PROF_CONV_P8
    iteration(m, pool, 1, [0])
    n, ms, fl, fx, nbn = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
    for c in ctxs:
        n1, ms1, fl1, fx1, nb1 = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        c.check(c.lib.rcgan_prof_end(c.h, C.byref(n1), C.byref(ms1), C.byref(fl1)))
        c.check(c.lib.rcgan_prof_executed_flops(c.h, C.byref(fx1)))
        c.check(c.lib.rcgan_prof_bn_in_launches(c.h, C.byref(nb1)))
        n.value += n1.value
        ms.value += ms1.value
        fl.value += fl1.value
        fx.value += fx1.value
        nbn.value += nb1.value
    m.use_graphs = saved
    if n.value == 0 or ms.value <= 0:
        return None
    achieved = fl.value / (ms.value * 1e-3) / 1e12
    # HBM bytes per launch of this kernel: measured in separate rocprofv3 --pmc passes of this same command (FETCH_SIZE and
    # WRITE_SIZE cannot share a pass; corrections as MI355X_MICROARCH.md prescribes) and committed with the profile
    traffic = None
    import glob
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    tfs = sorted(glob.glob(os.path.join(pdir, "r*_pmc_traffic_conv_p8.json")) + glob.glob(os.path.join(pdir, "r*_pmc_traffic_conv_h8.json")),
                 key=os.path.basename)
    mfma_busy, busy_extra = None, {}
    if default_workload and tfs:                         # the counters were collected on the default workload only; newest round
        with open(tfs[-1]) as f:
            tj = json.load(f)
        # ... and only a measurement of THIS build of the kernels counts (the profile records the hash of the sources it ran)
        if tj.get("source_sha16") == L.source_hash():
            traffic = float(tj["traffic_bytes_per_launch"])
    # the matrix pipe's busy fraction by the hardware counter (SQ_VALU_MFMA_BUSY_CYCLES / 4 x SQ_BUSY_CU_CYCLES), collected in its own
    # rocprofv3 --pmc pass of this command and committed with the profile; same gate
    bfs = sorted(glob.glob(os.path.join(pdir, "r*_pmc_mfma_busy.json")), key=os.path.basename)
    if default_workload and bfs:
        with open(bfs[-1]) as f:
            bj = json.load(f)
        if bj.get("source_sha16") == L.source_hash():
            mfma_busy = round(float(bj["mfma_busy"]), 4)
            # utilisation of conv2d as a WHOLE, not of its best kernel: every convolution kernel that issues MFMAs, weighted by
            # the time it takes in the kernel-trace pass of the same profile run (scripts/pmc_busy_table.py)
            for k in ("mfma_busy_weighting", "time_share", "conv_mfma_busy_time_weighted", "conv_time_share"):
                if k in bj:
                    busy_extra[k] = round(bj[k], 4) if isinstance(bj[k], float) else bj[k]
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "mfma_busy": mfma_busy,
            "kernel": "conv_mfma_h8_kernel", "launches_per_iteration": n.value,
            "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
            "flops_per_launch_avg": fl.value / n.value,
            # algorithmic = the reference's formulation; the upsample-3x3 layers run in their sub-pixel form (four 2x2
            # convolutions with summed filters): the matrix cores execute 4/9 of those layers' multiply-adds
            "executed_tflops": round(fx.value / (ms.value * 1e-3) / 1e12, 2), "executed_frac": round(fx.value / (ms.value * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            # (round 5) in the forward-only generator pass the kernel also applies the conditional batch norm + ReLU in front of the
            # convolution to its staged input (two of its five launches): their time includes that work, the FLOP count does not --
            # RCGAN_BN_INTO_PATCH=0 gives the plain launches back (and three batch-norm apply launches with them)
            # (counted by the runtime in this very section: rcgan_prof_bn_in_launches)
            "bn_in_patch_launches": nbn.value,
            **busy_extra}


PEAK_F32_TFLOPS = 157.3       # dense fp32 MFMA (v_mfma_f32_32x32x2_f32) peak, MI355X_MICROARCH.md


def kernel_roofline_f32(m, pool):
    """--dtype f32 (the reference's own precision): the dominant kernel is the fp32 gather GEMM (gemm_gather_kernel, conv_direct.hip) --
    every convolution and dense layer, forward, data and filter gradient, on v_mfma_f32_32x32x2_f32.  One eager iteration with each
    of its launches bracketed by HIP events; priced against the dense fp32 matrix peak."""
    ctx = m.ctx
    saved = m.use_graphs
    m.use_graphs = False
    ctx.check(ctx.lib.rcgan_prof_begin(ctx.h, 7))       # RCGAN_PROF_GATHER_F32
    iteration(m, pool, 1, [0])
    n, ms, fl = C.c_int(0), C.c_double(0), C.c_double(0)
    ctx.check(ctx.lib.rcgan_prof_end(ctx.h, C.byref(n), C.byref(ms), C.byref(fl)))
    m.use_graphs = saved
    if n.value == 0 or ms.value <= 0:
        return None
    achieved = fl.value / (ms.value * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_TFLOPS, 4),
            "traffic": None, "kernel": "gemm_gather_kernel (fp32 matrix cores)", "launches_per_iteration": n.value,
            "avg_launch_us": round(ms.value * 1e3 / n.value, 2), "flops_per_launch_avg": fl.value / n.value,
            "time_in_kernel_ms_per_iteration": round(ms.value, 3)}


def comm_profile(m, pool):
    """One eager iteration with every all-reduce group of the in-ABI communicator bracketed by HIP events on the stream it is issued
    on: (groups per iteration, their summed duration in ms, bytes this rank hands over per iteration).  The duration of an all-reduce
    includes waiting for the slowest rank to arrive."""
    ctx = m.ctx
    saved = m.use_graphs
    m.use_graphs = False
    ctx.check(ctx.lib.rcgan_prof_begin(ctx.h, 6))       # RCGAN_PROF_ALLREDUCE
    iteration(m, pool, 1, [0])
    n, ms, by = C.c_int(0), C.c_double(0), C.c_double(0)
    ctx.check(ctx.lib.rcgan_prof_end(ctx.h, C.byref(n), C.byref(ms), C.byref(by)))
    m.use_graphs = saved
    return n.value, ms.value, by.value


def effective_cores():
    """Cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU boxes report 256
    logical CPUs under a 16-CPU quota; 256 threads on 16 CPUs ran the same step 180x slower than 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(alpha, batch):
    """The same iteration on the host cores (kind "port": TensorFlow 1.5, the reference's CPU path, is not installable):
    oracle/torch_port.py -- the PyTorch-CPU restatement of the reference graph (autograd + TF-form Adam, fp32) -- on all
    usable cores (scheduler affinity capped by the cgroup CPU quota) at the benchmark's own per-GPU batch.  Bounded sample
    (SURVEY 8d): one warm-up and THREE timed (D step, G step) pairs; the iteration is 5 D steps + 1 G step, so
    images/sec = 5*B / (5*mean t_D + mean t_G) -- ``value``: the work the GPU run does.
    ``reference_faithful``: the reference's D-step session.run also fetches gen_cost (gan_resnet.py:936-947), i.e. one more
    Generator(2B) + Discriminator(2B) forward per critic step that only feeds the log; timed (3 samples) and added:
    5*B / (5*(t_D + t_log) + t_G).
    ``mnist_cfg1``: BASELINE configs[0] (MNIST RCGAN B=64, alpha 0.5): iteration = 1 D + 2 G updates (mnist/model.py:347-372)
    by the numpy oracle (oracle/mnist.py) with its BLAS threads capped at the same core count, 5 timed iterations after a
    warm-up; the faithful variant adds three forward evaluations of the loss graph for the reference's five logging-only
    .eval() calls (model.py:374-398: 3 x [G + D(fake)] + 2 x D(real); the stand-in runs one D(real) forward more)."""
    from oracle import cifar as oc
    from oracle.torch_port import CifarTorchTrainer
    B = batch
    rs = np.random.RandomState(0)
    P, U = oc.init_params(0, "rcgan")
    Cm = oc.c_alpha(alpha)
    cores = effective_cores()
    torch.set_num_threads(cores)
    cfg = dict(algorithm="rcgan", C=Cm)

    def batches():
        lab = rs.randint(10, size=B)
        db = dict(real=oc.preprocess_real(synthetic_images(rs, lab), rs.uniform(0, 1 / 128., size=(B, 3072))).astype(np.float32),
                  labels=lab, labels_random=rs.randint(10, size=B), labels_biased=rs.randint(10, size=B),
                  inv_weights=np.linalg.inv(Cm)[lab], z=rs.randn(B, 128))
        gb = dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B), z=rs.randn(2 * B, 128))
        return db, gb
    tr = CifarTorchTrainer(P, U, cfg, lr=2e-4)
    db, gb = batches()
    tr.d_step(db)                      # warm-up (thread pool, oneDNN primitive caches)
    tr.g_step(gb)
    tds, tgs, tls = [], [], []
    for _ in range(3):
        db, gb = batches()
        t0 = time.time()
        tr.d_step(db)
        t1 = time.time()
        tr.g_step(gb)
        t2 = time.time()
        with torch.no_grad():          # the logging-only fetch of gen_cost inside the reference's D-step run
            tr.net.gen_cost(cfg, gb)
        tr.net.U_new = {}
        t3 = time.time()
        tds.append(t1 - t0), tgs.append(t2 - t1), tls.append(t3 - t2)
    td, tg, tl = float(np.mean(tds)), float(np.mean(tgs)), float(np.mean(tls))
    out = {"value": round(5 * B / (5 * td + tg), 3), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "PyTorch-CPU restatement (oracle/torch_port.py), CIFAR RCGAN B=%d fp32, %d threads: 3 timed (D step, G step) pairs after one "
                     "warm-up, mean D %.2fs G %.2fs; iteration = 5 D + 1 G" % (B, cores, td, tg),
           "reference_faithful": {"value": round(5 * B / (5 * (td + tl) + tg), 3), "unit": "images/sec",
                                  "sample": "the same + the gen_cost forward the reference's D-step session.run fetches for its log "
                                            "(gan_resnet.py:936-947): G(2B)+D(2B) forward, mean %.2fs per critic step" % tl}}
    try:
        out["mnist_cfg1"] = cpu_baseline_mnist(cores)
    except Exception as e:             # the MNIST line is an extra: never lose the bench line over it
        out["mnist_cfg1"] = {"error": repr(e)}
    return out


def cpu_baseline_mnist(cores, B=64, alpha=0.5, iters=5):
    """BASELINE configs[0]: MNIST 28x28 RCGAN (run_rcgan.sh), bs=64, 50 % uniform label noise, CPU path."""
    from threadpoolctl import threadpool_limits
    from oracle import labels as LB
    from oracle import mnist as om
    rs = np.random.RandomState(1)
    Cm = LB.one_coin(alpha)
    eye = np.eye(10, dtype=np.float32)
    cfg = dict(algorithm="rcgan", disc_type="projection", estimate_confuse=False, loss_fn="hinge", perm_regularizer=True, perm_multiplier=10.0,
               spectral_norm=True, C=Cm, concat_y=False, concat_y_layers=(), max_norm=True, confuse_multiplier=10.0)

    def batch():
        yr = rs.randint(10, size=B)
        return dict(images=rs.rand(B, 28, 28, 1).astype(np.float32), z=rs.uniform(-1, 1, size=(B, 100)).astype(np.float32),
                    y_real=eye[yr], y_gen=eye[rs.randint(10, size=B)], y_fake=eye[rs.randint(10, size=B)],
                    y_real_weights=np.linalg.inv(Cm)[yr].astype(np.float32))
    P, S, U = om.init_params(0, "projection", False, True, True, ())
    tr = om.Trainer(P, S, U, cfg)
    with threadpool_limits(limits=cores):
        b = batch()
        tr.d_step(b), tr.g_step(b), tr.g_step(b)
        ts, tl = [], []
        for _ in range(iters):
            b = batch()
            t0 = time.time()
            tr.d_step(b), tr.g_step(b), tr.g_step(b)
            t1 = time.time()
            for _ in range(3):
                om.losses(om.Net(tr.P, {k: v.copy() for k, v in tr.S.items()}, dict(tr.U), "d", cfg, np.float32), b)
            t2 = time.time()
            ts.append(t1 - t0), tl.append(t2 - t1)
    t, l = float(np.mean(ts)), float(np.mean(tl))
    return {"value": round(B / t, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "numpy oracle (oracle/mnist.py), MNIST RCGAN B=%d fp32, SN projection D, %d timed iterations (1 D + 2 G updates) after one "
                      "warm-up, mean %.2fs" % (B, iters, t),
            "reference_faithful": {"value": round(B / (t + l), 3), "unit": "images/sec",
                                   "sample": "the same + 3 forward evaluations of the loss graph per iteration for the reference's five "
                                             "logging-only evals (model.py:374-398), mean %.2fs" % l}}


def launcher_command(args, argv, env):
    """The command that starts the ranks of ``--gpus N`` when this process is not one of them already (no WORLD_SIZE in the
    environment): one process per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1.  None when this
    process IS a rank (started by a launcher) or N = 1."""
    if args.gpus <= 1 or "WORLD_SIZE" in env:
        return None
    port = args.master_port
    if not port:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    rest, skip = [], False
    for a in argv:                      # the ranks get the same flags minus the launcher's own
        if skip:
            skip = False
        elif a == "--master-port":
            skip = True
        elif not a.startswith("--master-port=") and a != "--dry-run":
            rest.append(a)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + rest


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU critic batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="activation dtype: bf16 (BASELINE configs[2], default), f16 (configs[4]) or f32 (the reference's own precision: every "
                         "layer on the fp32 matrix cores; its roofline object prices the gather GEMM against the 157 TFLOP/s fp32 peak)")
    ap.add_argument("--algorithm", default="rcgan")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-stub", type=int, default=0,
                    help="single GPU only: run the world-size-N data-parallel step schedule (one whole-slab gradient bucket per optimiser group and "
                         "step, optimiser inside the captured graph) against the in-ABI test-double communicator -- measures what the schedule itself costs; no RCCL traffic")
    ap.add_argument("--dp-stub-gbps", type=float, default=0.0,
                    help="--dp-stub cost model: bus bandwidth in GB/s; every all-reduce of b bytes then occupies its stream for "
                         "latency + 2(N-1)/N * b / bandwidth (0 = free: the schedule's own cost only)")
    ap.add_argument("--dp-stub-lat-us", type=float, default=0.0, help="--dp-stub cost model: latency per all-reduce group in microseconds")
    ap.add_argument("--bucket-dtype", default=None, choices=["f32", "bf16"],
                    help="dtype the gradient buckets travel in (default f32, or RCGAN_DP_BUCKET_DTYPE)")
    ap.add_argument("--lr", type=float, default=2e-4, help="Adam learning rate (reference: 2e-4, gan_resnet.py:--lr)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the ranks --gpus N starts (default: a free one)")
    ap.add_argument("--dry-run", action="store_true", help="print the launcher command --gpus N would start, and exit")
    args = ap.parse_args()
    default_wl = args.batch == 64 and args.dtype == "bf16" and args.algorithm == "rcgan"

    # --gpus N outside a launcher: start the N ranks as CHILD processes (this process has made no GPU call yet and never
    # replaces itself) and relay their output; rank 0's JSON line is the last line of it
    cmd = launcher_command(args, sys.argv[1:], os.environ)
    if args.dry_run:
        print(json.dumps({"launcher": cmd}))
        return 0
    if cmd is not None:
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.run(cmd, env=env).returncode

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: the line would not describe the run" % (args.gpus, world))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN

    alpha = 0.6
    m = CifarRCGAN(algorithm=args.algorithm, alpha=alpha, batch_size=args.batch, dtype=args.dtype, seed=0, lr=args.lr,
                   device=local, use_graphs=not args.no_graphs, device_rng=True,
                   world_size=(args.dp_stub if args.dp_stub > 1 else world), rank=rank, comm=("stub" if args.dp_stub > 1 else None),
                   grad_bucket_dtype=args.bucket_dtype,
                   stub_model=((args.dp_stub_gbps, args.dp_stub_lat_us) if args.dp_stub > 1 and (args.dp_stub_gbps or args.dp_stub_lat_us) else None))
    ranks_reported = None
    if m.dp_active:
        rr = C.c_int(0)
        m.ctx.check(m.ctx.lib.rcgan_comm_count(m.ctx.h, C.byref(rr)))
        ranks_reported = rr.value
        if ranks_reported != m.world:
            raise SystemExit("bench.py: the communicator reports %d ranks, the run was started for %d" % (ranks_reported, m.world))
    pool = build_pool(m, rank, alpha)
    dcount = [0]
    warm = max(args.warmup, 2)      # iteration 0 has no G step; graphs are captured on first use
    for it in range(warm):
        iteration(m, pool, it, dcount)
    m.ctx.sync()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        iteration(m, pool, warm + k, dcount)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=m.ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    d_loss, g_loss = m.losses()
    ok = np.isfinite(d_loss) and np.isfinite(g_loss)

    # replicated state must stay bit-identical across ranks (same kernels on the same all-reduced gradients): every rank hashes its
    # parameters after the timed iterations, rank 0 reports whether all hashes agree
    weights_identical = None
    if world > 1:
        import hashlib
        hsh = hashlib.sha256()
        for k, v in sorted(m.get_params().items()):
            hsh.update(k.encode()), hsh.update(np.ascontiguousarray(v).tobytes())
        hashes = [None] * world
        dist.all_gather_object(hashes, hsh.hexdigest()[:16])
        weights_identical = len(set(hashes)) == 1

    comm = comm_profile(m, pool) if m.dp_active else None      # every rank: the extra iteration all-reduces
    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = N_CRITIC * args.batch * world / (dt / args.steps)
        # algorithmic flops of one iteration: 60.858*B GFLOP per GPU
        out = {"metric": "images/sec G+D train step (CIFAR-10 RCGAN bs=64)", "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": warm, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "CIFAR-10 32x32 SNGAN-projection ResNet %s, per-GPU batch %d, iteration = 1 G step (2B fakes) + 5 D steps (B real + B fake)"
                                      % (args.algorithm.upper(), args.batch),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world, "hip_graphs": not args.no_graphs,
                          "gradient_exchange": ("none (single rank)" if m.world == 1 else
                                                "in-ABI %s all-reduce(sum) of the gradient slabs inside the step's graph, %s, optimiser %s"
                                                % ("RCCL" if m.comm_kind == "rccl" else "TEST-DOUBLE (--dp-stub %d: schedule only, no traffic)" % args.dp_stub,
                                                   ("2 buckets per optimiser group and step, the last layers' leaving on the communication stream "
                                                    "during the backward pass (RCGAN_DP_OVERLAP=1)" if getattr(m, "dp_overlap", False) else
                                                    "1 whole-slab %s bucket per optimiser group and step" % m.grad_bucket_dtype),
                                                   "in the graph" if m.dp_adam_in_graph else "after the graph")),
                          "critic_generator_forwards": ("one pass over N_CRITIC x B samples, batch-norm statistics per critic step"
                                                        if BATCH_CRITIC_FAKES else "inside every critic step"),
                          "critic_steps": ("one captured graph of the N_CRITIC steps, optimiser inside each step's last launch (rcgan_sn_bwd_adam)"
                                           if (getattr(m, "critic_graph", False) and getattr(m, "fused_tail", False) and BATCH_CRITIC_FAKES and not args.no_graphs)
                                           else "one graph per step" + (", optimiser inside the step's last launch" if getattr(m, "fused_tail", False) else "")),
                          "iteration_tflops_algorithmic": round(60.858 * args.batch * world / 1e3, 3),
                          "sustained_tflops": round(60.858 * args.batch * world / 1e3 / (dt / args.steps), 2),
                          "losses_finite": bool(ok), "d_loss": round(d_loss, 4), "g_loss": round(g_loss, 4)}}
        if comm is not None:
            if weights_identical is not None:
                out["config"]["rank_weights_bit_identical"] = bool(weights_identical)
            out["config"].update({"communicator_ranks": ranks_reported, "gradient_bucket_dtype": m.grad_bucket_dtype,
                                  "allreduce_groups_per_iteration": comm[0], "allreduce_ms_per_iteration": round(comm[1], 4),
                                  "allreduce_mbytes_per_iteration": round(comm[2] / 1e6, 3)})
            if args.dp_stub > 1:
                out["config"]["stub_link_model"] = ("none (all-reduce is free)" if not (args.dp_stub_gbps or args.dp_stub_lat_us) else
                                                    "%.1f us + 2(N-1)/N * bytes / %.0f GB/s per all-reduce group" % (args.dp_stub_lat_us, args.dp_stub_gbps))
        out["roofline"] = kernel_roofline(m, pool, default_wl) if args.dtype in ("bf16", "f16") else kernel_roofline_f32(m, pool)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(alpha, args.batch)
    else:
        if args.dtype in ("bf16", "f16"):
            kernel_roofline(m, pool, default_wl)      # keep ranks in lock-step through the extra (all-reducing) iteration
        else:
            kernel_roofline_f32(m, pool)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    m.ctx.close()
    if rank == 0:
        # RCCL writes a version banner through C stdio: flush it first so the JSON line is the last line
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.exit(main())
