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
  roofline     dominant kernel (conv_mfma_p8_kernel: fwd + dgrad of the 256-channel convs, 256x256 tiles): algorithmic flops
               of its launches in one iteration / their summed duration measured with HIP events on the
               launch stream, against the dense bf16 MFMA peak.
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
POOL = 8                      # synthetic batches resident on the device
BATCH_CRITIC_FAKES = os.environ.get("RCGAN_BATCH_CRITIC_FAKES", "1") == "1"


def build_pool(m, rank, alpha):
    """Synthetic label / image streams uploaded once; returns device tensors to cycle through."""
    from rcgan_amd import data as D
    B = m.B
    rs = np.random.RandomState(1234 + rank)
    n = POOL * B
    images = rs.randint(0, 256, size=(n, 3072))
    clean = rs.randint(10, size=n)
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
    for _ in range(N_CRITIC):
        feed_d(m, pool, dcount[0])
        dcount[0] += 1
        m.d_step(iteration=it)


def kernel_roofline(m, pool, default_workload=True):
    """One eager (un-captured) iteration with every conv_mfma_p8_kernel launch (the dominant kernel: forward and data
    gradient of the 256-channel 3x3 / 1x1 convolutions, 256 x 256 tiles) bracketed by HIP events on the launch stream."""
    from rcgan_amd import _lib as L
    ctx = m.ctx
    saved = m.use_graphs
    m.use_graphs = False
    ctx.check(ctx.lib.rcgan_prof_begin(ctx.h, 4))       # RCGAN_PROF_CONV_P8
    iteration(m, pool, 1, [0])
    n, ms, fl = C.c_int(0), C.c_double(0), C.c_double(0)
    ctx.check(ctx.lib.rcgan_prof_end(ctx.h, C.byref(n), C.byref(ms), C.byref(fl)))
    fx = C.c_double(0)
    ctx.check(ctx.lib.rcgan_prof_executed_flops(ctx.h, C.byref(fx)))
    m.use_graphs = saved
    if n.value == 0 or ms.value <= 0:
        return None
    achieved = fl.value / (ms.value * 1e-3) / 1e12
    # HBM bytes per launch of this kernel: measured in separate rocprofv3 --pmc passes of this same command (FETCH_SIZE and
    # WRITE_SIZE cannot share a pass; corrections as MI355X_MICROARCH.md prescribes) and committed with the profile
    traffic = None
    import glob
    tfs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic_conv_p8.json")))
    if default_workload and tfs:                         # the counters were collected on the default workload only; newest round
        with open(tfs[-1]) as f:
            traffic = float(json.load(f)["traffic_bytes_per_launch"])
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "kernel": "conv_mfma_p8_kernel", "launches_per_iteration": n.value,
            "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
            "flops_per_launch_avg": fl.value / n.value,
            # algorithmic = the reference's formulation; the upsample-3x3 layers run in their sub-pixel form (four 2x2
            # convolutions with summed filters): the matrix cores execute 4/9 of those layers' multiply-adds
            "executed_tflops": round(fx.value / (ms.value * 1e-3) / 1e12, 2), "executed_frac": round(fx.value / (ms.value * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}


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
    usable cores (scheduler affinity capped by the cgroup CPU quota) at the benchmark's own per-GPU batch.  Bounded sample: one warm-up and one timed D step + G step; the iteration is
    5 D steps + 1 G step, so images/sec = 5*B / (5*t_D + t_G).  (The single-process numpy oracle, the parity checker, runs
    the same step at B=16 in ~2.1 s / ~4.1 s: ~5.4 images/sec.)"""
    from oracle import cifar as oc
    from oracle.torch_port import CifarTorchTrainer
    B = batch
    rs = np.random.RandomState(0)
    P, U = oc.init_params(0, "rcgan")
    Cm = oc.c_alpha(alpha)
    cores = effective_cores()
    torch.set_num_threads(cores)

    def batches():
        lab = rs.randint(10, size=B)
        db = dict(real=oc.preprocess_real(rs.randint(0, 256, size=(B, 3072)), rs.uniform(0, 1 / 128., size=(B, 3072))).astype(np.float32),
                  labels=lab, labels_random=rs.randint(10, size=B), labels_biased=rs.randint(10, size=B),
                  inv_weights=np.linalg.inv(Cm)[lab], z=rs.randn(B, 128))
        gb = dict(labels_random_G=rs.randint(10, size=2 * B), labels_biased_G=rs.randint(10, size=2 * B), z=rs.randn(2 * B, 128))
        return db, gb
    tr = CifarTorchTrainer(P, U, dict(algorithm="rcgan", C=Cm), lr=2e-4)
    db, gb = batches()
    tr.d_step(db)                      # warm-up (thread pool, oneDNN primitive caches)
    tr.g_step(gb)
    db, gb = batches()
    t0 = time.time()
    tr.d_step(db)
    t1 = time.time()
    tr.g_step(gb)
    t2 = time.time()
    td, tg = t1 - t0, t2 - t1
    return {"value": round(5 * B / (5 * td + tg), 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "PyTorch-CPU restatement (oracle/torch_port.py), CIFAR RCGAN B=%d fp32, %d threads: 1 D step (%.2fs) + 1 G step (%.2fs) "
                      "timed after one warm-up of each; iteration = 5 D + 1 G" % (B, cores, td, tg)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU critic batch")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--algorithm", default="rcgan")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lr", type=float, default=2e-4, help="Adam learning rate (reference: 2e-4, gan_resnet.py:--lr)")
    args = ap.parse_args()
    default_wl = args.batch == 64 and args.dtype == "bf16" and args.algorithm == "rcgan"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("RCGAN_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import rcgan_amd  # noqa: F401
    from rcgan_amd.cifar import CifarRCGAN

    alpha = 0.6
    m = CifarRCGAN(algorithm=args.algorithm, alpha=alpha, batch_size=args.batch, dtype=args.dtype, seed=0, lr=args.lr,
                   device=local, use_graphs=not args.no_graphs, device_rng=True, world_size=world, rank=rank)
    if force_dist:
        m.world = 2          # take the all-reduce branch; grad_scale 1/2 cancels against the doubled "sum" below
        from rcgan_amd import dp as _dp
        _orig = _dp.allreduce_sum_
        def _twice(flat, stream=None):
            _orig(flat, stream)
            with torch.cuda.stream(stream):
                flat.mul_(2.0)
            return flat
        _dp.allreduce_sum_ = _twice
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
                          "critic_generator_forwards": ("one pass over N_CRITIC x B samples, batch-norm statistics per critic step"
                                                        if BATCH_CRITIC_FAKES else "inside every critic step"),
                          "iteration_tflops_algorithmic": round(60.858 * args.batch * world / 1e3, 3),
                          "sustained_tflops": round(60.858 * args.batch * world / 1e3 / (dt / args.steps), 2),
                          "losses_finite": bool(ok), "d_loss": round(d_loss, 4), "g_loss": round(g_loss, 4)}}
        out["roofline"] = kernel_roofline(m, pool, default_wl) if args.dtype in ("bf16", "f16") else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(alpha, args.batch)
    else:
        if args.dtype in ("bf16", "f16"):
            kernel_roofline(m, pool, default_wl)      # keep ranks in lock-step through the extra (all-reducing) iteration
    if world > 1 or force_dist:
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
    main()
