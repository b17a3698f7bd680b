#!/usr/bin/env python3
"""Paired-seed precision study of the plain `rcgan` setting (gan_resnet.py:424-455,995-1005, README.md:75-80).

Every arm runs scripts/train_synthetic.py with the SAME seeds: a seed fixes the initial weights (create_variables), the data order and
the label-noise stream (np.random.seed(1000 + seed)) and the device random stream (z, dequantisation noise), so arm A seed s and arm B
seed s differ in the arithmetic only.  Score of a run = MEAN generated-label accuracy over the evaluations in the last 40 % of the run
(a single final point swings by 0.2 inside one run).  Output: per-arm mean +- SE and the paired differences against the reference arm.

The parent never touches the GPU: runs are child processes, --workers of them at a time on the one GPU (a B = 64 iteration leaves most
of the chip idle most of the time, so concurrent processes overlap).

  python scripts/precision_study.py --arms bf16,f16,f32 --seeds 12 --seeds_f32 10 --iters 5000 --workers 3 --out gpurun_out/r06_train_precision.json
An arm is `name=dtype[:ENV=VALUE[:ENV=VALUE...]]` or just a dtype, e.g. `bf16_nobatch=bf16:RCGAN_BATCH_CRITIC_FAKES=0`.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_arm(spec):
    parts = spec.split(":")
    name, eq, dtype = parts[0].partition("=")
    if not eq:
        dtype = name
    env = {}
    for kv in parts[1:]:
        k, _, v = kv.partition("=")
        env[k] = v
    return name, dtype, env


def score(curve, tail=0.4):
    """Mean accuracy over the evaluations whose iteration lies in the last `tail` of the run."""
    last = curve[-1]["iteration"]
    pts = [c["gen_label_acc"] for c in curve if c["iteration"] > last * (1 - tail) + 1e-9]
    return sum(pts) / len(pts), len(pts)


def mean_se(xs):
    n = len(xs)
    if n == 0:
        return None, None
    m = sum(xs) / n
    if n < 2:
        return m, None
    var = sum((x - m) ** 2 for x in xs) / (n - 1)
    return m, math.sqrt(var / n)


def summarise(results, arms, ref_arm, tail):
    per_arm, table = {}, {}
    for name, _, _ in arms:
        runs = {r["seed"]: r for r in results if r["arm"] == name}
        sc = {s: score(r["curve"], tail)[0] for s, r in runs.items()}
        fin = {s: r["curve"][-1]["gen_label_acc"] for s, r in runs.items()}
        table[name] = sc
        m, se = mean_se(list(sc.values()))
        mf, sef = mean_se(list(fin.values()))
        per_arm[name] = {"n_seeds": len(sc), "score_mean": m, "score_se": se, "final_point_mean": mf, "final_point_se": sef,
                         "scores_by_seed": {str(s): round(v, 4) for s, v in sorted(sc.items())},
                         "losses_finite": all(r["losses_finite"] for r in runs.values())}
    paired = {}
    for name, _, _ in arms:
        if name == ref_arm or ref_arm not in table:
            continue
        common = sorted(set(table[name]) & set(table[ref_arm]))
        d = [table[ref_arm][s] - table[name][s] for s in common]
        m, se = mean_se(d)
        paired["%s_minus_%s" % (ref_arm, name)] = {"n_pairs": len(common), "mean": m, "se": se,
                                                    "t": (m / se) if (se and se > 0) else None, "seeds": common}
    return per_arm, paired


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="bf16,f16,f32")
    ap.add_argument("--ref_arm", default="f32")
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--seeds_f32", type=int, default=None, help="seed count of the arms whose dtype is f32 (they cost 10x)")
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--eval_every", type=int, default=250)
    ap.add_argument("--alpha", type=float, default=0.6)
    ap.add_argument("--tail", type=float, default=0.4)
    ap.add_argument("--workers", type=int, default=3)
    ap.add_argument("--deadline_s", type=float, default=0, help="start no new run after this many seconds")
    ap.add_argument("--dir", default=os.path.join(ROOT, "gpurun_out", "r06_precision_runs"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_train_precision.json"))
    ap.add_argument("--summarise_only", action="store_true")
    a = ap.parse_args()
    arms = [parse_arm(s) for s in a.arms.split(",")]
    os.makedirs(a.dir, exist_ok=True)
    tasks = []
    for s in range(a.seed0, a.seed0 + a.seeds):
        for name, dtype, env in arms:
            if dtype == "f32" and a.seeds_f32 is not None and s >= a.seed0 + a.seeds_f32:
                continue
            tasks.append((name, dtype, env, s))
    # long runs first inside a seed so the tail of the schedule is short runs
    tasks.sort(key=lambda t: (t[3], 0 if t[1] == "f32" else 1))
    t0 = time.time()
    running = []

    def path_of(name, s):
        return os.path.join(a.dir, "%s_s%d.json" % (name, s))

    todo = [t for t in tasks if not os.path.exists(path_of(t[0], t[3]))]
    while (todo or running) and not a.summarise_only:
        running = [(p, t) for p, t in running if p.poll() is None]
        while todo and len(running) < a.workers and not (a.deadline_s and time.time() - t0 > a.deadline_s):
            name, dtype, env, s = todo.pop(0)
            e = dict(os.environ)
            e.update(env)
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_synthetic.py"), "--algorithm", "rcgan", "--dtype", dtype,
                   "--iters", str(a.iters), "--eval_every", str(a.eval_every), "--alpha", str(a.alpha), "--seed", str(s),
                   "--out", path_of(name, s)]
            log = open(os.path.join(a.dir, "%s_s%d.log" % (name, s)), "w")
            running.append((subprocess.Popen(cmd, env=e, stdout=log, stderr=subprocess.STDOUT), (name, s)))
            print("[%6.0f s] started %s seed %d" % (time.time() - t0, name, s), flush=True)
        if a.deadline_s and time.time() - t0 > a.deadline_s and not running:
            break
        time.sleep(2)
    results = []
    for name, dtype, env, s in tasks:
        p = path_of(name, s)
        if os.path.exists(p):
            with open(p) as f:
                r = json.loads(f.readline())
            r["arm"] = name
            r["env"] = env
            results.append(r)
    per_arm, paired = summarise(results, arms, a.ref_arm, a.tail)
    out = {"what": "paired-seed precision study, plain rcgan on the class-pattern stand-in (scripts/train_synthetic.py); score = mean "
                   "generated-label accuracy over the evaluations in the last %.0f %% of the run" % (a.tail * 100),
           "alpha": a.alpha, "iterations": a.iters, "eval_every": a.eval_every, "workers": a.workers,
           "arms": {n: {"dtype": d, "env": e} for n, d, e in arms}, "reference_arm": a.ref_arm, "per_arm": per_arm,
           "paired_differences": paired, "wall_s": round(time.time() - t0, 1),
           "curves": {"%s_s%d" % (r["arm"], r["seed"]): [[c["iteration"], c["gen_label_acc"]] for c in r["curve"]] for r in results}}
    with open(a.out, "w") as f:
        json.dump(out, f)
        f.write("\n")
    print(json.dumps({"per_arm": {k: {x: v[x] for x in ("n_seeds", "score_mean", "score_se", "final_point_mean")} for k, v in per_arm.items()},
                      "paired": paired}, indent=1))


if __name__ == "__main__":
    main()
