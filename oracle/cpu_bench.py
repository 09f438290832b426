"""CPU baseline of bench.py (the `cpu_baseline*` fields): the oracle stepping the BASELINE configs[1] workload on host cores.

TEST INFRASTRUCTURE: this times the CHECKER (oracle/navsim_oracle.c, a scalar C restatement of the reference's Env.step with the
authored kinematic sim + ray-cast), it is never part of the product path.  Run as its own process by bench.py so that the
worker pool forks from an interpreter that has never touched the HIP runtime:

    python -m oracle.cpu_bench --procs P --envs N --budget S      -> one JSON line

P worker processes each own a contiguous shard of the N envs (env ids keyed like the GPU shards) and step it with random
actions until the time budget is spent; value = total env-steps / wall time.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _worker(args):
    lo, n, budget, t_start = args
    from navbot_ppo_amd import maps
    from oracle import navsim_oracle as O
    sim = O.OracleSim(n, max_episode_steps=500, auto_reset=True, seed=0, env_id_base=lo)
    sim.set_map(maps.stage_1())
    sim.reset()
    rng = np.random.default_rng(lo)
    a = np.stack([rng.uniform(0, 1, n), rng.uniform(-1, 1, n)], 1).astype(np.float32)
    sim.step(a)
    while time.time() < t_start:      # all workers start together
        pass
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget:
        sim.step(a)
        steps += 1
    return steps * n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--budget", type=float, default=8.0)
    a = ap.parse_args()
    from oracle import navsim_oracle as O
    O.build()
    procs = max(1, min(a.procs, a.envs))
    per = a.envs // procs
    shards = [(k * per, per if k < procs - 1 else a.envs - per * (procs - 1)) for k in range(procs)]
    t_start = time.time() + (1.0 + 0.01 * procs if procs > 1 else 0.0)
    jobs = [(lo, n, a.budget, t_start) for lo, n in shards]
    if procs == 1:
        res = [_worker(jobs[0])]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_worker, jobs, chunksize=1)
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    print(json.dumps(dict(value=round(total / wall, 1), unit="env-steps/s", cores=procs, kind="port",
                          sample=f"{total} env-steps of the {a.envs}-env workload (stage_1, 10 beams, random actions, env step only) "
                                 f"in {procs} process(es) x {per} envs, {wall:.1f} s",
                          host_cpus=os.cpu_count())))


if __name__ == "__main__":
    main()
