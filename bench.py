#!/usr/bin/env python3
"""bench.py — ICP iterations/sec on 1M-point synthetic cloud pairs (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One *step* = one full pass of the hot path over one cloud pair: ``ma_icp`` with a fixed 50
iterations (tol = 0, no early exit) from the identity start, i.e. BASELINE.json
``configs[1]`` ("1M-point synthetic cloud pair, 50 ICP iterations, 1xMI355X").  Both clouds
and the gold grid are resident in HBM before the timed region (grid build is reported
separately in ``config.grid_build_ms``).  ``value`` = iterations of all ranks / wall time.

N > 1 (launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N``): pair-per-GPU, the way the path shards — every rank aligns its
own 1M-point pair (seed = rank) with no collective inside the loop; weak scaling.  The
timed region is bracketed by barrier + synchronize and the MAX over ranks is taken (RCCL).
``--mode sharded`` instead spreads ONE pair over the ranks with a per-iteration RCCL
all-reduce of the 17-double accumulator (strong scaling; not the driver's default).

Extra objects on the JSON line:
  roofline     k_nn_accum (the NN-correspondence kernel): algorithmic bytes per launch
                / average launch duration measured here with hipEvents on the
               library's stream, against the 8 TB/s HBM peak
  cpu_baseline the CPU oracle (NumPy fp64 + scikit-learn KD-tree, all host cores) timed on
               a bounded sample of the same workload, rank 0 at N = 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ITERS = 50
GATE = 0.10
N_POINTS = 1_000_000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--cell", type=float, default=0.0, help="grid cell edge (m); 0 = density-driven")
    ap.add_argument("--mode", choices=["pairs", "sharded"], default="pairs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    # rehearsal on a one-GPU box only: several ranks share device 0 and rendezvous over gloo
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--force-device", type=int, default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def algorithmic_bytes(trace, n_src, n_blocks):
    """bytes one k_nn_accum launch must move, from the exact work counters.

    16 B per source point (float4) + 8 B per point for the remembered match (4 B read +
    4 B write) + 4 B per grid-table word read + 12 B per candidate gold point examined
    (coordinates only, the remembered match included: SURVEY.md §8d's 12*C) + 12 B per
    matched point (re-read of the winner) + 19 fp64 partials per workgroup.  Averaged over
    the launches of one step."""
    words, cands, matched = trace[:, 18], trace[:, 19], trace[:, 0]
    per_launch = 24.0 * n_src + 4.0 * words + 12.0 * cands + 12.0 * matched + 19 * 8.0 * n_blocks
    return float(per_launch.mean()), float(words.mean() / n_src), float(cands.mean() / n_src)


def cpu_baseline(gold, comp, iters):
    """Oracle (NumPy + scikit-learn KD-tree, n_jobs=-1) on a bounded sample: `iters` iterations."""
    from oracle import icp_oracle as orc

    cores = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    index = orc.GoldIndex(gold, n_jobs=-1)
    t_build = time.perf_counter() - t0
    R, t = np.eye(3), np.zeros(3)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        _, _, _, R, t, _ = orc.icp_step(index, comp, R, t, GATE)
        times.append(time.perf_counter() - t0)
    per_it = float(np.median(times))
    return {"value": 1.0 / per_it, "unit": "ICP iterations/s", "cores": cores, "kind": "port",
            "sample": f"first {iters} of the 50 iterations of the same {len(comp)}-point pair, median per-iteration "
                      f"time {per_it:.3f} s; KD-tree build {t_build:.2f} s excluded (GPU grid build is excluded too)",
            "os_cpu_count": os.cpu_count()}, (R, t)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node "
                             f"{args.gpus} --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import mapalign
    from mapalign import dist as mdist
    from mapalign import synthetic as syn

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":      # RCCL
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs: resident in HBM before the timed region --------------------------------
    sharded = args.mode == "sharded" and world > 1
    seed = 0 if sharded else rank
    gold, comp, R0, t0 = syn.make_pair(seed, args.points)
    eng = mapalign.MapAligner(local_rank)
    grid = eng.set_gold(gold, args.cell)
    if sharded:
        lo, hi = mdist.shard_bounds(len(comp), world, rank)
        eng.set_source(comp[lo:hi])
        he = mdist.HipShardEngine(eng, GATE, 0.0)
        he.bind_current_stream()
        acc = he.new_acc()

        def step():
            return mdist.sharded_icp(he, acc, max_iter=ITERS)
    else:
        eng.set_source(comp)

        def step():
            return eng.icp(ITERS, 0.0, GATE)

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    pairs_per_step = 1 if sharded else world
    value = pairs_per_step * args.steps * ITERS / elapsed

    # ---- result check + fleet stats (exercises RCCL at N > 1 outside the timed region) --
    T = out[0] if sharded else out.T
    err_R = float(np.linalg.norm(T[:3, :3] - R0))
    err_t = float(np.linalg.norm(T[:3, 3] - t0))
    fleet = torch.tensor([err_R, err_t], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(fleet, op=dist.ReduceOp.MAX)

    line = None
    if rank == 0:
        # ---- roofline of the dominant kernel, measured live (untimed extra passes) -----
        roof = None
        if not sharded:
            res_t = eng.icp(ITERS, 0.0, GATE, time_kernels=True)
            kt = eng.kernel_times()
            res_c = eng.icp(ITERS, 0.0, GATE, count_work=True, with_trace=True)
            n_blocks = (len(comp) + 255) // 256
            b_launch, rows_pp, cands_pp = algorithmic_bytes(res_c.trace, len(comp), n_blocks)
            t_launch_s = kt["nn_ms"] / max(kt["nn_launches"], 1) * 1e-3
            achieved = b_launch / t_launch_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath) and args.points == N_POINTS and args.cell == 0.0:   # measured for this workload only
                try:
                    traffic = json.load(open(tpath)).get("k_nn_accum_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": "k_nn_accum", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                    "bytes_per_launch": b_launch, "launch_us": t_launch_s * 1e6,
                    "table_words_per_point": rows_pp, "candidates_per_point": cands_pp,
                    "compulsory_bytes_per_launch": 24.0 * len(comp) + 16.0 * len(gold),
                    "compulsory_frac": (24.0 * len(comp) + 16.0 * len(gold)) / t_launch_s / 1e9 / HBM_PEAK_GBPS,
                    "solve_kernel_us": kt["solve_ms"] / max(kt["solve_launches"], 1) * 1e3,
                    "timed_loop_it_per_s": ITERS / (res_t.gpu_ms * 1e-3)}
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, (Rc, tc) = cpu_baseline(gold, comp, args.cpu_iters)
            # parity on the same inputs at the same iteration count (the CPU sample's length)
            res_p = eng.icp(args.cpu_iters, 0.0, GATE)
            parity = {"iterations": args.cpu_iters,
                      "R_fro_vs_cpu": float(np.linalg.norm(res_p.R - Rc)),
                      "t_m_vs_cpu": float(np.linalg.norm(res_p.t - tc))}
        line = {
            "metric": "ICP iterations/sec on 1M-pt cloud pair; final (R,t) error vs CPU ref",
            "value": value, "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f32 (f64 accumulate/solve)",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: 1M-point synthetic cloud pair, 50 ICP iterations, 1xMI355X"
                                    if args.points == N_POINTS else
                                    f"{args.points}-point synthetic cloud pair (configs[1] at another size), 50 ICP iterations")
                       + ("" if world == 1 else f" - one pair per GPU x{world}" if not sharded
                          else f" - one pair sharded over {world} GPUs, RCCL all-reduce of 17 doubles/iteration"),
                       "points_gold": len(gold), "points_comparison": len(comp), "iterations_per_step": ITERS,
                       "max_corr_dist_m": GATE, "grid_cell_m": grid["cell"], "grid_dims": list(grid["dims"]),
                       "grid_build_ms": grid["build_ms"], "parallelism": "pair-per-gpu" if not sharded else "sharded-source"},
            "final_error_vs_truth": {"R_fro_max_over_ranks": float(fleet[0].item()), "t_m_max_over_ranks": float(fleet[1].item())},
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_cpu": parity,
        }
        if cpu:
            line["speedup_vs_cpu_baseline"] = value / cpu["value"]
    eng.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
