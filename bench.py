#!/usr/bin/env python3
"""bench.py — ICP iterations/sec on 1M-point synthetic cloud pairs (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One *step* = one full pass of the hot path over one cloud pair: ``ma_icp`` with a fixed 50
iterations (tol = 0, no early exit) from the identity start, i.e. BASELINE.json
``configs[1]`` ("1M-point synthetic cloud pair, 50 ICP iterations, 1xMI355X").  Both clouds
and the gold grid are resident in HBM before the timed region (grid build and the
PCIe-inclusive end-to-end time are reported separately in ``config``).  ``value`` =
iterations of all ranks / wall time.

N > 1: pair-per-GPU, the way the path shards — every rank aligns its own
1M-point pair (seed = rank) with no collective inside the loop; weak scaling.  The timed
region is bracketed by barrier + synchronize and the MAX over ranks is taken (RCCL).  Either
launch it as ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` or
just as ``python bench.py --gpus N``: without WORLD_SIZE in the environment the script starts
that launcher itself as a CHILD process (before anything touches a GPU), relays rank 0's JSON
line and exits with the child's return code.
The same N > 1 run then times a second region in the same processes and reports it as
``"sharded"`` beside the headline: ONE 1M-point pair spread over the ranks (gold map
replicated, comparison cloud block-partitioned), the per-iteration all-reduce of the 17-double
accumulator INSIDE libmapalign (``ma_icp_sharded``: ncclAllReduce on the library's stream, loop
in C; strong scaling) — the second curve of SURVEY.md §8e.  ``--mode sharded`` makes that
region the headline instead.

Extra objects on the JSON line (all measured in this run, outside the timed region):
  roofline       k_nn_accum (the NN-correspondence kernel): algorithmic bytes per launch
                  / average launch duration measured here with hipEvents on the
                 library's stream, against the 8 TB/s HBM peak
  cpu_baseline   the CPU oracle (NumPy fp64 + scikit-learn KD-tree, all host cores) timed on
                 a bounded sample of the same workload, rank 0 at N = 1 only
  parity_vs_cpu  final (R,t) of the GPU path against the oracle after the SAME 50 iterations
                 of the same 1M-point pair (north_star tolerance: 1e-4 Frobenius / 1 mm)
  configs        c3 (1280x960 depth frame -> fused unprojection -> 50 iterations against the
                 1M gold map), c4 (batch-32 detector inference on PyTorch-ROCm + one-call box
                 lift), c5_per_gpu (a 2M-point pair, the per-GPU size of configs[4]), stray_points (the
                 headline pair with stray points far from the scene in both clouds),
                 concurrent_pairs (1 / 2 / 4 independent alignments iterating at once on this GPU)
  config.end_to_end_*  the drop-in call itself from host arrays (first / warm / fp64 / device-resident
                 inputs / resident gold map / a stream of pairs through align_many) and the
                 reference-shaped Alignment(...).compare() on two .ply files

``--pairs-per-gpu K`` (default 1 = the headline configuration) runs the timed region with K independent
pairs iterating at once on every GPU; stdout carries exactly ONE line (everything else goes to stderr).
"""
import argparse
import hashlib
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
# what the loop's kernels and the plan of its launches are compiled from (profiles/pmc_traffic.json is "fresh" for this hash)
KERNEL_SOURCES = tuple("3d-localisation-and-mapping_amd/csrc/" + f for f in (
    "ma_internal.h", "kernels_search.hip", "kernels_solve.hip", "host_loop.hip", "host_math.h", "kabsch.h", "list_planner.h"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--cell", type=float, default=0.0, help="grid cell edge (m); 0 = density-driven")
    ap.add_argument("--mode", choices=["pairs", "sharded"], default="pairs")
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="alignments iterating at once on each GPU (separate contexts / streams / host threads); the headline "
                         "metric is quoted at 1 — more is the fleet-throughput figure of configs.concurrent_pairs as a whole run")
    ap.add_argument("--sharded-timeout", type=float, default=180.0,
                    help="N > 1: seconds the sharded-source leg (library RCCL) may take before the line is printed without it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=ITERS, help="oracle iterations for parity_vs_cpu (default: all 50)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the configs.c3 / c4 / c5_per_gpu objects")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed region and the roofline passes only (no end-to-end / CPU / extra legs, whose concurrent "
                         "alignments would mix overlapped launches into a profiler's per-kernel averages): the run "
                         "tools/profile_round.sh puts under rocprofv3 --kernel-trace --stats")
    # rehearsal on a one-GPU box only: several ranks share device 0 and rendezvous over gloo
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--force-device", type=int, default=None, help=argparse.SUPPRESS)
    # rehearsal of the N > 1 run's second region on ONE GPU: a 1-rank RCCL process group and communicator
    ap.add_argument("--rehearse-sharded", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_extra_configs = True
    return args


def launcher_command(args_list, n_gpus, port):
    """The driver's own spelling of an N-GPU run (one rank per GPU over RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(args_list)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args_list, n_gpus, command=None):
    """``python bench.py --gpus N`` from a plain shell: run the N ranks as a child process (never
    an exec: this process may not replace itself once anything has touched the GPU, and a child
    keeps the rule simple), relay the ONE JSON line of rank 0 and return the child's code."""
    import subprocess

    cmd = command or launcher_command(args_list, n_gpus, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL across processes)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in proc.stdout.decode("utf-8", "replace").splitlines():
        if raw.startswith("{") and '"metric"' in raw:
            line = raw
        else:
            print(raw, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif proc.returncode == 0:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def kernel_source_sha():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(trace, n_src, n_blocks, plan=None, kind=0):
    """bytes one k_nn_accum launch must move, from the exact work counters.

    Search launches (kinds 0 and 1): 16 B per source point (float4) + 8 B per point for the
    remembered match (4 B read + 4 B write) + 4 B per grid-table word read + 12 B per candidate
    gold point examined (coordinates only, the remembered match included: SURVEY.md §8d's 12*C)
    + 12 B per matched point (re-read of the winner) + 19 fp64 partials per workgroup; what a
    list-writing launch (kind 1) stores is left out.
    A list launch (kind 2: DESIGN.md 5 "List launches", round-4 layout: positions only) reads 16 B source
    point + 4 B radius = 20 B per point, and per counted list word (4 per point, 8 where the list holds
    more than four entries: the kernel's word counter in these launches) the 4-byte position and the 12 B
    of the gold point it names: 16 B per word.  No remembered match is read or written and the winner is
    not re-read (its coordinates travel with the minimum).
    Averaged over the launches of ``kind`` (``plan`` = engine.list_plan(): what each iteration
    launched; None: every launch searched)."""
    if plan is not None:
        sel = np.asarray(plan[:len(trace)]) == kind
        trace = trace[:len(sel)][sel]
    if len(trace) == 0:
        return 0.0, 0.0, 0.0, 0.0
    words, cands, matched = trace[:, 18], trace[:, 19], trace[:, 0]
    per_launch = 24.0 * n_src + 4.0 * words + 12.0 * cands + 12.0 * matched + 19 * 8.0 * n_blocks
    if kind == 2:
        per_launch = 20.0 * n_src + 16.0 * words + 19 * 8.0 * n_blocks
    # SURVEY.md 8d's literal formula — 12 N (source xyz) + 4 per table word + 12 C — without the storage actually
    # used (float4 source points), the remembered-match words and the winner re-read: the strictest reading
    strict = 12.0 * n_src + 4.0 * words + 12.0 * cands
    return float(per_launch.mean()), float(words.mean() / n_src), float(cands.mean() / n_src), float(strict.mean())


def survey_cell_leg(gold, comp, device):
    """SURVEY.md 8d, read literally: grid cell edge = the gate h (0.10 m) instead of the density-driven edge the
    headline runs with.  Own context, same pair, same 50 iterations; its own work counters, launch time and fractions."""
    import mapalign
    with mapalign.MapAligner(device) as e:
        grid = e.set_gold(gold, GATE)
        e.set_source(comp)
        e.icp(ITERS, 0.0, GATE)
        ms = min(e.icp(ITERS, 0.0, GATE).gpu_ms for _ in range(3))
        e.icp(ITERS, 0.0, GATE, time_kernels=True)
        kt = e.kernel_times()
        res_c = e.icp(ITERS, 0.0, GATE, count_work=True, with_trace=True)
        n_blocks, _ = e.search_geometry()
        b_launch, rows_pp, cands_pp, b_strict = algorithmic_bytes(res_c.trace, len(comp), n_blocks)
        t_launch_s = kt["nn_ms"] / max(kt["nn_launches"], 1) * 1e-3
        return {"grid_cell_m": grid["cell"], "grid_dims": list(grid["dims"]), "loop_ms": ms,
                "it_per_s": ITERS / (ms * 1e-3), "launch_us": t_launch_s * 1e6,
                "table_words_per_point": rows_pp, "candidates_per_point": cands_pp,
                "bytes_per_launch": b_launch, "bytes_per_launch_survey_8d": b_strict,
                "frac": b_launch / t_launch_s / 1e9 / HBM_PEAK_GBPS,
                "frac_survey_8d": b_strict / t_launch_s / 1e9 / HBM_PEAK_GBPS,
                "note": "same kernels with the cell edge SURVEY 8d prescribes: more candidates per query, so MORE algorithmic "
                        "bytes per launch and a higher fraction, but fewer iterations/s - why the headline uses the density-driven edge"}


def cpu_baseline_and_parity(eng, gold, comp, iters):
    """Oracle (the build's own fp64 restatement: NumPy + scikit-learn KD-tree, n_jobs=-1): `iters` iterations of the SAME pair.

    cpu_baseline.value comes from the first 10 iterations, on the scikit-learn backend north_star names (the bounded
    sample: ~20 s on the 256-core GPU box).  The oracle then runs on to `iters` — from the same state, on its cKDTree
    backend (the same nearest neighbours: tests/test_oracle.py; 4-7 times faster) — so that the final (R,t) is compared
    with the GPU path at EQUAL iteration count, all 50, list launches included, on every host; a budget of 150 s of oracle
    wall remains as a backstop (never below 32 iterations), and the line says at which iteration both sides stopped."""
    from oracle import icp_oracle as orc

    cores = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    index = orc.GoldIndex(gold, n_jobs=-1)
    t_build = time.perf_counter() - t0
    fast = None
    R, t = np.eye(3), np.zeros(3)
    times, inliers = [], []
    k = 0
    n_timed = 10
    while k < iters:
        if k == n_timed:
            fast = orc.GoldIndex(gold, n_jobs=-1, backend="ckdtree")
        t0 = time.perf_counter()
        acc, _, _, R, t, ok = orc.icp_step(index if k < n_timed else fast, comp, R, t, GATE)
        times.append(time.perf_counter() - t0)
        inliers.append(float(acc[0]))
        k += 1
        if k == n_timed + 3:
            iters = min(iters, max(32, n_timed + int((150.0 - sum(times)) / max(np.mean(times[n_timed:]), 1e-3))))
    sample = times[:n_timed]
    per_it = float(np.median(sample))
    cpu = {"value": 1.0 / per_it, "unit": "ICP iterations/s", "cores": cores, "kind": "restatement",
           "kind_note": "the build's own fp64 oracle (oracle/icp_oracle.py: textbook point-to-point ICP, NumPy + scikit-learn KD-tree); "
                        "the reference ships no implementation of this path to time (SURVEY.md 0 F1)",
           "sample": f"first {len(sample)} of the 50 iterations of the same {len(comp)}-point pair, median per-iteration "
                     f"time {per_it:.3f} s ({sum(sample):.1f} s of CPU wall); KD-tree build {t_build:.2f} s excluded "
                     f"(GPU grid build is excluded too)",
           "os_cpu_count": os.cpu_count(), "oracle_wall_s_all_iterations": float(sum(times))}
    res = eng.icp(iters, 0.0, GATE, with_trace=True)
    parity = {"iterations": iters,
              "oracle_backends": f"scikit-learn kd_tree for iterations 0-{n_timed - 1} (the timed sample), scipy cKDTree from there "
                                 f"(same nearest neighbours; {float(np.median(times[n_timed:])) if len(times) > n_timed else 0.0:.2f} s per iteration)",
              "launches_compared": "".join("SWL"[int(v)] for v in eng.list_plan()[:iters]) + " (S search, W search + write lists, L read lists)",
              "R_fro_vs_cpu": float(np.linalg.norm(res.R - R)),
              "t_m_vs_cpu": float(np.linalg.norm(res.t - t)),
              "max_inlier_count_difference": float(np.abs(res.trace[:, 0] - np.array(inliers)).max()),
              "tolerance": {"R_fro": 1e-4, "t_m": 1e-3}}
    parity["within_tolerance"] = bool(parity["R_fro_vs_cpu"] <= 1e-4 and parity["t_m_vs_cpu"] <= 1e-3)
    if iters < ITERS:
        parity["note"] = (f"SHORTENED from {ITERS} iterations: this host's cores need {np.mean(times[:3]):.1f} s per oracle iteration; "
                          f"both sides ran {iters} (tests/test_gpu_round2.py compares seed 0 after all {ITERS} whatever the host)")
    return cpu, parity


def end_to_end(gold, comp, device):
    """The drop-in call itself: mapalign.align(gold, comp) from host arrays — H2D of both
    clouds, gold grid build, source sort, 50 iterations, read-back.  First call (context
    creation + device allocations) and warm calls (cached context: no allocation)."""
    import mapalign

    mapalign.release_cached_engines()
    t0 = time.perf_counter()
    mapalign.align(gold, comp, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
    first = (time.perf_counter() - t0) * 1e3
    warm, allocs = [], []
    for _ in range(7):
        _, _, info = mapalign.align(gold, comp, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device, return_info=True)
        allocs.append(info["mem"]["device_allocs"])      # (return_info reads the trace back: not part of the plain call timed next)
        t0 = time.perf_counter()
        mapalign.align(gold, comp, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
        warm.append((time.perf_counter() - t0) * 1e3)
    with mapalign.GoldMap(gold, device=device) as gm:
        mapalign.align(gm, comp, max_iter=ITERS, tol=0.0, max_corr_dist=GATE)
        t0 = time.perf_counter()
        for _ in range(3):
            mapalign.align(gm, comp, max_iter=ITERS, tol=0.0, max_corr_dist=GATE)
        resident = (time.perf_counter() - t0) / 3 * 1e3
    # the same call with both clouds already in HBM (torch tensors): no PCIe in the call
    import torch
    g_d, c_d = torch.from_numpy(gold).to(f"cuda:{device}"), torch.from_numpy(comp).to(f"cuda:{device}")
    mapalign.align(g_d, c_d, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
    dev_in = []
    for _ in range(5):
        t0 = time.perf_counter()
        mapalign.align(g_d, c_d, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
        dev_in.append((time.perf_counter() - t0) * 1e3)
    del g_d, c_d
    # ... and with fp64 host arrays (what Open3D hands over): converted on the device
    g64, c64 = gold.astype(np.float64), comp.astype(np.float64)
    mapalign.align(g64, c64, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
    f64_in = []
    for _ in range(5):
        t0 = time.perf_counter()
        mapalign.align(g64, c64, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
        f64_in.append((time.perf_counter() - t0) * 1e3)
    # a stream of pairs on this GPU (mapalign.align_many): three worker threads, one context each, take pairs as they finish
    stream = [(gold.copy(), comp) for _ in range(24)]     # distinct gold objects: every pair builds its own grid
    mapalign.align_many(stream[:3], max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
    piped = []
    for _ in range(3):                                   # (a run of 24 pairs varies by +-10 % with the workers' interleaving)
        t0 = time.perf_counter()
        mapalign.align_many(stream, max_iter=ITERS, tol=0.0, max_corr_dist=GATE, device=device)
        piped.append((time.perf_counter() - t0) / len(stream) * 1e3)
    piped = float(np.median(piped))
    # the reference's call shape (task_def.py:175-187): Alignment(...).compare() on two .ply files, the scan far off
    import tempfile
    from mapalign import ply, synthetic as syn
    with tempfile.TemporaryDirectory() as d:
        far = syn.make_pair(0, len(comp), angle_deg=150.0, t0=(7.0, -1.0, 0.5))[1]
        gp, sp = os.path.join(d, "gold.ply"), os.path.join(d, "scan.ply")
        ply.write_ply(gp, gold)
        ply.write_ply(sp, far)
        cmp_ms = []
        for _ in range(4):
            t0 = time.perf_counter()
            mapalign.Alignment(base_map_filepath=gp, comparison_map_filepath=sp, device=device).compare(d)
            cmp_ms.append((time.perf_counter() - t0) * 1e3)
    mapalign.release_cached_engines()
    return {"end_to_end_ms": float(np.median(warm)), "end_to_end_ms_first_call": first,
            "compare_call_ms_two_1m_ply_files": float(np.median(cmp_ms[1:])),
            "end_to_end_ms_per_pair_pipelined_stream": piped, "end_to_end_it_per_s_pipelined_stream": ITERS / (piped * 1e-3),
            "end_to_end_ms_gold_resident": resident, "end_to_end_ms_device_inputs": float(np.median(dev_in)),
            "end_to_end_ms_fp64_host_inputs": float(np.median(f64_in)),
            "device_allocs_in_warm_calls": int(allocs[-1] - allocs[0]),
            "end_to_end_it_per_s": ITERS / (float(np.median(warm)) * 1e-3)}


def config_c3(eng, gold, cpu_iters=ITERS):
    """configs[2]: 1280x960 depth image -> fused unprojection -> 50 iterations vs the 1M gold map."""
    from mapalign import synthetic as syn

    depth, K, T = syn.make_depth_frame(0)
    dT = np.eye(4)
    dT[:3, :3] = syn.rotation_about_axis([0.2, 1.0, 0.3], np.deg2rad(1.0))
    dT[:3, 3] = [0.03, -0.02, 0.025]
    Tp = dT @ T                      # the pose the mapper believes: 1 degree / 4.4 cm off
    eng.set_gold(gold)
    walls, kern, icp_ms, walls_async, frame_to_T = [], [], [], [], []
    for _ in range(5):
        t0 = time.perf_counter()
        n = eng.set_source_from_depth(depth, K, Tp, 1000.0)          # with the count of valid pixels: waits for the device
        walls.append((time.perf_counter() - t0) * 1e3)
        kern.append(eng.last_device_ms())
        r = eng.icp(ITERS, 0.0, GATE)
        icp_ms.append(r.gpu_ms)
        # the caller that wants the transform, not the count: upload + unprojection enqueued, the alignment right behind
        t0 = time.perf_counter()
        eng.set_source_from_depth(depth, K, Tp, 1000.0, count=False)
        walls_async.append((time.perf_counter() - t0) * 1e3)
        eng.icp(ITERS, 0.0, GATE)
        frame_to_T.append((time.perf_counter() - t0) * 1e3)
    res_c = eng.icp(ITERS, 0.0, GATE, count_work=True, with_trace=True)
    lists = dict(eng.list_stats(), plan="".join("SWL"[int(v)] for v in eng.list_plan()))
    parity = None
    if cpu_iters > 0:
        # configs[2] against the oracle at EQUAL iteration count: the oracle's own unprojection of the frame (fp64, valid
        # pixels in row-major order), its 50 iterations (cKDTree backend), against unprojection + loop on the device
        from oracle import icp_oracle as orc
        opts, mask = orc.unproject_oracle(depth, K, Tp, 1000.0)
        t_0 = time.perf_counter()
        index = orc.GoldIndex(gold, n_jobs=-1, backend="ckdtree")
        R, t = np.eye(3), np.zeros(3)
        inl = []
        for _ in range(cpu_iters):
            acc, _, _, R, t, ok = orc.icp_step(index, opts, R, t, GATE)
            inl.append(float(acc[0]))
        wall = time.perf_counter() - t_0
        eng.set_source_from_depth(depth, K, Tp, 1000.0, count=False)
        g = eng.icp(cpu_iters, 0.0, GATE, with_trace=True)
        parity = {"iterations": cpu_iters, "R_fro_vs_cpu": float(np.linalg.norm(g.R - R)), "t_m_vs_cpu": float(np.linalg.norm(g.t - t)),
                  "max_inlier_count_difference": float(np.abs(g.trace[:, 0] - np.array(inl)).max()),
                  "valid_pixels_device_vs_oracle": [int(eng.n_source), int(mask.sum())],
                  "launches_compared": "".join("SWL"[int(v)] for v in eng.list_plan()[:cpu_iters]),
                  "tolerance": {"R_fro": 1e-4, "t_m": 1e-3}, "oracle_wall_s": wall, "oracle_backend": "scipy cKDTree (oracle.GoldIndex backend='ckdtree')"}
        parity["within_tolerance"] = bool(parity["R_fro_vs_cpu"] <= 1e-4 and parity["t_m_vs_cpu"] <= 1e-3)
    # the same call with the frame already in HBM (a decoder's output, a torch tensor): no PCIe in it
    dev_wall = None
    try:
        import torch
        d_dev = torch.from_numpy(depth).to(f"cuda:{eng.device}")
        eng.set_source_from_depth(d_dev, K, Tp, 1000.0, count=False)
        eng.sync()
        ws = []
        for _ in range(5):
            t0 = time.perf_counter()
            eng.set_source_from_depth(d_dev, K, Tp, 1000.0, count=False)
            ws.append((time.perf_counter() - t0) * 1e3)
            eng.sync()
        dev_wall = float(np.median(ws))
        eng.set_source_from_depth(depth, K, Tp, 1000.0)
    except Exception:          # pragma: no cover - torch is present wherever this runs
        dev_wall = None
    unproj_bytes = depth.size * depth.itemsize + 16.0 * depth.size   # depth read once + one float4 slot per pixel written (in place)
    k_us = float(np.min(kern[1:])) * 1e3
    return {"workload": "configs[2]: 1280x960 float32 depth frame -> k_unproject -> 50 ICP iterations against the 1M-point gold map",
            "valid_pixels": int(n),
            # (key names: rounds 1-3 reported the call WITH the count under upload_plus_unproject_ms_host_wall; that key keeps
            # that meaning, and the call that only enqueues has its own)
            "upload_plus_unproject_ms_host_wall": float(np.median(walls[1:])),
            "upload_plus_unproject_enqueue_only_ms_host_wall": float(np.median(walls_async[1:])),
            "upload_plus_unproject_note": "upload_plus_unproject_ms_host_wall: the call with the count of valid pixels returned (upload + "
                                          "kernel + wait); ..._enqueue_only_...: n_valid not asked for — upload and kernel enqueued, nothing "
                                          "waited for, the alignment follows on the stream (what a pipeline uses)",
            "parity_vs_cpu": parity, "list_launches": lists,
            "depth_frame_to_transform_ms_host_wall": float(np.median(frame_to_T[1:])),
            "unproject_call_ms_host_wall_device_resident_frame": dev_wall,
            "unproject_device_us": k_us, "icp_ms": float(np.median(icp_ms[1:])),
            "it_per_s": ITERS / (float(np.median(icp_ms[1:])) * 1e-3),
            "rmse_mm": float(r.rmse * 1e3), "inliers": float(r.inliers),
            "roofline_unproject": {"bound": "hbm", "kernel": "k_unproject (one streaming pass, in place)", "bytes": unproj_bytes,
                                   "achieved": unproj_bytes / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": unproj_bytes / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS},
            "table_words_per_point": float(res_c.trace[:, 18].mean() / n), "candidates_per_point": float(res_c.trace[:, 19].mean() / n)}


def config_c4(eng, device):
    """configs[3]: batch-32 detector inference on PyTorch-ROCm + one-call 2D->3D box lift."""
    import torch
    from mapalign import detect

    torch.manual_seed(0)
    n_frames = 32
    model = detect.build_yolov10_like()
    det = detect.Detector(model, None, batch_size=n_frames, conf_thresh=0.0, device=f"cuda:{device}", half=True)
    imgs = torch.randint(0, 256, (n_frames, 640, 640, 3), dtype=torch.uint8).to(f"cuda:{device}")
    for _ in range(3):                                   # MIOpen picks its kernels on the first calls
        raw, _ = det.infer(imgs)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        raw, _ = det.infer(imgs)
    torch.cuda.synchronize()
    infer_ms = (time.perf_counter() - t0) / reps * 1e3
    det.conf_thresh = float(torch.quantile(raw[:, :, 4].flatten().float(), 0.9))     # ~30 boxes per frame
    preds = det(imgs)
    rng = np.random.default_rng(5)
    depth = rng.uniform(500.0, 9000.0, size=(n_frames, 192, 256)).astype(np.float32)   # dataset.py:19-20 size
    poses = np.tile([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0], (n_frames, 1))
    poses[:, :3] = rng.uniform(-5, 5, size=(n_frames, 3))
    intr = {"fx": 1500.0, "fy": 1500.0, "cx": 320.0, "cy": 320.0, "image_width": 640, "image_height": 640}
    detect.lift_batch(eng, preds, depth, intr, poses)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = detect.lift_batch(eng, preds, depth, intr, poses)
    lift_ms = (time.perf_counter() - t0) / reps * 1e3
    n_boxes = sum(len(v) for v in out.values())
    t0 = time.perf_counter()
    preds2 = det(imgs)
    detect.lift_batch(eng, preds2, depth, intr, poses)
    e2e_ms = (time.perf_counter() - t0) * 1e3
    # the damage leg (detector.py:119-127 -> classifier.py:69-88): BEiT of the reference's architecture (random init)
    # on every kept box, seen through the reference's crop geometry (box warped onto the frame, processor resize)
    damage = {"skipped": "transformers has no BeitForImageClassification here"}
    beit = detect.build_beit_like(num_labels=3)
    if beit is not None:
        det_d = detect.Detector(model, beit, batch_size=n_frames, conf_thresh=det.conf_thresh, device=f"cuda:{device}", half=True,
                                crop="full_frame")
        raw_d, x_d = det_d.infer(imgs)
        keep = raw_d[:, :, 4] >= det_d.conf_thresh
        for _ in range(2):
            det_d._damage(x_d, raw_d, keep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            labels = det_d._damage(x_d, raw_d, keep)
        torch.cuda.synchronize()
        dmg_ms = (time.perf_counter() - t0) / 3 * 1e3
        n_crops = int(keep.sum())
        damage = {"classifier": "BeitForImageClassification (base, patch 16, 224; random init)", "crop": "full_frame (detector.py:197-231)",
                  "crops_per_detector_batch": n_crops, "classifier_ms_per_detector_batch": dmg_ms,
                  "classifier_ms_per_batch_of_32_crops": dmg_ms / max(n_crops / 32.0, 1e-9), "crops_per_s": n_crops / (dmg_ms * 1e-3),
                  "classes_seen": sorted(set(int(v) for v in labels.cpu().numpy()))}
        del det_d, beit
    return {"workload": "configs[3]: batch-32 3x640x640 detector inference (random-init YOLOv10-n-shaped stand-in, fp16, "
                        "PyTorch-ROCm; ultralytics / weights are not available here) + 2D->3D lift of all boxes of the "
                        "32 frames in one ma_lift_boxes_batch call (192x256 depth frames)",
            "model_parameters_M": sum(p.numel() for p in model.parameters()) / 1e6,
            "inference_ms_per_batch": infer_ms, "frames_per_s": n_frames / (infer_ms * 1e-3),
            "boxes_lifted": int(n_boxes), "lift_ms_per_batch_host_wall": lift_ms, "boxes_per_s": n_boxes / (lift_ms * 1e-3),
            "detect_rows_plus_lift_ms_per_batch": e2e_ms, "device_calls_per_batch_for_the_lift": 1, "damage_leg": damage}


def config_concurrent(gold, comp, device, R0):
    """K independent alignments of resident 1M-point pairs iterating at the same time on this GPU (one
    context = one stream and one host thread each): what a fleet of pairs gets out of one GPU.  Not the
    headline metric (that is ONE pair); reported beside it."""
    import threading

    import mapalign

    from mapalign import synthetic as syn

    engs, truth = [], []
    for k in range(4):                                   # four different pairs (seeds 0-3), as a fleet would hold
        g, c, Rk, _ = (gold, comp, R0, None) if k == 0 else syn.make_pair(k, len(gold))
        e = mapalign.MapAligner(device)
        e.set_gold(g)
        e.set_source(c)
        e.icp(ITERS, 0.0, GATE)
        engs.append(e)
        truth.append(Rk)
    out, reps = {}, 20
    for K in (1, 2, 4):
        errs = [0.0] * K

        def work(i):
            for _ in range(reps):
                r = engs[i].icp(ITERS, 0.0, GATE)
            errs[i] = float(np.linalg.norm(r.R - truth[i]))

        ths = [threading.Thread(target=work, args=(i,)) for i in range(K)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        out[f"{K}_at_once"] = {"it_per_s_total": K * reps * ITERS / dt, "ms_per_alignment": dt / reps * 1e3,
                               "R_fro_vs_truth_max": max(errs)}
    for e in engs:
        e.close()
    out["workload"] = ("K independent alignments of the configs[1] pair (1M points, 50 iterations) resident in HBM and iterating "
                       "concurrently on one GPU, one context / HIP stream / host thread each")
    return out


def config_strays(eng, gold, comp, R0, t0):
    """The headline pair with 20 stray points over 50 times its extent in BOTH clouds (reflections, sensor noise: real
    RTAB-Map exports carry them, the reference's ingest removes none).  The grid covers the bulk of the map, the strays
    sit in its boundary cells (DESIGN.md 3 "Stray points"): the alignment took 2 400 times as long before that."""
    rng = np.random.default_rng(13)
    lo, hi = gold.min(0), gold.max(0)
    ctr, ext = 0.5 * (lo + hi), float((hi - lo).max())
    stray = lambda k: (ctr + rng.uniform(-0.5, 0.5, size=(k, 3)) * ext * 50.0).astype(np.float32)
    g, c = np.concatenate([gold, stray(20)]), np.concatenate([comp, stray(20)])
    grid = eng.set_gold(g)
    eng.set_source(c)
    for _ in range(2):
        eng.icp(ITERS, 0.0, GATE)
    ms = [eng.icp(ITERS, 0.0, GATE).gpu_ms for _ in range(5)]
    res = eng.icp(ITERS, 0.0, GATE)
    return {"workload": "configs[1] with 20 stray points over 50x the scene's extent in the gold map AND in the scan (not the headline configuration)",
            "icp_ms": float(np.median(ms)), "it_per_s": ITERS / (float(np.median(ms)) * 1e-3), "grid_cell_m": grid["cell"],
            "grid_dims": list(grid["dims"]), "grid_covers_bulk_only": bool(eng.grid_box()["bulk_only"]),
            "list_launches": "".join("SWL"[int(v)] for v in eng.list_plan()),
            "final_error_vs_truth": {"R_fro": float(np.linalg.norm(res.R - R0)), "t_m": float(np.linalg.norm(res.t - t0))},
            "note": "the same pair took 5.3-7.9 s per alignment while the grid covered the bounding box (profiles/r05_stray_points.txt)"}


def config_c5(eng, seed=0, cpu_iters=ITERS):
    """The per-GPU workload of configs[4] (8 pairs x 2M points, one pair per GPU): one 2M pair here — timed over its
    50 iterations, checked against the ground truth, and compared with the CPU oracle after the SAME `cpu_iters`
    iterations of this independently resampled pair (all 50 by default — past the pair's list launches — on the oracle's
    cKDTree backend: ~0.6 s per iteration at 2M points where the scikit-learn backend needs 4-9 s)."""
    from mapalign import synthetic as syn

    gold, comp, R0, t0 = syn.make_pair(seed, 2_000_000)
    grid = eng.set_gold(gold)
    grid = eng.set_gold(gold)           # warm build
    eng.set_source(comp)
    for _ in range(2):
        eng.icp(ITERS, 0.0, GATE)
    ms = [eng.icp(ITERS, 0.0, GATE).gpu_ms for _ in range(5)]
    res = eng.icp(ITERS, 0.0, GATE)
    out = {"workload": "configs[4] per-GPU share: one 2M-point pair, 50 ICP iterations (the 8-GPU run is 8 such pairs, one per GPU)",
           "icp_ms": float(np.median(ms)), "it_per_s": ITERS / (float(np.median(ms)) * 1e-3), "grid_build_ms": grid["build_ms"],
           "grid_cell_m": grid["cell"], "list_launches": dict(eng.list_stats(), plan="".join("SWL"[int(v)] for v in eng.list_plan())),
           "final_error_vs_truth": {"R_fro": float(np.linalg.norm(res.R - R0)), "t_m": float(np.linalg.norm(res.t - t0))}}
    if cpu_iters > 0:
        from oracle import icp_oracle as orc
        t_0 = time.perf_counter()
        index = orc.GoldIndex(gold, n_jobs=-1, backend="ckdtree")
        R, t = np.eye(3), np.zeros(3)
        inl = []
        for _ in range(cpu_iters):
            acc, _, _, R, t, ok = orc.icp_step(index, comp, R, t, GATE)
            inl.append(float(acc[0]))
        wall = time.perf_counter() - t_0
        g = eng.icp(cpu_iters, 0.0, GATE, with_trace=True)
        out["parity_vs_cpu"] = {"iterations": cpu_iters, "R_fro_vs_cpu": float(np.linalg.norm(g.R - R)),
                                "t_m_vs_cpu": float(np.linalg.norm(g.t - t)),
                                "max_inlier_count_difference": float(np.abs(g.trace[:, 0] - np.array(inl)).max()),
                                "launches_compared": "".join("SWL"[int(v)] for v in eng.list_plan()[:cpu_iters]),
                                "tolerance": {"R_fro": 1e-4, "t_m": 1e-3}, "oracle_wall_s": wall,
                                "oracle_backend": "scipy cKDTree (oracle.GoldIndex backend='ckdtree')",
                                "note": f"{cpu_iters} iterations of the pair on both sides"}
        out["parity_vs_cpu"]["within_tolerance"] = bool(out["parity_vs_cpu"]["R_fro_vs_cpu"] <= 1e-4 and out["parity_vs_cpu"]["t_m_vs_cpu"] <= 1e-3)
    return out


def run_guarded(fn, seconds, device=None):
    """Run ``fn`` on a helper thread and wait at most ``seconds`` for it -> (result, timed_out).
    The N > 1 run executes library-RCCL code that no builder's one-GPU box can rehearse; if a communicator
    bootstrap or a collective ever hangs on the real node, the headline line must still come out.  A timed-out
    leg leaves its (daemon) thread behind, so the caller finishes with os._exit once the line is written."""
    import threading

    box = {}

    def work():
        try:
            if device is not None:
                import torch
                torch.cuda.set_device(device)          # the current device is per thread
            box["result"] = fn()
        except BaseException as e:                    # noqa: BLE001 - reported in the line
            box["result"] = {"error": f"{type(e).__name__}: {e}"}

    th = threading.Thread(target=work, name="bench-guarded-leg", daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": f"no answer within {seconds:.0f} s; the headline value above was measured before this leg and is unaffected"}, True
    return box.get("result"), False


def sharded_region(args, world, rank, local_rank, barrier, dist, torch):
    """ONE 1M-point pair (seed 0 on every rank) over all ranks through ma_icp_sharded, timed like
    the headline region.  Every rank returns the same dict; a failure anywhere is agreed on by all
    ranks BEFORE anyone enters a collective of the library, so no rank is left waiting."""
    import mapalign
    from mapalign import dist as mdist
    from mapalign import synthetic as syn

    dev = f"cuda:{local_rank}"

    def all_ok(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) >= 1.0

    rccl = args.backend == "nccl"        # (a single-GPU rehearsal over gloo: RCCL cannot join two ranks that share one device;
    tdev = dev if rccl else "cpu"        #  the peer-to-peer exchange can — its rows are mapped through IPC handles either way)
    eng, err = None, None
    try:
        gold, comp, R0, t0 = syn.make_pair(0, args.points)
        lo, hi = mdist.shard_bounds(len(comp), world, rank)
        eng = mapalign.MapAligner(local_rank)
        eng.set_gold(gold, args.cell)
        eng.set_source(comp[lo:hi])
        if rccl:
            eng.comm_unique_id()             # can this rank reach librccl at all?
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    if not all_ok(err is None):
        if eng is not None:
            eng.close()
        return {"error": err or "another rank could not stage its shard / load librccl"}
    T_rccl = None
    try:
        if not rccl:
            raise RuntimeError("skipped: single-GPU rehearsal over gloo (RCCL cannot join two ranks that share one device)")
        mdist.bootstrap_comm(eng)
        for _ in range(args.warmup):
            out = eng.icp_sharded(ITERS, 0.0, GATE)
        barrier()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            out = eng.icp_sharded(ITERS, 0.0, GATE)
        barrier()
        elapsed = time.perf_counter() - t_start
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        res = {"value": args.steps * ITERS / elapsed, "unit": "ICP iterations/s", "ms_per_step": elapsed / args.steps * 1e3,
               "scaling": "strong", "workload": f"ONE {args.points}-point pair, comparison cloud block-partitioned over {world} GPUs, "
                                               f"gold map replicated, 50 iterations",
               "library_rccl": "ma_icp_sharded: ncclAllReduce(17 x int64 fixed point: exact, order-free) per iteration inside libmapalign",
               "final_error_vs_truth": {"R_fro": float(np.linalg.norm(out.R - R0)), "t_m": float(np.linalg.norm(out.t - t0))}}
        T_rccl = out.T.copy()
        eng.comm_destroy()
    except Exception as e:       # pragma: no cover - depends on the node
        res = {"skipped": str(e)[9:]} if str(e).startswith("skipped: ") else {"error": f"{type(e).__name__}: {e}"}
    # ---- the same pair through the peer-to-peer exchange (no library collective on the data path; two launches per iteration)
    p2p, err = None, None
    try:
        mdist.bootstrap_p2p(eng)
        eng.comm_config("p2p")
    except Exception as e:       # pragma: no cover - depends on the node (IPC handles, peer access)
        err = f"{type(e).__name__}: {e}"
    if not all_ok(err is None):
        p2p = {"error": err or "another rank could not set up the exchange"}
    else:
        try:
            for _ in range(args.warmup):
                out = eng.icp_sharded(ITERS, 0.0, GATE)
            barrier()
            t_start = time.perf_counter()
            for _ in range(args.steps):
                out = eng.icp_sharded(ITERS, 0.0, GATE)
            barrier()
            elapsed = time.perf_counter() - t_start
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            p2p = {"value": args.steps * ITERS / elapsed, "unit": "ICP iterations/s", "ms_per_step": elapsed / args.steps * 1e3,
                   "transport": "ma_p2p_*: the 17 sums written as tagged 8-byte words into every peer's rows by k_reduce_exchange_solve "
                                "(sum + exchange + solve in one launch), rows mapped with hipIpcOpenMemHandle",
                   "bitwise_equal_to_rccl_leg": None if T_rccl is None else bool(np.array_equal(out.T, T_rccl)),
                   "final_error_vs_truth": {"R_fro": float(np.linalg.norm(out.R - R0)), "t_m": float(np.linalg.norm(out.t - t0))}}
            barrier()
            eng.p2p_destroy()
        except Exception as e:   # pragma: no cover - depends on the node; the exchange's wait is bounded (MA_ERR_COMM after 2 s)
            p2p = {"error": f"{type(e).__name__}: {e}"}
    res["p2p"] = p2p
    eng.close()
    return res


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher's parent before torch or libmapalign is loaded
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version
    # banner to stdout when a communicator is created), so everything but that line is sent to
    # stderr: fd 1 is pointed at fd 2 for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
    if world > 1 or args.rehearse_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if args.backend == "nccl":      # RCCL
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1 or args.rehearse_sharded:
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
    grid_first = eng.set_gold(gold, args.cell)          # first build of the context: includes its hipMallocs
    grid = eng.set_gold(gold, args.cell)                # steady state (what every later map costs)
    lib_rccl = None
    if sharded:
        lo, hi = mdist.shard_bounds(len(comp), world, rank)
        eng.set_source(comp[lo:hi])
        mdist.bootstrap_comm(eng)                        # ncclUniqueId over torch.distributed; the collective itself is in the library
        lib_rccl = "ma_icp_sharded: ncclAllReduce(17 x int64 fixed point: exact, order-free) per iteration inside libmapalign"

        def step():
            return eng.icp_sharded(ITERS, 0.0, GATE)
    else:
        eng.set_source(comp)
        K = max(1, args.pairs_per_gpu)
        if K == 1:
            def step():
                return eng.icp(ITERS, 0.0, GATE)
        else:
            from concurrent.futures import ThreadPoolExecutor
            others = []
            for k in range(1, K):                       # further pairs of this rank: other seeds, own contexts
                gk, ck, _, _ = syn.make_pair(seed + 1000 * k, args.points)
                ek = mapalign.MapAligner(local_rank)
                ek.set_gold(gk, args.cell)
                ek.set_source(ck)
                others.append(ek)
            pool = ThreadPoolExecutor(max_workers=K)

            def step():
                futs = [pool.submit(e.icp, ITERS, 0.0, GATE) for e in [eng] + others]
                return [f.result() for f in futs][0]

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
    pairs_per_step = 1 if sharded else world * max(1, args.pairs_per_gpu)
    value = pairs_per_step * args.steps * ITERS / elapsed

    # ---- result check + fleet stats (outside the timed region) ---------------------------
    T = out.T
    err = np.array([float(np.linalg.norm(T[:3, :3] - R0)), float(np.linalg.norm(T[:3, 3] - t0))])
    if world > 1:
        if not sharded:       # fleet statistics: max over ranks (the library's own RCCL path is exercised by the sharded leg below)
            fleet = torch.tensor(err, dtype=torch.float64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
            dist.all_reduce(fleet, op=dist.ReduceOp.MAX)
            err = fleet.cpu().numpy()
        else:
            err = eng.comm_allreduce(err, "max")

    # ---- the second curve of SURVEY.md 8e in the same run: one pair sharded over all ranks ----
    line = None
    if rank == 0:
        # ---- roofline of the dominant kernel, measured live (untimed extra passes) -----
        roof = None
        if not sharded:
            res_t = eng.icp(ITERS, 0.0, GATE, time_kernels=True)
            kt = eng.kernel_times()
            it_ms = eng.iteration_times()                    # [iterations, 2]: search launch, reduce + solve launch
            plan_t = eng.list_plan()
            res_c = eng.icp(ITERS, 0.0, GATE, count_work=True, with_trace=True)
            plan = eng.list_plan()                           # what each iteration launched: 0 search, 1 search + write lists, 2 list launch
            same_plan = len(plan) == len(plan_t) and bool((plan == plan_t).all())
            n_blocks, block = eng.search_geometry()          # the grid actually launched (padded), not ceil(N / 256)
            # the graded kernel is the SEARCH kernel, k_nn_accum<false,0,0> — the launches that read lists instead (a
            # different instantiation, far fewer bytes asked for and a shorter launch) are reported beside it
            b_launch, rows_pp, cands_pp, b_strict = algorithmic_bytes(res_c.trace, len(comp), n_blocks, plan, 0)
            sel = lambda k: it_ms[:len(plan_t), 0][plan_t[:len(it_ms)] == k] if len(it_ms) else np.zeros(0)
            t_launch_s = float(sel(0).mean()) * 1e-3 if same_plan and len(sel(0)) else kt["nn_ms"] / max(kt["nn_launches"], 1) * 1e-3
            achieved = b_launch / t_launch_s / 1e9
            lists = {"launches_of_the_50": {"search": int((plan == 0).sum()), "search_and_write_lists": int((plan == 1).sum()),
                                            "read_lists": int((plan == 2).sum())}}
            for name, k in (("read_lists", 2), ("search_and_write_lists", 1)):
                if same_plan and (plan == k).any():
                    bk, wk, ck, sk = algorithmic_bytes(res_c.trace, len(comp), n_blocks, plan, k)
                    tk = float(sel(k).mean()) * 1e-3
                    lists[name] = {"launch_us": tk * 1e6, "bytes_per_launch": bk, "candidates_per_point": ck,
                                   "table_words_per_point": wk, "frac": bk / tk / 1e9 / HBM_PEAK_GBPS}
            # the loop-wide figure: every launch with the bytes of ITS kind, over the time of all launches
            all_t = kt["nn_ms"] / max(kt["nn_launches"], 1) * 1e-3
            frac_all = None
            if same_plan and len(it_ms):
                tot_b = sum(algorithmic_bytes(res_c.trace, len(comp), n_blocks, plan, k)[0] * int((plan == k).sum()) for k in (0, 1, 2))
                tot_t = float(it_ms[:len(plan_t), 0].sum()) * 1e-3
                frac_all = tot_b / tot_t / 1e9 / HBM_PEAK_GBPS
                lists["all_launches"] = {"launch_us": all_t * 1e6, "bytes_all_launches": tot_b, "time_all_launches_us": tot_t * 1e6,
                                         "achieved": tot_b / tot_t / 1e9, "frac": frac_all,
                                         "note": "every launch of the loop with the algorithmic bytes of its own kind / the summed launch "
                                                 "times; list launches ask for a fifth of a search's bytes (work avoided), so this figure "
                                                 "falls as more of the loop runs on lists while iterations/s rise"}
            # HBM-side traffic comes from separate rocprofv3 --pmc passes (tools/pmc_traffic.sh);
            # it is only quoted while the kernel sources are the ones it was measured on
            traffic, traffic_note = None, "not measured for this workload"
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath) and args.points == N_POINTS and args.cell == 0.0:
                try:
                    tj = json.load(open(tpath))
                    if tj.get("kernel_source_sha") == kernel_source_sha():
                        traffic = tj.get("k_nn_accum_hbm_bytes_per_launch")
                        traffic_note = f"rocprofv3 --pmc passes of tools/pmc_traffic.sh on kernel sources {tj.get('kernel_source_sha')}"
                    else:
                        traffic_note = (f"profiles/pmc_traffic.json was measured on kernel sources {tj.get('kernel_source_sha')}, "
                                        f"this run is {kernel_source_sha()}: stale, not quoted")
                except Exception:
                    traffic = None
            measured_gbps = traffic / t_launch_s / 1e9 if traffic else None
            roof = {"bound": "hbm", "kernel": "k_nn_accum", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "frac_scope": (f"the {int((plan == 0).sum())} of {len(plan)} launches that search (k_nn_accum<false,0,0>); "
                                   f"BENCH_r02 and earlier graded all 50 (every launch searched then) - compare frac_all_launches"),
                    "frac_all_launches": frac_all,
                    "traffic": traffic, "traffic_note": traffic_note,
                    # how to read `frac`: algorithmic bytes are mostly served by L1/L2 at this size (working set ~80 MB);
                    # what reaches the fabric is `traffic`, and the kernel is bound by instruction issue / L1 lookups, not HBM
                    "frac_survey_8d": b_strict / t_launch_s / 1e9 / HBM_PEAK_GBPS,
                    "measured_hbm_gbps": measured_gbps,
                    "measured_hbm_frac": measured_gbps / HBM_PEAK_GBPS if measured_gbps else None,
                    "cell_edge_note": (f"SURVEY 8d fixes the cell edge to the 0.10 m gate; this run uses the density-driven "
                                       f"{grid['cell'] * 100:.2f} cm (fewer candidates per query: LOWERS the algorithmic bytes, so it "
                                       f"does not inflate frac)") if args.cell == 0.0 else f"cell edge given: {args.cell} m",
                    "launch_us_note": "hipEvents between every launch of an untimed extra run (adds ~2 us per launch; rocprofv3: profiles/)",
                    "workgroups_per_launch": n_blocks, "points_per_workgroup": block,
                    "bytes_per_launch": b_launch, "bytes_per_launch_survey_8d": b_strict, "launch_us": t_launch_s * 1e6,
                    "table_words_per_point": rows_pp, "candidates_per_point": cands_pp,
                    "compulsory_bytes_per_launch": 24.0 * len(comp) + 16.0 * len(gold),
                    "compulsory_frac": (24.0 * len(comp) + 16.0 * len(gold)) / t_launch_s / 1e9 / HBM_PEAK_GBPS,
                    "solve_kernel_us": kt["solve_ms"] / max(kt["solve_launches"], 1) * 1e3,
                    # DESIGN.md 5 "List launches": the launches of the 50 that read per-query candidate lists instead of
                    # searching (another instantiation of the kernel); `frac`, `achieved`, `launch_us`, `bytes_per_launch`
                    # above are those of the search launches alone
                    "kernel_instantiation": "k_nn_accum<false, 0, 0, false> (search launches; the fourth parameter: a grid over the bulk of a map with stray points)",
                    "list_launches": lists,
                    "timed_loop_it_per_s": ITERS / (res_t.gpu_ms * 1e-3)}
        cpu = parity = e2e = None
        extra = {}
        if world == 1:
            if roof is not None and args.points == N_POINTS and args.cell == 0.0 and not args.no_extra_configs:
                try:
                    roof["survey_8d_cell"] = survey_cell_leg(gold, comp, local_rank)
                except Exception as e:   # an auxiliary leg must never take the headline line down
                    roof["survey_8d_cell"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.headline_only:
                e2e = end_to_end(gold, comp, local_rank)
            if not args.no_cpu_baseline:
                cpu, parity = cpu_baseline_and_parity(eng, gold, comp, args.cpu_iters)
            if not args.no_extra_configs and args.points == N_POINTS:
                for name, fn in (("c3", lambda: config_c3(eng, gold, 0 if args.no_cpu_baseline else ITERS)), ("c4", lambda: config_c4(eng, local_rank)),
                                 ("c5_per_gpu", lambda: config_c5(eng, cpu_iters=0 if args.no_cpu_baseline else ITERS)),
                                 ("stray_points", lambda: config_strays(eng, gold, comp, R0, t0)),
                                 ("concurrent_pairs", lambda: config_concurrent(gold, comp, local_rank, R0))):
                    try:
                        extra[name] = fn()
                    except Exception as e:   # an auxiliary leg must never take the headline line down
                        extra[name] = {"error": f"{type(e).__name__}: {e}"}
        line = {
            "metric": "ICP iterations/sec on 1M-pt cloud pair; final (R,t) error vs CPU ref",
            "value": value, "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f32 (f64 accumulate/solve)",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: 1M-point synthetic cloud pair, 50 ICP iterations, 1xMI355X"
                                    if args.points == N_POINTS else
                                    f"{args.points}-point synthetic cloud pair (configs[1] at another size), 50 ICP iterations")
                       + (f" - {args.pairs_per_gpu} independent pairs iterating at once per GPU (NOT the headline configuration)"
                          if args.pairs_per_gpu > 1 and not sharded else "")
                       + ("" if world == 1 else f" - one pair per GPU x{world}" if not sharded
                          else f" - one pair sharded over {world} GPUs, all-reduce of the 17 int64 fixed-point sums per iteration inside the library"),
                       "points_gold": len(gold), "points_comparison": len(comp), "iterations_per_step": ITERS,
                       "max_corr_dist_m": GATE, "grid_cell_m": grid["cell"], "grid_dims": list(grid["dims"]),
                       "grid_build_ms": grid["build_ms"], "grid_build_ms_first_build_of_context": grid_first["build_ms"],
                       "parallelism": "pair-per-gpu" if not sharded else "sharded-source",
                       "kernel_source_sha": kernel_source_sha(), "library_rccl": lib_rccl},
            "final_error_vs_truth": {"R_fro_max_over_ranks": float(err[0]), "t_m_max_over_ranks": float(err[1])},
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_cpu": parity,
        }
        if e2e:
            line["config"].update(e2e)
        if cpu:
            line["speedup_vs_cpu_baseline"] = value / cpu["value"]
        if extra:
            line["configs"] = extra
    # ---- the second curve of SURVEY.md 8e in the same run: one pair sharded over all ranks, LAST and under a
    # watchdog — the headline line is complete before this leg starts and comes out whatever the leg does
    wedged = False
    if (world > 1 or args.rehearse_sharded) and not sharded:
        sharded_leg, wedged = run_guarded(lambda: sharded_region(args, world, rank, local_rank, barrier, dist, torch),
                                          args.sharded_timeout, local_rank)
        if line is not None:
            line["sharded"] = sharded_leg
            line["config"]["library_rccl"] = (sharded_leg or {}).get("library_rccl") or (sharded_leg or {}).get("error") \
                or (sharded_leg or {}).get("skipped")
    # The line goes out BEFORE any teardown or collective: a peer that wedged in RCCL would otherwise hold rank 0 in the
    # barrier below until the process-group timeout, and the line — complete since before the sharded leg — would be lost.
    if line is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if wedged:                 # a stuck collective cannot be cancelled: leave without the interpreter's shutdown, and NOT with
        os._exit(3)            # success — the launcher (and self_launch, which still relays the line) must see that a rank hung

    def teardown():
        eng.close()
        if world > 1 or args.rehearse_sharded:
            barrier()
            dist.destroy_process_group()
        return {}

    _, stuck = run_guarded(teardown, 60.0, local_rank)
    if stuck:                  # a PEER wedged: this rank's barrier never returns
        os._exit(3)


if __name__ == "__main__":
    main()
