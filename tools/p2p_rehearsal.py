"""Rehearsal of the sharded loop over the peer-to-peer exchange (ma_p2p_*; row e) on ONE GPU — what can be exercised
without a multi-GPU node: the protocol, the kernel, the bootstrap, the timeout.  Prints one JSON line.

  python tools/p2p_rehearsal.py --mode threads --ranks 4      # 4 contexts of one process, one host thread each
  python tools/p2p_rehearsal.py --mode procs --ranks 2        # 2 processes (gloo carries the blobs; rows mapped with hipIpc*)
  python tools/p2p_rehearsal.py --mode mixed --ranks 8 --procs 4   # 4 processes x 2 contexts: 8 ranks within the box's process limit
  python tools/p2p_rehearsal.py --mode timeout                # one of two ranks never runs: MA_ERR_COMM, not a hang

Every mode compares with ma_icp of the whole cloud on one context: the transforms must be the same BITS.
Run in a fresh process: the contexts' streams must land on different hardware queues (a waiting exchange kernel in front
of a peer's search kernel in the same queue would only end by the timeout — which is what happened to the 4-context run
of profiles/r05_variants_ab.txt call 3 under the runtime's default of 4 queues per process): the script raises
GPU_MAX_HW_QUEUES before the runtime starts.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["threads", "procs", "mixed", "timeout"], default="threads")
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--procs", type=int, default=4, help="mixed: processes the ranks are spread over (ranks / procs contexts each)")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--tol", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=5)
    return ap.parse_args()


def reference(gold, comp, iters, tol):
    import mapalign
    with mapalign.MapAligner(0) as full:
        full.set_pair(gold, comp)
        ref = full.icp(iters, tol, 0.10)
        ms = sorted(full.icp(iters, tol, 0.10).gpu_ms for _ in range(5))[2]
    return ref, ms


def run_threads(a):
    import numpy as np
    import mapalign
    from mapalign import dist as mdist, synthetic as syn
    gold, comp, _, _ = syn.make_pair(0, a.points)
    ref, ref_ms = reference(gold, comp, a.iters, a.tol)
    engs = []
    for r in range(a.ranks):
        lo, hi = mdist.shard_bounds(len(comp), a.ranks, r)
        e = mapalign.MapAligner(0)
        e.set_gold(gold)
        e.set_source(comp[lo:hi])
        engs.append(e)
    blobs = [e.p2p_export() for e in engs]
    for r, e in enumerate(engs):
        e.p2p_init(blobs, r)
    out = [None] * a.ranks
    times = []

    def work(r, rep):
        try:
            out[r] = engs[r].icp_sharded(a.iters, a.tol, 0.10)
        except Exception as ex:          # noqa: BLE001 - reported below
            out[r] = ex

    for rep in range(a.reps):
        th = [threading.Thread(target=work, args=(r, rep)) for r in range(a.ranks)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        times.append((time.perf_counter() - t0) * 1e3)
        bad = [f"rank {r}: {o}" for r, o in enumerate(out) if isinstance(o, Exception)]
        if bad:
            res = {"ok": False, "mode": "threads", "ranks": a.ranks, "error": "; ".join(bad)}
            if all("did not arrive within" in b for b in bad):
                res["note"] = ("every rank ended by the bounded wait: contexts of ONE process share the runtime's hardware queues, and a "
                               "waiting exchange kernel in front of a peer's search kernel in the same queue can only end by the timeout "
                               "- an artefact of rehearsing several ranks in one process on one GPU (ranks in processes of their own, "
                               "--mode procs, have their own queues)")
            return res
    same = all(np.array_equal(o.T, ref.T) and o.iterations == ref.iterations and o.rmse == ref.rmse and o.inliers == ref.inliers
               for o in out)
    res = {"mode": "threads", "ranks": a.ranks, "points": a.points, "iterations": int(out[0].iterations),
           "bitwise_equal_to_one_context": bool(same),
           "gpu_ms_per_rank": [round(float(o.gpu_ms), 4) for o in out], "ma_icp_gpu_ms": round(float(ref_ms), 4),
           "wall_ms": round(sorted(times)[len(times) // 2], 3), "rows": engs[0].p2p_rows}
    for e in engs:
        e.p2p_destroy()
        e.close()
    res["ok"] = bool(same)
    return res


def _proc_worker(rank, world, port, a_dict, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import numpy as np
    import torch.distributed as dist
    from mapalign import dist as mdist, synthetic as syn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gold, comp, _, _ = syn.make_pair(0, a_dict["points"])
    try:
        ms = []
        for _ in range(a_dict["reps"]):
            R, t, info = mdist.sharded_align(gold, comp, max_iter=a_dict["iters"], tol=a_dict["tol"], max_corr_dist=0.10, device=0,
                                             return_info=True, transport="p2p")
            ms.append(info["gpu_ms"])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        q.put((rank, T, info, sorted(ms)[len(ms) // 2], None))
    except Exception as ex:              # noqa: BLE001
        q.put((rank, None, None, None, f"{type(ex).__name__}: {ex}"))
    dist.barrier()
    dist.destroy_process_group()


def run_procs(a):
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    from mapalign import synthetic as syn
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_proc_worker, args=(r, a.ranks, port, {"points": a.points, "iters": a.iters, "tol": a.tol, "reps": a.reps}, q))
             for r in range(a.ranks)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    errs = [g[4] for g in got if g[4]]
    if errs:
        return {"ok": False, "mode": "procs", "ranks": a.ranks, "error": "; ".join(errs)}
    # the reference AFTER the children are gone: this process touches the GPU only now
    gold, comp, _, _ = syn.make_pair(0, a.points)
    ref, ref_ms = reference(gold, comp, a.iters, a.tol)
    same = all(np.array_equal(g[1], ref.T) and g[2]["iterations"] == ref.iterations and g[2]["rmse"] == ref.rmse for g in got)
    return {"ok": bool(same), "mode": "procs", "ranks": a.ranks, "points": a.points, "iterations": int(got[0][2]["iterations"]),
            "bitwise_equal_to_one_context": bool(same), "gpu_ms_per_rank": [round(float(g[3]), 4) for g in sorted(got)],
            "ma_icp_gpu_ms": round(float(ref_ms), 4)}


def _mixed_worker(proc, n_procs, port, a_dict, q):
    """Process `proc` of `n_procs` holds c = ranks / n_procs contexts (global ranks proc * c ... proc * c + c - 1), one host
    thread each: rows of the same process are plain pointers, the other processes' come through hipIpcOpenMemHandle."""
    ranks = a_dict["ranks"]
    c = ranks // n_procs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * c + 2)))
    import numpy as np
    import torch
    import torch.distributed as dist
    import mapalign
    from mapalign import dist as mdist, synthetic as syn
    dist.init_process_group("gloo", rank=proc, world_size=n_procs)
    gold, comp, _, _ = syn.make_pair(0, a_dict["points"])
    err, out, ms = None, [None] * c, []
    engs = []
    try:
        for k in range(c):
            lo, hi = mdist.shard_bounds(len(comp), ranks, proc * c + k)
            e = mapalign.MapAligner(0)
            e.set_gold(gold)
            e.set_source(comp[lo:hi])
            engs.append(e)
        mine = b"".join(bytes(e.p2p_export()) for e in engs)
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        got = [torch.empty_like(t) for _ in range(n_procs)]
        dist.all_gather(got, t)
        blob_len = len(mine) // c
        allb = b"".join(bytes(g.numpy().tobytes()) for g in got)
        blobs = [allb[i * blob_len:(i + 1) * blob_len] for i in range(ranks)]
        for k, e in enumerate(engs):
            e.p2p_init(blobs, proc * c + k)
        dist.barrier()

        def work(k):
            try:
                out[k] = engs[k].icp_sharded(a_dict["iters"], a_dict["tol"], 0.10)
            except Exception as ex:      # noqa: BLE001
                out[k] = ex

        for _ in range(a_dict["reps"]):
            dist.barrier()
            th = [threading.Thread(target=work, args=(k,)) for k in range(c)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            bad = [f"rank {proc * c + k}: {o}" for k, o in enumerate(out) if isinstance(o, Exception)]
            if bad:
                err = "; ".join(bad)
                break
            ms.append([float(o.gpu_ms) for o in out])
    except Exception as ex:              # noqa: BLE001
        err = f"{type(ex).__name__}: {ex}"
    if err:
        q.put((proc, None, err))
    else:
        q.put((proc, [(np.asarray(o.T), int(o.iterations), float(o.rmse), float(o.inliers)) for o in out],
               [round(float(np.median([m[k] for m in ms])), 4) for k in range(c)]))
    try:
        dist.barrier()                  # nobody unmaps rows a peer may still be writing into
        for e in engs:
            e.p2p_destroy()
            e.close()
        dist.destroy_process_group()
    except Exception:                    # noqa: BLE001 - a peer that failed has left
        pass


def run_mixed(a):
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    from mapalign import synthetic as syn
    if a.ranks % a.procs or a.ranks // a.procs < 1:
        return {"ok": False, "mode": "mixed", "error": "--ranks must be a multiple of --procs"}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    args = {"points": a.points, "iters": a.iters, "tol": a.tol, "reps": a.reps, "ranks": a.ranks}
    procs = [ctx.Process(target=_mixed_worker, args=(p, a.procs, port, args, q)) for p in range(a.procs)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    errs = [g[2] for g in got if g[1] is None]
    if errs:
        return {"ok": False, "mode": "mixed", "ranks": a.ranks, "procs": a.procs, "error": "; ".join(errs)[:2000]}
    gold, comp, _, _ = syn.make_pair(0, a.points)
    ref, ref_ms = reference(gold, comp, a.iters, a.tol)
    same = all(np.array_equal(T, ref.T) and it == ref.iterations and rm == ref.rmse and inl == ref.inliers
               for g in got for (T, it, rm, inl) in g[1])
    return {"ok": bool(same), "mode": "mixed", "ranks": a.ranks, "procs": a.procs, "contexts_per_process": a.ranks // a.procs,
            "points": a.points, "iterations": int(got[0][1][0][1]), "bitwise_equal_to_one_context": bool(same),
            "gpu_ms_per_rank": [v for g in sorted(got, key=lambda g: g[0]) for v in g[2]], "ma_icp_gpu_ms": round(float(ref_ms), 4)}


def run_timeout(a):
    os.environ["MAPALIGN_P2P_TIMEOUT_MS"] = "300"
    import mapalign
    from mapalign import _lib, dist as mdist, synthetic as syn
    gold, comp, _, _ = syn.make_pair(1, 200_000)
    engs = []
    for r in range(2):
        lo, hi = mdist.shard_bounds(len(comp), 2, r)
        e = mapalign.MapAligner(0)
        e.set_gold(gold)
        e.set_source(comp[lo:hi])
        engs.append(e)
    blobs = [e.p2p_export() for e in engs]
    for r, e in enumerate(engs):
        e.p2p_init(blobs, r)
    whole = mapalign.MapAligner.combine_sum_bounds([e.sum_bounds() for e in engs])
    for e in engs:
        e.set_sum_bounds(whole)          # (agreed by hand: the silent rank would not take part in the agreement either)
    t0 = time.perf_counter()
    status, msg = None, ""
    try:
        engs[0].icp_sharded(30, 0.0, 0.10)        # rank 1 never runs
    except mapalign.MapAlignError as ex:
        status, msg = ex.status, str(ex)
    waited = time.perf_counter() - t0
    again = None
    try:
        engs[0].icp_sharded(30, 0.0, 0.10)
    except mapalign.MapAlignError as ex:
        again = ex.status
    ok = status == _lib.MA_ERR_COMM and 0.25 <= waited <= 5.0 and again == _lib.MA_ERR_COMM
    # and the context is still good for everything else
    res1 = engs[0].icp(20, 0.0, 0.10)
    return {"ok": bool(ok and res1.iterations == 20), "mode": "timeout", "status": status, "waited_s": round(waited, 3),
            "second_call_status": again, "message": msg[:160]}


def main():
    a = parse()
    if a.mode == "threads":
        # every context's stream on a hardware queue of its own (the runtime's default is 4 queues for all streams of a
        # process, and this process also holds the reference context's streams); ranks on different GPUs or in different
        # processes — the real thing — have their own queues anyway
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * a.ranks + 4)))
    res = {"threads": run_threads, "procs": run_procs, "mixed": run_mixed, "timeout": run_timeout}[a.mode](a)
    print(json.dumps(res))
    sys.exit(0 if res.get("ok") else 1)


if __name__ == "__main__":
    main()
