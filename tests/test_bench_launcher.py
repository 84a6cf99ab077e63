"""`python bench.py --gpus N` from a plain shell starts the N ranks itself:
command construction and return-code / result-line relay of the launcher, on the CPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT

import bench


def test_launcher_command_is_the_drivers_spelling():
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "5", "--warmup", "1"], 8, 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]      # the caller's arguments, unchanged
    assert 0 < bench.free_port() < 65536


def test_self_launch_relays_one_line_and_the_childs_return_code(capfd):
    child = [sys.executable, "-c",
             "import sys; print('RCCL banner'); print('{\"metric\": \"m\", \"n_gpus\": 2}'); sys.exit(0)"]
    assert bench.self_launch([], 2, command=child) == 0
    out, err = capfd.readouterr()
    assert out.count("\n") == 1 and json.loads(out)["n_gpus"] == 2      # exactly ONE line on stdout
    assert "RCCL banner" in err                                         # everything else goes to stderr
    failing = [sys.executable, "-c", "import sys; sys.exit(7)"]
    assert bench.self_launch([], 2, command=failing) == 7
    silent = [sys.executable, "-c", "pass"]
    assert bench.self_launch([], 2, command=silent) == 1               # rc 0 without a result line is a failure
    # a rank that wedged in a collective leaves with a non-zero code AFTER rank 0 wrote the line: the line is
    # still relayed, and the failure is still reported
    capfd.readouterr()
    wedged = [sys.executable, "-c", "import os, sys; print('{\"metric\": \"m\", \"n_gpus\": 2}'); sys.stdout.flush(); os._exit(3)"]
    assert bench.self_launch([], 2, command=wedged) == 3
    out, _ = capfd.readouterr()
    assert out.count("\n") == 1 and json.loads(out)["metric"] == "m"


def test_plain_shell_run_with_gpus_2_goes_through_the_launcher_before_any_gpu_use():
    """End to end on the CPU box: the child ranks start (gloo rendezvous), find no GPU and stop with the
    product's own message; the parent reports their failure instead of raising the old
    'launch with: python -m torch.distributed.run' SystemExit."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--force-device", "0",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra-configs", "--points", "2000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    err = p.stderr.decode("utf-8", "replace")
    assert "launch with: python -m torch.distributed.run" not in err
    if p.returncode != 0:            # no GPU here: the ranks say so themselves
        assert "needs an MI355X" in err or "no HIP device" in err or "No HIP GPUs" in err, err[-2000:]
    else:                            # on a GPU box the same command yields the line
        assert json.loads(p.stdout.decode().strip().splitlines()[-1])["n_gpus"] == 2


def test_guarded_leg_returns_its_result_or_gives_up_after_the_timeout():
    """The N > 1 run's library-RCCL leg sits under a watchdog: a leg that never answers must not take the headline
    line with it."""
    import time as _time

    res, timed_out = bench.run_guarded(lambda: {"value": 1.0}, 5.0)
    assert res == {"value": 1.0} and not timed_out
    res, timed_out = bench.run_guarded(lambda: (_ for _ in ()).throw(RuntimeError("rccl said no")), 5.0)
    assert not timed_out and "rccl said no" in res["error"]
    t0 = _time.perf_counter()
    res, timed_out = bench.run_guarded(lambda: _time.sleep(30), 0.3)
    assert timed_out and "no answer within" in res["error"] and _time.perf_counter() - t0 < 5.0


def test_algorithmic_bytes_follow_the_layouts_the_kernels_read():
    """bench.py::algorithmic_bytes, the numerator of roofline.achieved: per launch kind, the bytes DESIGN.md 5 counts —
    and for list launches the POSITIONS-ONLY layout the library is built with (MA_LIST_INLINE 0): 16 B point + 4 B
    radius per query, 4 B position + 12 B gold point per list word.  (Round 4 quoted 52 B per query for a while: the
    inline layout that had been measured and dropped.)"""
    import re

    import numpy as np

    src = open(os.path.join(ROOT, "3d-localisation-and-mapping_amd", "csrc", "ma_internal.h")).read()
    assert re.search(r"#define MA_LIST_INLINE 0\b", src), "the list layout changed: revisit bench.algorithmic_bytes(kind=2)"
    n, blocks = 1000, 4
    trace = np.zeros((3, 20))
    trace[:, 0] = [900, 950, 1000]          # matched
    trace[:, 18] = [15000, 4000, 4200]      # table / list words
    trace[:, 19] = [20000, 2300, 2400]      # candidates
    plan = [0, 2, 2]
    b_search, words_pp, cands_pp, b_strict = bench.algorithmic_bytes(trace, n, blocks, plan, 0)
    assert b_search == 24 * n + 4 * 15000 + 12 * 20000 + 12 * 900 + 152 * blocks
    assert b_strict == 12 * n + 4 * 15000 + 12 * 20000
    assert (words_pp, cands_pp) == (15.0, 20.0)
    b_list, words_pp, _, _ = bench.algorithmic_bytes(trace, n, blocks, plan, 2)
    assert b_list == 20 * n + 16 * 4100 + 152 * blocks and words_pp == 4.1
    assert bench.algorithmic_bytes(trace, n, blocks, plan, 1) == (0.0, 0.0, 0.0, 0.0)      # no launch of that kind
