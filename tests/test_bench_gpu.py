"""GPU: rehearsal of the multi-rank bench on ONE device (MOBROB_DP_SAME_DEVICE=1).  `python bench.py --gpus 2` starts its
two ranks itself; both use cuda:0, rendezvous over gloo and run the C loop mobrob_ppo_train_dp with the host-staged
all-reduce callback at the HEADLINE shape -- everything an 8-GPU run executes except the transport of the sum (RCCL refuses
two ranks on one device).  Checked: the one-line contract, the all-reduce accounting, bit-identical replicas, and that a rank
killed mid-update fails the job with every child reaped.  (the semantics being distributed:
/root/reference/src/mobrob/rl_control/ppo.py:73-74.)"""
import json
import os
import subprocess
import sys
import time

import pytest

from tests.test_bench_cpu import BENCH, _clean_env

pytestmark = pytest.mark.gpu


def _run(extra_env, *argv, timeout=600):
    return subprocess.run([sys.executable, BENCH, *argv], env=_clean_env(MOBROB_DP_SAME_DEVICE="1", **extra_env),
                          capture_output=True, text=True, timeout=timeout)


def test_two_rank_rehearsal_of_the_headline_shape_on_one_device():
    r = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["value"] is None and "rehearsal" in o and o["steps"] == 2 and o["warmup"] == 1
    cfg = o["config"]
    assert cfg["workload"] == "doggo-4096env-2x256" and cfg["parallelism"] == "dp2" and cfg["n_ranks_seen"] == 2
    assert "gloo" in cfg["n_ranks_source"] and cfg["minibatch_per_gpu"] == 65536
    nmb, E = cfg["minibatches_per_epoch"], cfg["n_epochs"]
    assert nmb == 63 and o["allreduces_per_step"] == E * (nmb + 1)        # one per optimizer step + one per epoch
    P = 165145
    assert o["allreduce_bytes"]["gradient_message"] == (P + 8) * 4
    assert o["allreduce_bytes"]["per_step"] == E * (nmb * (P + 8) * 4 + nmb * 32)
    assert o["replicas_bit_identical"] is True and o["rehearsal"]["replicas_bit_identical"] is True
    assert o["phase_ms_per_step"]["allreduce"] > 0 and o["phase_ms_per_step"]["train_grad"] > 0
    assert o["roofline"]["launches"] == 2 * E * nmb


def test_a_rank_killed_mid_update_fails_the_rehearsal_and_children_are_reaped():
    t0 = time.time()
    r = _run({"MOBROB_BENCH_FAULT": "1:40"}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
             "--workload", "point-1024env-2x64")
    assert r.returncode != 0 and not r.stdout.strip(), (r.returncode, r.stdout)
    assert "rank 1 exited with code 17" in r.stderr and "all 2 ranks reaped" in r.stderr, r.stderr[-3000:]
    assert time.time() - t0 < 300


def test_forced_dp_world1_reports_the_engine_communicator():
    """MOBROB_FORCE_DP=1 at world size 1: the RCCL C loop runs; n_ranks_seen comes from ncclCommCount."""
    env = _clean_env(MOBROB_FORCE_DP="1", MASTER_PORT="29671")
    r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload",
                        "point-1024env-2x64"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert o["config"]["n_ranks_seen"] == 1 and "ncclCommCount" in o["config"]["n_ranks_source"]
    assert o["allreduces_per_step"] == 10 * 33 and o["replicas_bit_identical"] is True and o["value"] > 0
