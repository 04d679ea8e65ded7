"""CPU: `python bench.py --gpus N` starts its own ranks (no torchrun), forwards exactly one JSON line from rank 0 and
fails when a rank fails.  Driven through `--dry-run-cpu` (gloo, no PPO work: the launcher, the rendezvous on
127.0.0.1, the barrier + max-over-ranks timing and the one-line contract are what is under test here; the GPU
bench line itself is covered by tests/test_api_gpu.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


@pytest.mark.parametrize("n", [1, 2, 4, 8])   # 8: the driver's full-node SCALE run
def test_bench_launches_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "1", "--dry-run-cpu"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["config"]["n_ranks_seen"] == n and out["config"]["parallelism"] == f"dp{n}"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] is None and "dry_run" in out


def test_bench_under_a_torchrun_environment_does_not_spawn():
    """With RANK/WORLD_SIZE already set (torch.distributed.run form) the process IS a rank: a world-size mismatch is
    an error instead of a second level of children."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-cpu"],
                       env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_fails_when_a_rank_fails():
    """Without a GPU every real rank exits with an error; the launcher must report it (non-zero, no JSON line) and
    not hang on the surviving ranks."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not r.stdout.strip()
    assert "exited with code" in r.stderr


def test_a_rank_killed_inside_an_all_reduce_fails_the_job_and_every_child_is_reaped():
    """MOBROB_BENCH_FAULT=1:3: rank 1 dies (os._exit, no cleanup) inside its third all-reduce while the others sit in
    the same collective.  The launcher must come back non-zero, without a result line, with every child reaped."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "6", "--warmup", "1", "--dry-run-cpu"],
                       env=_clean_env(MOBROB_BENCH_FAULT="1:3"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "rank 1 exited with code 17" in r.stderr and "all 3 ranks reaped" in r.stderr
    assert time.time() - t0 < 120


def test_traffic_figure_is_tied_to_the_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.csrc_sha256()
    assert len(sha) == 64 and sha == bench.csrc_sha256()
    val, note = bench.measured_traffic("void mobrob::k_chain_train<")
    newest = next(r for r in bench.profile_rounds() if os.path.exists(os.path.join(ROOT, "profiles", r, "hbm_traffic_pmc.json")))
    recorded = json.load(open(os.path.join(ROOT, "profiles", newest, "hbm_traffic_pmc.json"))).get("csrc_sha256")
    if recorded == sha:
        assert isinstance(val, int) and val > 0
    else:
        assert val is None and "not reported" in note


def test_blended_matrix_peak_of_the_x3_gradient_kernel():
    """roofline.peak of bench.py for the headline shape: 91.9 % of 3 * F_fwd on the bf16 pipe at six MFMAs per multiply-add,
    the rest on v_mfma_f32; without x3 the f32 matrix peak."""
    import bench
    assert bench.blended_peak(58, 256, 12, False) == (157.3, 0.0)
    peak, share = bench.blended_peak(58, 256, 12, True)
    assert abs(share - 905216.0 / 984576.0) < 1e-12 and abs(share - 0.919) < 1e-3
    ideal = share * 6 / (16 * 157.3) + (1 - share) / 157.3
    assert abs(peak - 1.0 / ideal) < 1e-9 and 369.0 < peak < 370.5


def test_rocm_smi_samples_are_paired_clock_then_power():
    """bench.parse_smi_samples: the text of repeated `rocm-smi --showclocks --showpower` calls -> (power, shader clock) pairs; the
    cap line of --showmaxpower and a power line that no clock line precedes are not samples."""
    import bench
    one = ("GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: mclk clock level: 0: (2000Mhz)\nGPU[0]\t\t: sclk clock level: 1: (%dMhz)\n"
           "=== Power Consumption ===\nGPU[0]\t\t: Current Socket Graphics Package Power (W): %s\n")
    text = "GPU[0]\t\t: Current Socket Graphics Package Power (W): 999.0\n" + one % (2255, "1297.0") + one % (2253, "1288.0") + one % (114, "244.0")
    assert bench.parse_smi_samples(text) == [(1297.0, 2255), (1288.0, 2253), (244.0, 114)]
    assert bench.parse_smi_samples("") == []
    # the blended peak with the sustained pipe rates: below the nominal blend, above the f32 pipe alone
    D, H, A = 58, 256, 12
    peak, share = bench.blended_peak(D, H, A, True)
    capped = bench.power_capped_peak(share)
    assert bench.SUSTAINED_F32_MFMA_TFLOPS < capped < peak and 0.9 < share < 0.93

