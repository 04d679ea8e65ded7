#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the PPO hot path (BASELINE.json metric) on N MI355X of one node.

A "step" = one full PPO iteration on every rank: a T-step rollout of N vectorised envs from the
device-resident synthetic env source (policy/value forward + sampling + store + time-limit bootstrap),
the GAE(lambda) scan, and E epochs of minibatch updates (forward, loss, backward, global-norm clip, Adam),
with one RCCL gradient all-reduce per optimizer step when N_gpus > 1.  value = env-steps of all ranks / time.

Workload (config.workload): BASELINE.json configs[2] "doggo env (58 obs / 12 act), 1xMI355X, 4096 vec envs,
2x256 MLP" with T=1000, E=5 from the reference YAML (data/configs/doggo-ppo.yaml:12-14) and a stated
minibatch of 65536 per GPU (the YAML's batch_size=100 would be 40960 serial Adam steps per epoch).
Inputs are resident in HBM (the synthetic env source generates observations on the device).

Launch: python bench.py --gpus N --steps K --warmup W      (N > 1 without a torchrun environment: this process
                                                            starts the N ranks itself, one child per GPU)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: obs, act, hidden, envs/GPU, T, E, minibatch/GPU, p_term, time_limit
    "doggo-4096env-2x256": dict(D=58, A=12, H=256, N=4096, T=1000, E=5, B=65536, p_term=1 / 107.0, tl=1000),
    "point-1024env-2x64": dict(D=14, A=2, H=64, N=1024, T=2048, E=10, B=65536, p_term=1 / 119.0, tl=1000),
    # the headline's env count and minibatch with the network the reference YAML actually uses for doggo (2x64)
    "doggo-4096env-2x64": dict(D=58, A=12, H=64, N=4096, T=1000, E=5, B=65536, p_term=1 / 107.0, tl=1000),
    # BASELINE.md §3 B1 "reference-shaped": data/configs/doggo-ppo.yaml, CPU baseline on ONE thread (examples/train.py:13)
    "doggo-ref-16env-2x64": dict(D=58, A=12, H=64, N=16, T=1000, E=5, B=100, p_term=1 / 107.0, tl=1000, cpu_threads=1),
    # BASELINE configs[0]: data/configs/point-ppo.yaml as it stands (2 envs, n_steps 4000, batch 100, 10 epochs, 2x64)
    "point-ref-2env-2x64": dict(D=14, A=2, H=64, N=2, T=4000, E=10, B=100, p_term=1 / 119.0, tl=1000, cpu_threads=1),
    # the headline shape with the environments on the HOST (native C goal env, pinned zero-copy staging): the
    # PCIe-inclusive rate of DESIGN.md -- never the headline `value`, which keeps its inputs resident in HBM
    "doggo-4096env-2x256-hostenv": dict(D=58, A=12, H=256, N=4096, T=1000, E=5, B=65536, p_term=1 / 107.0, tl=1000,
                                        host_env="doggo"),
    # BASELINE configs[1] with the environments on the HOST: the 2x64 networks every reference YAML trains, served by k_rollout64_tile<.., 3>
    "point-1024env-2x64-hostenv": dict(D=14, A=2, H=64, N=1024, T=2048, E=10, B=65536, p_term=1 / 119.0, tl=1000, host_env="point"),
    # data/configs/doggo-ppo.yaml as it stands (16 envs, batch 100, 2x64) with the environments on the HOST: one row range, half a tile
    "doggo-ref-16env-2x64-hostenv": dict(D=58, A=12, H=64, N=16, T=1000, E=5, B=100, p_term=1 / 107.0, tl=1000, host_env="doggo", cpu_threads=1),
    # BASELINE configs[4]: mixed fleet, ragged obs/act dims packed into one rollout arena (mobrob_amd/fleet.py)
    "fleet-car-drone-turtlebot3-2x64": dict(segments=["car", "drone", "turtlebot3"], H=64, N=1024, T=2048, E=10,
                                            B=65536, tl=1000),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 16 * PEAK_F32_MFMA_TFLOPS  # same table: the f32 matrix rate is "1/16 of BF16 MFMA" (~2.5 PF dense)
X3_PRODUCTS = 6  # bf16 x bf16 MFMAs issued per float32 multiply-add of a product computed from three-way split operands
# What the two matrix pipes SUSTAIN on this part under its 1400 W package cap (profiles/r5/mfma_power_probe.txt: a stream of nothing
# but MFMAs on every SIMD for seconds, random operands): v_mfma_f32_16x16x32_bf16 at full issue rate holds 1.88 - 1.97 GHz, not 2.4;
# v_mfma_f32 is not power limited.  Never the `peak` of the roofline object (that is the guide's figure) -- reported beside it.
SUSTAINED_BF16_MFMA_TFLOPS = 1950.7
SUSTAINED_F32_MFMA_TFLOPS = 155.0


def f_fwd(D, H, A):
    return 2 * (2 * D * H + 2 * H * H + H * A + H)


def executed_flops_per_sample(D, H, A):
    """forward + backward of both networks without the layer-1 input gradient (nothing consumes d loss / d obs)"""
    return 3.0 * f_fwd(D, H, A) - 2 * (2 * D * H)


def init_params(D, A, H, seed):
    """Orthogonal init with SB3 gains (random-init weights of the named architecture)."""
    from mobrob_amd.rl_control.init import orthogonal_policy_init
    return orthogonal_policy_init(D, A, (H, H), (H, H), seed)


def _oracle_throughput(w, budget_s, threads=None):
    """The CPU oracle (SB3-semantics NumPy restatement) on a bounded sample of workload `w`."""
    from oracle import ppo_oracle as O
    from mobrob_amd.envs.shm_vec_env import usable_cores
    limiter = None
    want = int(threads) if threads else usable_cores()  # never more BLAS threads than the cgroup grants: an
    try:                                                 # oversubscribed pool (128 threads on a 16-core quota) is SLOWER
        from threadpoolctl import threadpool_info, threadpool_limits
        limiter = threadpool_limits(limits=want)
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [want])
    except Exception:
        cores = os.cpu_count() or 1
    D, A, H, N = w["D"], w["A"], w["H"], w["N"]
    Ts = w["T"] if N * w["T"] <= 65536 else max(2, min(w["T"], int(np.ceil(w["B"] / N))))  # >= one full minibatch
    p = O.init_params(D, A, (H, H), (H, H), seed=0)
    h = O.Hyper(n_epochs=w["E"], batch_size=w["B"], ent_coef=0.01)
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    done_steps, reps = 0, 0
    while True:
        env = O.NumpySyntheticVecEnv(N, D, A, p_term=w["p_term"], time_limit=w["tl"], seed=reps)
        obs = env.reset()
        buf, _, _ = O.collect_rollout(p, env, obs, np.ones(N, bool), Ts, h,
                                      lambda t: rng.standard_normal((N, A), dtype=np.float32))
        perms = [rng.permutation(Ts * N) for _ in range(h.n_epochs)]
        O.train(p, O.AdamState.zeros_like(p), buf, h, perms)
        done_steps += Ts * N
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or el + el / reps > 1.5 * budget_s:
            break
    if limiter is not None:
        limiter.restore_original_limits()
    return {"value": done_steps / el, "unit": "env-steps/s", "cores": int(min(cores, usable_cores())), "kind": "port",
            "threads": int(cores), "cpu_quota_cores": usable_cores(),
            "sample": f"{reps} x (rollout {Ts} steps x {N} envs + {h.n_epochs} epochs, minibatch {w['B']}) "
                      f"= {done_steps} env-steps of the NumPy oracle in {el:.1f} s"}


def _torch_throughput(w, budget_s, threads=None):
    """The same bounded sample with the torch-CPU restatement (oracle/torch_baseline.py: nn.Linear / Tanh / Normal / autograd /
    clip_grad_norm_ / optim.Adam strung together in SB3's order) -- the arithmetic the reference really runs (SB3 on torch-CPU;
    /root/reference/examples/train.py:13 pins it to one thread)."""
    try:
        import torch
        from oracle import ppo_oracle as O
        from oracle import torch_baseline as TB
        from mobrob_amd.envs.shm_vec_env import usable_cores
    except Exception as ex:  # noqa: BLE001
        return {"available": False, "reason": f"{type(ex).__name__}: {ex}"}
    want = int(threads) if threads else usable_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(want)
    try:
        D, A, H, N = w["D"], w["A"], w["H"], w["N"]
        Ts = w["T"] if N * w["T"] <= 65536 else max(2, min(w["T"], int(np.ceil(w["B"] / N))))
        h = O.Hyper(n_epochs=w["E"], batch_size=w["B"], ent_coef=0.01)
        policy = TB.TorchPolicy(O.init_params(D, A, (H, H), (H, H), seed=0))
        opt = TB.make_optimizer(policy, h)
        rng = np.random.default_rng(0)
        t0 = time.perf_counter()
        done_steps, reps = 0, 0
        while True:
            env = O.NumpySyntheticVecEnv(N, D, A, p_term=w["p_term"], time_limit=w["tl"], seed=reps)
            obs = env.reset()
            buf, _, _ = TB.collect_rollout(policy, env, obs, np.ones(N, bool), Ts, h,
                                           lambda t: rng.standard_normal((N, A), dtype=np.float32))
            TB.train(policy, opt, buf, h, [rng.permutation(Ts * N) for _ in range(h.n_epochs)])
            done_steps += Ts * N
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or el + el / reps > 1.5 * budget_s:
                break
        return {"available": True, "value": done_steps / el, "unit": "env-steps/s", "cores": int(min(want, usable_cores())), "kind": "port-torch",
                "threads": int(want), "torch": torch.__version__,
                "sample": f"{reps} x (rollout {Ts} steps x {N} envs + {h.n_epochs} epochs, minibatch {w['B']}) "
                          f"= {done_steps} env-steps of the torch-CPU restatement in {el:.1f} s"}
    except Exception as ex:  # noqa: BLE001
        return {"available": False, "reason": f"{type(ex).__name__}: {ex}"}
    finally:
        torch.set_num_threads(prev)


def _sb3_throughput(w, budget_s):
    """when stable-baselines3 happens to be importable on the box, time the REAL reference stack
    (`stable_baselines3.PPO(device="cpu")`, what /root/reference/src/mobrob/rl_control/ppo.py:50-59 constructs) on a
    synthetic VecEnv of the reference shape.  Neither this container nor the GPU image ships SB3, so this normally
    reports why it is absent."""
    try:
        import gymnasium as gym
        import stable_baselines3 as sb3
        from stable_baselines3.common.vec_env import DummyVecEnv
    except Exception as ex:  # noqa: BLE001 - any import problem means "no third column"
        return {"available": False, "reason": f"{type(ex).__name__}: {ex}"}
    try:
        import torch
        torch.set_num_threads(1)  # examples/train.py:13
        D, A, H, N, T = w["D"], w["A"], w["H"], w["N"], w["T"]

        class Synthetic(gym.Env):
            observation_space = gym.spaces.Box(-np.inf, np.inf, (D,), np.float32)
            action_space = gym.spaces.Box(-1.0, 1.0, (A,), np.float32)

            def __init__(self, seed):
                self.rng, self.t = np.random.default_rng(seed), 0

            def reset(self, *, seed=None, options=None):
                self.t = 0
                return self.rng.standard_normal(D, dtype=np.float32), {}

            def step(self, action):
                self.t += 1
                term = bool(self.rng.random() < w["p_term"])
                rew = float(0.03 + 0.1 * self.rng.standard_normal() + 5.0 * term)
                return self.rng.standard_normal(D, dtype=np.float32), rew, term, (self.t >= w["tl"]) and not term, {}

        venv = DummyVecEnv([(lambda i=i: Synthetic(i)) for i in range(N)])
        model = sb3.PPO("MlpPolicy", venv, n_steps=T, batch_size=w["B"], n_epochs=w["E"], ent_coef=0.01, device="cpu",
                        policy_kwargs=dict(net_arch=dict(pi=[H, H], vf=[H, H])), verbose=0, seed=0)
        t0 = time.perf_counter()
        iters = 0
        while time.perf_counter() - t0 < budget_s:
            model.learn(total_timesteps=N * T, reset_num_timesteps=False)
            iters += 1
        el = time.perf_counter() - t0
        return {"available": True, "value": iters * N * T / el, "unit": "env-steps/s", "cores": 1, "kind": "sb3",
                "version": sb3.__version__, "sample": f"{iters} x PPO.learn({N * T}) in {el:.1f} s, DummyVecEnv of {N} synthetic envs"}
    except Exception as ex:  # noqa: BLE001
        return {"available": False, "reason": f"stable_baselines3 imported but the run failed: {type(ex).__name__}: {ex}"}


def cpu_baseline(w, budget_s=14.0, workload_name=None):
    """Contract object = BASELINE.md §3 column B2 (all host cores, the benchmarked shape); beside it column B1 (ONE
    thread, the reference's own shape data/configs/doggo-ppo.yaml + examples/train.py:13) and the SB3 probe."""
    def better(np_leg, torch_leg):
        """The faster of the two CPU restatements is the column's value; both stay in the object, `kind` says which won."""
        np_leg = dict(np_leg)
        legs = {"numpy": dict(np_leg), "torch": torch_leg}
        best = np_leg
        if torch_leg.get("available") and torch_leg["value"] > np_leg["value"]:
            best = {k: v for k, v in torch_leg.items() if k != "available"}
        best = dict(best)
        best["legs"] = legs
        return best

    out = better(_oracle_throughput(w, budget_s * 0.5, w.get("cpu_threads")), _torch_throughput(w, budget_s * 0.5, w.get("cpu_threads")))
    ref_name = "doggo-ref-16env-2x64"
    if workload_name != ref_name:
        b1 = better(_oracle_throughput(WORKLOADS[ref_name], 5.0, threads=1), _torch_throughput(WORKLOADS[ref_name], 5.0, threads=1))
        b1["shape"] = ref_name + " (16 envs x 1000 steps, minibatch 100, 2x64, 5 epochs)"
        out["reference_shape_1thread"] = b1
    out["sb3"] = _sb3_throughput(WORKLOADS[ref_name], 8.0)
    return out


def csrc_sha256():
    """Fingerprint of the kernel sources; profiles/rN/hbm_traffic_pmc.json records the one it was measured on."""
    d = os.path.join(ROOT, "mobrob_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def blended_peak(D, H, A, x3_train):
    """(peak in TFLOP/s algorithmic, share of the algorithmic flops on the bf16 pipe) for the gradient kernel of a 2xH net.
    With forward_x3 the hidden-layer products of the gradient kernel (forward of both layers, dh1, dW2, dW1 -- everything but
    the heads and the never-computed input gradient that the 3*F_fwd convention counts) are issued as X3_PRODUCTS bf16 x bf16
    MFMAs per float32 multiply-add on the bf16 pipe, the rest as v_mfma_f32: the bound is the time both pipes need at THEIR
    peaks, expressed as the algorithmic rate that time corresponds to."""
    if not x3_train:
        return PEAK_F32_MFMA_TFLOPS, 0.0
    hidden_fwd = 2.0 * (2 * D * H + 2 * H * H)            # forward of both hidden layers, both networks, per sample
    dh1 = 2.0 * (2 * H * H)                                # dh1 = dz2 . W2 of the backward pass, both networks
    dw2 = 2.0 * (2 * H * H)                                # dW2 = dz2^T . h1
    dw1 = 2.0 * (2 * D * H)                                # dW1 = dz1^T . x
    x3_share = (hidden_fwd + dh1 + dw2 + dw1) / (3.0 * f_fwd(D, H, A))
    ideal_s_per_flop = x3_share * X3_PRODUCTS / (PEAK_BF16_MFMA_TFLOPS * 1e12) + (1.0 - x3_share) / (PEAK_F32_MFMA_TFLOPS * 1e12)
    return 1.0 / ideal_s_per_flop / 1e12, x3_share


def power_capped_peak(x3_share):
    """The blended peak with the SUSTAINED rates of the two pipes in place of the nominal ones (see SUSTAINED_*_TFLOPS)."""
    return 1.0 / (x3_share * X3_PRODUCTS / SUSTAINED_BF16_MFMA_TFLOPS + (1.0 - x3_share) / SUSTAINED_F32_MFMA_TFLOPS)


def parse_smi_samples(text):
    """[(package power in W, shader clock in MHz)] from the concatenated output of `rocm-smi --showclocks --showpower` calls: one call
    prints its clock lines first, then the power line; a power line without a clock line in front of it is dropped."""
    import re
    samples, clk = [], None
    for m in re.finditer(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)|Power \(W\):\s*([0-9.]+)", text):
        if m.group(1):
            clk = int(m.group(1))
        elif clk is not None:
            samples.append((float(m.group(2)), clk))
            clk = None
    return samples


def sample_power_and_clock(run, min_samples=4):
    """Package power and shader clock (rocm-smi) while run() keeps the GPU busy: is the dominant kernel power limited?  rocm-smi
    runs in a child process with a clean environment; the samples of the busy part (>= 90 % of the highest power seen) are kept."""
    import re
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return {"error": "rocm-smi not found"}
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",) and not k.startswith(("ROCP", "ROCPROF"))}
    try:
        cap = subprocess.run([smi, "--showmaxpower"], capture_output=True, text=True, timeout=20, env=env).stdout
        m = re.search(r"Power \(W\):\s*([0-9.]+)", cap)
        cap_w = float(m.group(1)) if m else None
        with tempfile.TemporaryFile("w+") as log:
            child = subprocess.Popen(["bash", "-c", f"while :; do {smi} --showclocks --showpower; sleep 0.25; done"], stdout=log,
                                     stderr=subprocess.DEVNULL, env=env, start_new_session=True)
            try:
                run()
            finally:
                os.killpg(child.pid, signal.SIGTERM)   # the process group this function started, nothing else
                child.wait()
            log.seek(0)
            text = log.read()
    except Exception as ex:  # noqa: BLE001 -- a diagnostic leg never fails the bench
        return {"error": f"{type(ex).__name__}: {ex}"}
    samples = parse_smi_samples(text)
    if len(samples) < min_samples:
        return {"error": f"only {len(samples)} rocm-smi samples", "package_cap_w": cap_w}
    top = max(p for p, _ in samples)
    busy = [(p, c) for p, c in samples if p >= 0.9 * top]
    return {"package_cap_w": cap_w, "samples": len(samples), "busy_samples": len(busy),
            "package_power_w": statistics.median(p for p, _ in busy), "shader_clock_mhz": statistics.median(c for _, c in busy),
            "how": "rocm-smi --showclocks --showpower every ~0.6 s during extra iterations AFTER the timed region; medians of the busy samples"}


def dominant_kernel_name(H, rows_per_launch, generic, x3_train=False, chain=False, epoch_kernel=False):
    """Which gradient kernel the engine launches for this shape (engine.hip: fused64_minibatch_grad / fused_minibatch_grad)."""
    if generic:
        return "generic GEMM chain"
    if epoch_kernel:
        return ("k_epoch64 (ONE co-operative launch per epoch: per minibatch the k_split64_train gradient, the fixed-order slab reduction and "
                "clip + Adam + packs as phases behind grid barriers; `avg_launch_ms` is a whole epoch, `achieved` prices all three phases)")
    if H == 256 and x3_train and chain:
        return ("k_chain_train (minibatch forward+loss+backward, 16 rows per wave chained through registers; hidden-layer products on the "
                "bf16 pipe, split float32 operands, weights streamed through an LDS ring)")
    if H == 256:
        return ("k_fused_train<.., X3> (minibatch forward+loss+backward; hidden-layer products on the bf16 pipe, split float32 operands)"
                if x3_train else "k_fused_train (minibatch forward+loss+backward)")
    tiles = -(-rows_per_launch // 32)
    return ("k_split64_train (one workgroup per 32-row tile)" if tiles <= 64 else "k_pair64_train (two waves per tile, two per SIMD)") + ": minibatch forward+loss+backward"


def profile_rounds():
    """profiles/rN directories, newest round first"""
    pd = os.path.join(ROOT, "profiles")
    names = [d for d in os.listdir(pd) if d[:1] == "r" and d[1:].isdigit()] if os.path.isdir(pd) else []
    return sorted(names, key=lambda d: -int(d[1:]))


def measured_traffic(kernel_prefix):
    """(bytes per launch | None, note): PMC-counted HBM traffic of the dominant kernel from the newest committed
    profile -- only if that profile was taken on exactly these kernel sources; otherwise None (never a stale figure)."""
    for rnd in profile_rounds():
        tj = os.path.join(ROOT, "profiles", rnd, "hbm_traffic_pmc.json")
        if not os.path.exists(tj):
            continue
        j = json.load(open(tj))
        if j.get("csrc_sha256") != csrc_sha256():
            return None, (f"profiles/{rnd}/hbm_traffic_pmc.json was measured on other kernel sources "
                          f"(csrc_sha256 {str(j.get('csrc_sha256'))[:12]} != {csrc_sha256()[:12]}): not reported")
        k = next((v for n, v in j["kernels"].items() if n.startswith(kernel_prefix)), None)
        if k is None:
            return None, f"kernel not in profiles/{rnd}/hbm_traffic_pmc.json"
        return k["hbm_bytes_per_launch_corrected"], (f"HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                                      f"command on these kernel sources (profiles/{rnd}/hbm_traffic_pmc.json, "
                                                      f"csrc_sha256 {csrc_sha256()[:12]}); not re-measured inside this run")
    return None, "no PMC profile committed"


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as child processes (one per GPU, RCCL rendezvous
    on 127.0.0.1) BEFORE this process touches the GPU, pass rank 0's JSON line through, fail if any rank fails (the
    survivors -- possibly blocked in a collective -- are terminated, then killed, and reaped)."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    chunks = []
    os.set_blocking(procs[0].stdout.fileno(), False)  # drain rank 0's pipe while polling: a full pipe must not block it

    def drain():
        try:
            while True:
                b = procs[0].stdout.read()
                if not b:
                    return
                chunks.append(b)
        except (BlockingIOError, ValueError):
            return

    while pending and rc == 0:
        time.sleep(0.05)
        drain()
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
    if rc != 0:
        for r in pending:  # our own children, by handle
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
        print(f"bench.py: all {n} ranks reaped", file=sys.stderr)
    drain()
    line = b"".join(chunks) if rc == 0 else b""
    if rc == 0 and not line.strip():
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    os.write(1, line)
    return rc


def _fault_spec():
    """MOBROB_BENCH_FAULT="<rank>:<nth all-reduce>" (tests): that rank dies without cleanup inside its nth all-reduce
    of the timed run -- a rank killed mid-update."""
    v = os.environ.get("MOBROB_BENCH_FAULT")
    if not v:
        return None
    r, n = v.split(":")
    return int(r), int(n)


def dry_run_cpu(args, rank, world):
    """Launcher / rendezvous / timing-contract check without a GPU (tests only): gloo ranks run the barrier + the
    per-optimizer-step all-reduce of a gradient-sized buffer and print the line's skeleton.  No PPO work happens here
    and no throughput is claimed (`value` is null)."""
    import torch
    import torch.distributed as dist
    w = WORKLOADS[args.workload]
    if world > 1:
        dist.init_process_group("gloo")
    P = 2 * (w["D"] * w["H"] + w["H"] * w["H"] + 2 * w["H"]) + w["H"] * (w["A"] + 1) + 2 * w["A"] + 1
    g = torch.full((P + 8,), float(rank + 1))
    fault, calls = _fault_spec(), [0]

    def all_reduce(x):
        calls[0] += 1
        if fault and fault[0] == rank and calls[0] == fault[1]:
            os._exit(17)
        if world > 1:
            dist.all_reduce(x)

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        if world > 1:
            dist.all_reduce(g.clone())
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = g.clone()
        all_reduce(x)
        assert float(x[0]) == world * (world + 1) / 2
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({"metric": "env-steps/sec (whole node), doggo PPO", "value": None, "unit": "env-steps/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(tt) / max(args.steps, 1),
              "dry_run": "cpu/gloo launcher check: no PPO work, no throughput claim",
              "config": {"workload": args.workload, "parallelism": f"dp{world}",
                         "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
                         "n_ranks_source": "torch.distributed (gloo)"}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Job:
    """What a rank knows about the data-parallel job it is part of."""

    def __init__(self, args, rank, local_rank, world):
        self.rank, self.world = rank, world
        # MOBROB_FORCE_DP=1 runs the data-parallel code path (process group, the C loop, all-reduces) even at world
        # size 1 -- validates the multi-GPU plumbing on a single-GPU box.
        self.force_dp = os.environ.get("MOBROB_FORCE_DP", "0") == "1"
        # MOBROB_DP_SAME_DEVICE=1: REHEARSAL of an N-rank run on ONE GPU -- every rank uses device 0, the process
        # group is gloo (RCCL refuses two ranks on one device) and the C loop mobrob_ppo_train_dp exchanges through
        # the host-staged callback.  Everything but the transport of the sum is the code an N-GPU run executes; the
        # ranks time-share the GPU, so no throughput is claimed (`value` null, "rehearsal" key).
        self.same_device = os.environ.get("MOBROB_DP_SAME_DEVICE", "0") == "1" and world > 1
        self.device_id = 0 if self.same_device else local_rank
        self.use_dp = world > 1 or self.force_dp
        self.backend_name = "gloo" if self.same_device else "nccl"

    def init(self):
        if not self.use_dp:
            return
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if self.same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.device_id))

    def barrier(self):
        if self.use_dp:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(self, x):
        if not self.use_dp:
            return x
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.same_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def replicas_identical(self, flat):
        """Every rank holds the same parameter BITS (no broadcast ever happens: identical updates keep them so)."""
        if not self.use_dp:
            return True
        import torch
        import torch.distributed as dist
        digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(flat).tobytes()).digest(), np.uint8).astype(np.int64)
        mine = torch.from_numpy(digest.copy())
        if not self.same_device:
            mine = mine.cuda()
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))

    def finish(self):
        if self.use_dp:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def bench_fleet(args, name, steps, warmup, job, phases):
    """Mixed-fleet workload: every rank holds all segments (N envs each); segments overlap on per-segment streams."""
    import torch
    from mobrob_amd.fleet import MixedFleet, train_fleet_data_parallel
    from mobrob_amd.parallel import EngineBackend
    w = WORKLOADS[name]
    rank, world, use_dp = job.rank, job.world, job.use_dp
    H, N, T, E, B = w["H"], w["N"], w["T"], w["E"], w["B"]
    fleet = MixedFleet(w["segments"], n_envs=N, n_steps=T, batch_size=B * world, n_epochs=E, pi=(H, H), vf=(H, H),
                       ent_coef=0.01, seed=0, device_id=job.device_id, rank=rank, world_size=world,
                       fast_kernels=not args.generic)
    for s in fleet.segments:
        s.engine.set_params(init_params(s.obs_dim, s.act_dim, H, seed=0))
    streams, backends = [], []
    if use_dp:
        for s in fleet.segments:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                backends.append(EngineBackend(s.engine))
            streams.append(st)

    def iteration():
        fleet.collect_synthetic(time_limit=w["tl"])
        if use_dp:
            train_fleet_data_parallel(backends, streams, force_collectives=job.force_dp)
        else:
            fleet.train_enqueue()

    def fence():
        fleet.synchronize()
        torch.cuda.synchronize()
        if use_dp:
            job.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        iteration()
    fence()
    for s in fleet.segments:
        s.engine.profile(True, only=None if phases else ["train_grad"])
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    fence()
    dt = time.perf_counter() - t0
    profs = [s.engine.profile_read() for s in fleet.segments]
    dt = job.max_over_ranks(dt)
    # The fleet's honest price: the segments' gradient launches OVERLAP on per-segment streams, so a launch's
    # HIP-event duration includes the time it shares the chip with the other segments' launches -- the sum of such durations
    # counts the same wall time up to three times.  What is priced instead: the algorithmic flops of ALL segments' updates over
    # the WALL time of the update phase (every epoch of every segment: gradient kernels, reductions, clip + Adam), measured on
    # its own with the event brackets off.
    update_wall = None
    if not use_dp:
        for s in fleet.segments:
            s.engine.profile(False)
        fence()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            fleet.train_enqueue()
        fence()
        update_wall = (time.perf_counter() - t0) / reps
    out = None
    if rank == 0:
        env_steps = fleet.env_steps_per_iteration * world * steps
        ms = sum(p["train_grad"][0] for p in profs)
        calls = sum(p["train_grad"][1] for p in profs)
        flops = sum(3.0 * f_fwd(s.obs_dim, H, s.act_dim) * float(N) * T * E * steps for s in fleet.segments)
        achieved_launch = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        achieved = (flops / steps) / update_wall / 1e12 if update_wall else achieved_launch
        out = {
            "metric": "env-steps/sec (whole node)", "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name,
                       "segments": [{"robot": s.name, "obs_dim": s.obs_dim, "act_dim": s.act_dim, "envs_per_gpu": s.n_envs,
                                     "arena_offset": s.offset, "arena_bytes": s.nbytes} for s in fleet.segments],
                       "net_arch": [H, H], "n_steps": T, "n_epochs": E, "minibatch_per_gpu": B,
                       "env_source": "device-resident synthetic (Philox)", "parallelism": f"dp{world}",
                       "kernels": "generic" if args.generic else "fused"},
            "roofline": {"bound": "mfma", "kernel": "k_pair64_train, all segments",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "priced_on": ("algorithmic flops of all segments' updates / wall time of the update phase (gradient kernels + "
                                       "reductions + clip + Adam of every segment, overlapping on per-segment streams)" if update_wall else
                                       "sum of per-launch HIP-event durations (data parallel: the update phase is not timed on its own)"),
                         "update_phase_ms_per_step": 1e3 * update_wall if update_wall else None,
                         "per_launch_sum": {"achieved": achieved_launch, "frac": achieved_launch / PEAK_F32_MFMA_TFLOPS,
                                            "note": "round 2-4 accounting: per-launch durations of overlapping streams added up (time sharing counted per segment)"},
                         "avg_launch_ms": ms / max(calls, 1), "launches": calls, "flops_per_launch": flops / max(calls, 1)},
            "phases_bracketed": "all" if phases else "dominant kernel only",
            "phase_ms_per_step": {k: sum(p[k][0] for p in profs) / steps for k in profs[0]
                                  if sum(p[k][1] for p in profs) > 0},
        }
    fleet.close()
    return out


def bench_single(args, name, steps, warmup, job, phases):
    """One learner per rank (every workload but the fleet): rank 0 returns the result object, the others None."""
    import torch
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.parallel import EngineBackend, train_data_parallel
    w = WORKLOADS[name]
    rank, world, use_dp, force_dp = job.rank, job.world, job.use_dp, job.force_dp
    D, A, H, N, T, E, B = w["D"], w["A"], w["H"], w["N"], w["T"], w["E"], w["B"]
    eng = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B * world, n_epochs=E, pi=(H, H), vf=(H, H),
                    gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.01, seed=0, device_id=job.device_id,
                    rank=rank, world_size=world, fast_kernels=not args.generic, forward_x3=not getattr(args, "f32_pipe", False))
    eng.set_params(init_params(D, A, H, seed=0))  # identical replicas on every rank
    x3_mode = eng.x3_mode()
    backend = EngineBackend(eng) if use_dp else None
    fault, ar_calls = _fault_spec(), [0]
    if use_dp and job.same_device and fault and fault[0] == rank:  # tests: this rank dies inside an all-reduce
        inner = backend.gloo_all_reduce

        def faulty(group=None):
            fn = inner(group)

            def reduce_in_place(ptr, count, dtype, stream):
                ar_calls[0] += 1
                if ar_calls[0] == fault[1]:
                    os._exit(17)
                fn(ptr, count, dtype, stream)
            return reduce_in_place
        backend.gloo_all_reduce = faulty
    if use_dp:  # the engine's own RCCL communicator (the update loop runs in C: mobrob_ppo_train_dp); RCCL sets up
        # channels / peer connections lazily at the first collective of a given size, so one full update is run
        # outside the timed region whatever --warmup is (its input is a throw-away rollout)
        if not job.same_device:
            backend.ensure_comm()
        if not (fault and job.same_device):
            eng.collect_synthetic(p_term=w["p_term"], time_limit=w["tl"])
            train_data_parallel(backend, force_collectives=force_dp)
        p0 = init_params(D, A, H, seed=0)
        eng.set_params(p0)                                   # the timed region starts from the same replicas ...
        zeros = {k: np.zeros_like(v) for k, v in p0.items()}
        eng.set_optimizer_state(zeros, zeros, 0)             # ... and a fresh optimizer
        torch.cuda.synchronize()

    host = None
    if w.get("host_env"):
        from mobrob_amd.envs.native_env import NativeGoalVecEnv
        host = NativeGoalVecEnv.for_robot(w["host_env"], N, time_limit=w["tl"], seed=1000 * rank)
        hb = dict(obs=eng.pinned((N, D)), clip=eng.pinned((N, A)), rew=eng.pinned((N,)), done=eng.pinned((N,), np.uint8),
                  trunc=eng.pinned((N,), np.uint8), term=eng.pinned((N, D)))
        host.use_buffers(obs=hb["obs"], rewards=hb["rew"], dones=hb["done"], truncated=hb["trunc"], terminal_obs=hb["term"])
        if getattr(args, "host_env_threads", 0):
            host.set_threads(args.host_env_threads)
        host.reset()
        pipe = eng.part_pipeline(max(1, args.host_parts), hb["obs"], hb["clip"], hb["rew"], hb["done"], hb["trunc"], hb["term"])

    def host_rollout():
        eng.rollout_begin()
        if not args.host_python_loop:  # the whole collector loop in one native call (one row range included: served, no Python frame per step)
            pipe.collect(host.step_range_fn, host.handle)
            return
        if args.host_parts > 1:  # same pipeline driven from Python (what a Python-stepped env would use)
            for p in range(args.host_parts):
                pipe.act(p)
            for t in range(T):
                for p in range(args.host_parts):
                    pipe.wait(p)
                    nt = host.step_range(*pipe.bounds[p], hb["clip"])
                    pipe.store(p, nt > 0)
                    if t + 1 < T:
                        pipe.act(p)
        else:
            for _ in range(T):
                eng.act(hb["obs"], out_clipped=hb["clip"], want_all=False)
                nt = host.step_arrays(hb["clip"])[5]
                eng.store(hb["rew"], hb["done"], hb["trunc"] if nt else None, hb["term"] if nt else None)
        eng.finish_rollout(hb["obs"], hb["done"])

    def iteration():
        if host is not None:
            host_rollout()
        else:
            eng.collect_synthetic(p_term=w["p_term"], time_limit=w["tl"])
        if use_dp:
            train_data_parallel(backend, force_collectives=force_dp)
        else:
            eng.train(None)

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if use_dp:
            job.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        iteration()
    fence()
    # HIP events around every launch of the dominant kernel over the whole timed region (the roofline figure) and, in
    # a data-parallel run, around every all-reduce (so that a scaling loss can be attributed); the other phases only
    # with --phases: an event pair costs GPU time at every launch boundary
    eng.profile(True, only=None if phases else (["train_grad", "allreduce"] if use_dp and not args.no_allreduce_phase else ["train_grad"]))
    eng.allreduce_counters(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    ar_n, ar_bytes = eng.allreduce_counters()
    dt = job.max_over_ranks(dt)
    identical = job.replicas_identical(eng.get_flat_params())
    comm_n, _ = eng.comm_info()

    out = None
    if rank == 0:
        import torch.distributed as dist
        env_steps = N * T * world * steps
        nmb = eng.n_minibatches
        ms, calls = prof["train_grad"]
        # ALGORITHMIC flops of the dominant kernel: forward + 2x backward = 3*F_fwd per sample, times
        # the samples one launch processes (T*N*E samples per iteration spread over its launches; the last
        # minibatch of an epoch is short when batch does not divide T*N)
        flops_per_launch = 3.0 * f_fwd(D, H, A) * (float(N) * T * E * steps) / max(calls, 1)
        achieved = (flops_per_launch * calls / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
        # The peak the kernel is priced against.  With forward_x3 the forward of the two hidden layers inside the gradient
        # kernel (a third of its algorithmic flops minus the heads) is issued as six bf16 x bf16 MFMAs per float32
        # multiply-add on the bf16 pipe, everything else as v_mfma_f32: the bound is the time both pipes need at THEIR
        # peaks, expressed as the algorithmic rate that time corresponds to.
        x3_train = bool(x3_mode & 2) and not args.generic
        peak, x3_share = blended_peak(D, H, A, x3_train)
        traffic, traffic_note = None, "PMC traffic is profiled for the default workload on one GPU only"
        if not args.generic and name == "doggo-4096env-2x256" and not use_dp and not phases:
            traffic, traffic_note = measured_traffic("void mobrob::k_chain_train<" if x3_train and (x3_mode & 4) else "void mobrob::k_fused_train<64")
        if comm_n > 0:      # the engine's own communicator carried the collectives: ITS size is what took part
            ranks_seen, ranks_src = comm_n, "ncclCommCount of the engine's communicator"
        elif use_dp:
            ranks_seen, ranks_src = dist.get_world_size(), f"torch.distributed ({job.backend_name})"
        else:
            ranks_seen, ranks_src = 1, "single rank"
        value = env_steps / dt
        # which exchange carried the sums, and what its set-up self-check found (parallel.EngineBackend.choose_exchange)
        chosen = getattr(backend, "exchange", None) if use_dp else None
        exchange = {None: None,
                    "oneshot": "one-shot all-reduce over peer-mapped buffers (hipIpc, csrc/oneshot_allreduce.h), C loop",
                    "rccl": "ncclAllReduce on the engine's own RCCL communicator, C loop",
                    "gloo-callback": "host-staged gloo all-reduce callback, C loop",
                    "torch.distributed": "torch.distributed all-reduce, Python loop (fallback)"}.get(chosen, chosen)
        out = {
            "metric": "env-steps/sec (whole node), doggo PPO" if "doggo" in name else "env-steps/sec (whole node)",
            "value": None if job.same_device else value, "unit": "env-steps/s", "n_gpus": world, "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": ("float32 storage, float32 accumulation; matrix products of the forward passes (rollout, value pass, forward "
                           "inside the gradient kernel) and of the hidden layers' backward pass (dh1, dW2, dW1) as six bf16 x bf16 partial products of three-way split float32 operands "
                           "(error against float64 not larger than v_mfma_f32's: tests/test_engine_gpu.py::"
                           "test_x3_forward_kernels_are_float32_accurate), the heads and their gradients on v_mfma_f32" if x3_mode
                           else "float32 throughout (v_mfma_f32)"),
            "config": {"workload": name, "obs_dim": D, "act_dim": A, "net_arch": [H, H], "envs_per_gpu": N,
                       "n_steps": T, "n_epochs": E, "minibatch_per_gpu": B, "minibatches_per_epoch": nmb,
                       "env_source": (f"native host env (csrc/host_env.c, OpenMP), pinned zero-copy staging over PCIe, {args.host_parts} pipelined row ranges"
                                      if host is not None else "device-resident synthetic (Philox)"),
                       "parallelism": f"dp{world}", "n_ranks_seen": ranks_seen, "n_ranks_source": ranks_src,
                       "kernels": "generic" if args.generic else "fused"},
            "roofline": {"bound": "mfma", "kernel": dominant_kernel_name(H, B, args.generic, x3_train, bool(x3_mode & 4), bool(eng.update_mode() & 1)),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_note": traffic_note,
                         "peak_note": (f"blend of two matrix pipes: {100 * x3_share:.1f} % of the algorithmic flops (forward of the hidden "
                                       f"layers, their dh1 / dW2 / dW1 of the backward pass) run as {X3_PRODUCTS} bf16 MFMAs per float32 multiply-add (peak {PEAK_BF16_MFMA_TFLOPS:.0f} / "
                                       f"{X3_PRODUCTS} TFLOP/s), the rest on v_mfma_f32 (peak {PEAK_F32_MFMA_TFLOPS}); against the f32 peak alone "
                                       f"the kernel would read {achieved / PEAK_F32_MFMA_TFLOPS:.3f}" if x3_train else
                                       "every matrix product on v_mfma_f32 (peak 157.3 TFLOP/s)"),
                         "avg_launch_ms": ms / max(calls, 1), "launches": calls,
                         "flops_per_launch": flops_per_launch,
                         # the flops the kernel must EXECUTE: BASELINE.md's 3 F_fwd convention counts a layer-1 dX = dz1 . W1 that PPO
                         # never needs (2 networks x 2 D H per sample); `achieved` / `frac` keep the convention
                         "executed_flops_per_launch": executed_flops_per_sample(D, H, A) * (float(N) * T * E * steps) / max(calls, 1),
                         "executed_frac": (executed_flops_per_sample(D, H, A) / (3.0 * f_fwd(D, H, A))) * achieved / peak},
            "phases_bracketed": "all" if phases else ("dominant kernel + all-reduces" if use_dp else "dominant kernel only"),
            "phase_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1] > 0},
        }
        if use_dp:
            out["allreduces_per_step"] = ar_n / steps
            out["allreduce_bytes"] = {"per_step": ar_bytes / steps, "gradient_message": (eng.P + 8) * 4,
                                      "advantage_statistics_message": nmb * 4 * 8}
            out["replicas_bit_identical"] = identical
            out["config"]["exchange"] = exchange
            out["exchange"] = chosen          # "rccl" | "oneshot" | "torch.distributed" | "gloo-callback": read against DESIGN.md §5's table
            out["allreduce_ms_per_step"] = out["phase_ms_per_step"].get("allreduce")
            out["exchange_selfcheck"] = getattr(backend, "exchange_selfcheck", {}) or {"note": "no engine-owned exchange to check (host-staged callback)"}
        if job.same_device:
            out["rehearsal"] = {"what": f"{world} ranks time-sharing ONE GPU (MOBROB_DP_SAME_DEVICE=1): gloo process group, the C "
                                        f"loop mobrob_ppo_train_dp, sums through: {exchange}; no throughput is claimed",
                                "env_steps_per_s_time_shared": value, "replicas_bit_identical": identical}
    if out is not None and getattr(args, "power_leg", False) and not use_dp and host is None and name == "doggo-4096env-2x256":
        # Is the dominant kernel power limited?  ~3 s of further iterations (outside the timed region) with rocm-smi sampled beside
        # them, and the blended peak recomputed from what the two matrix pipes sustain under the package cap (a committed probe).
        def busy():
            for _ in range(max(8, int(3.0 * steps / max(dt, 1e-3)))):
                iteration()
            fence()
        pw = sample_power_and_clock(busy)
        if x3_train:
            capped = power_capped_peak(x3_share)
            pw.update({"power_capped_peak": capped, "frac_of_power_capped_peak": achieved / capped,
                       "power_capped_peak_note": (f"the blended peak with the rates the pipes SUSTAIN under the package cap in place of the nominal ones: "
                                                  f"bf16 MFMA {SUSTAINED_BF16_MFMA_TFLOPS} TFLOP/s (full issue rate, random operands, 1.88 - 1.97 GHz at ~1320 W), "
                                                  f"f32 MFMA {SUSTAINED_F32_MFMA_TFLOPS} (not power limited); profiles/r5/mfma_power_probe.txt, "
                                                  f"scratch/mfma_power_probe.hip -- a measured property of the part, not the roofline's `peak`")})
        if pw.get("package_power_w"):
            # launch time x package power: at the cap the launch time IS (dynamic energy) / (cap - static power) -- the ablation of
            # profiles/r5/chain_energy_ablation.txt splits the 0.33 J of k_chain_train into MFMAs 61 %, on-CU skeleton 25 %, ring DMA 10 %,
            # dW1 running sums 6 %; the clocked-but-idle chip draws ~390 W
            pw["joules_per_launch"] = pw["package_power_w"] * (ms / max(calls, 1)) * 1e-3
            pw["dynamic_joules_per_launch_above_390W"] = (pw["package_power_w"] - 390.0) * (ms / max(calls, 1)) * 1e-3
        out["roofline"]["power"] = pw
    if use_dp and not identical:
        raise SystemExit("bench.py: the replicas' parameters differ between ranks")
    if use_dp:
        backend.close()       # one-shot exchange: no rank unmaps its buffer while a peer may still read it
    eng.close()
    return out


def native_env_legs(args, job, wname, parts_sweep=(2, 4, 8, 1), launch_sweep=(2, 4)):
    """Three numbers per vector step for the native C env behind mobrob_ppo_collect_host at workload `wname` (simulator alone / GPU act +
    store alone / pipelined, served and launch-per-step collectors), and the whole PPO iteration at the best row-range count."""
    import copy
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    res = {}
    w = WORKLOADS[wname]
    D, A, H, N, T, E, B = w["D"], w["A"], w["H"], w["N"], w["T"], w["E"], w["B"]
    tag = "" if wname.startswith("doggo-4096env-2x256") else f" [{wname}]"

    def timed(fn, sync):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        return time.perf_counter() - t0

    try:
        eng = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01,
                        seed=0, device_id=job.device_id)
        eng.set_params(init_params(D, A, H, seed=0))
        host = NativeGoalVecEnv.for_robot(w["host_env"], N, time_limit=w["tl"], seed=0)
        hb = dict(obs=eng.pinned((N, D)), clip=eng.pinned((N, A)), rew=eng.pinned((N,)), done=eng.pinned((N,), np.uint8),
                  trunc=eng.pinned((N,), np.uint8), term=eng.pinned((N, D)))
        host.use_buffers(obs=hb["obs"], rewards=hb["rew"], dones=hb["done"], truncated=hb["trunc"], terminal_obs=hb["term"])
        host.reset()
        sync = eng.synchronize

        rng = np.random.default_rng(0)
        pool = [np.ascontiguousarray(rng.uniform(-1, 1, (N, A)), np.float32) for _ in range(16)]   # the sim legs step on varied actions

        def sim_alone():
            for t in range(T):
                hb["clip"][:] = pool[t & 15]
                host.step_arrays(hb["clip"])

        def gpu_alone():
            eng.rollout_begin()
            for _ in range(T):
                eng.act(hb["obs"], out_clipped=hb["clip"], want_all=False)
                eng.store(hb["rew"], hb["done"], None, None)
            eng.finish_rollout(hb["obs"], hb["done"])

        def collector(parts, served=True, c_loop=False):
            def run():
                eng.rollout_begin()
                if parts > 1 or c_loop:
                    pipe = eng.part_pipeline(parts, hb["obs"], hb["clip"], hb["rew"], hb["done"], hb["trunc"], hb["term"])
                    # served: the persistent rollout kernel serves the host env (flags in pinned memory, no launch / event per step;
                    # the default on fused engines: 256-wide x3 and 64-wide); else the launch-per-step collector (act_part / store_part per row range)
                    os.environ["MOBROB_COLLECT_SERVER"] = "1" if served else "0"
                    try:
                        pipe.collect(host.step_range_fn, host.handle)  # the whole loop in one native call, finish_rollout included
                    finally:
                        os.environ.pop("MOBROB_COLLECT_SERVER", None)
                else:
                    for _ in range(T):
                        eng.act(hb["obs"], out_clipped=hb["clip"], want_all=False)
                        nt = host.step_arrays(hb["clip"])[5]
                        eng.store(hb["rew"], hb["done"], hb["trunc"] if nt else None, hb["term"] if nt else None)
                    eng.finish_rollout(hb["obs"], hb["done"])
            return run
        # Every leg is timed INSIDE a PPO iteration -- leg, then the five-epoch update -- as (iteration - update alone): a rollout of
        # small launches does not keep the GPU's clocks up by itself, and a leg timed on a GPU that has just idled through the
        # CPU-only legs read 40 % longer than the same loop does between two updates (the only place it ever runs).
        gpu_alone()                                            # warm-up (first launches, pinned mappings)
        eng.train(None)
        reps = 2

        def in_iteration(leg):
            def run():
                for _ in range(reps):
                    leg()
                    if leg is sim_alone:                       # the simulator leg leaves no rollout to train on: the last one stays
                        eng.mark_rollout_ready()
                    eng.train(None)
            run()
            return timed(run, sync) / reps
        t_upd = in_iteration(lambda: None)
        t_sim = in_iteration(sim_alone) - t_upd
        t_gpu = in_iteration(gpu_alone) - t_upd
        sweep, sweep_launch = {}, {}
        for parts in parts_sweep:
            sweep[parts] = in_iteration(collector(parts)) - t_upd
        t_one_c = in_iteration(collector(1, c_loop=True)) - t_upd          # ONE row range through mobrob_ppo_collect_host (served where the engine can)
        for parts in launch_sweep:
            sweep_launch[parts] = in_iteration(collector(parts, served=False)) - t_upd
        best = min(sweep, key=sweep.get)
        best_t = sweep[best]
        if t_one_c < best_t:
            best, best_t = 1, t_one_c          # (bench_single's host_parts=1 is this C loop, not the Python loop)
        us = lambda t: 1e6 * t / T   # noqa: E731 - microseconds per vector step of N environments
        res["native-c-env (csrc/host_env.c), pinned zero-copy, mobrob_ppo_collect_host" + tag] = {
            "envs": N, "steps_per_rollout": T, "env_threads": host.threads,
            "us_per_vector_step": {"host_sim_alone": us(t_sim), "gpu_act_store_alone": us(t_gpu),
                                   "pipelined": {f"host_parts={k}" + (" (Python loop, whole batch per step)" if k == 1 else ""): us(v)
                                                 for k, v in sorted(sweep.items())},
                                   "one_range_c_loop (mobrob_ppo_collect_host, host_parts=1: served, nothing overlaps)": us(t_one_c),
                                   "pipelined_launch_per_step_collector": {f"host_parts={k}": us(v) for k, v in sorted(sweep_launch.items())}},
            "collector": ("host_parts >= 2: the persistent rollout kernel SERVES the host environment -- its env phase is the host's: a workgroup "
                          "writes its rows' clipped actions into the pinned buffer, raises a flag word in pinned memory and polls the host's word "
                          "for its row range (the range's first workgroup polls host memory and relays through a device word); no launch, no event, "
                          "no HIP call inside the step loop (csrc/kernels_rollout.h KIND 3, engine.hip collect_host_served); "
                          "pipelined_launch_per_step_collector = the round-4 form (two launches and an event per range and step), MOBROB_COLLECT_SERVER=0"),
            # served collector: the device's work hides under the simulator's -> overhead over the simulator alone; launch-per-step
            # collector: over the longer of ITS two legs (gpu_act_store_alone is that collector's GPU leg: act + store launches per step)
            "collector_overhead_us_per_step": {
                "served (pipelined - host_sim_alone)": us(best_t) - us(t_sim),
                "launch_per_step (pipelined - longer leg)": (us(min(sweep_launch.values())) - max(us(t_sim), us(t_gpu))) if sweep_launch else None},
            "rollout_only_env_steps_per_s": N * T / best_t, "best_host_parts": best,
            "update_ms": 1e3 * t_upd,
            "note": ("every leg = (leg + update) - update alone, inside PPO iterations (GPU clocks as in the real loop); overhead = what the "
                     "hand-off itself costs once simulator and policy overlap"),
        }
        host.close()
        eng.close()
        a2 = copy.copy(args)
        a2.host_parts = best
        o = bench_single(a2, wname, 2, 1, job, False)
        res[wname] = {
            "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": 2, "warmup": 1, "host_parts": best,
            "what": "whole PPO iteration (host rollout + GAE + all epochs) with the environments on the host: the PCIe-inclusive rate, never the headline value",
            "roofline": {k: o["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches")}}
    except Exception as ex:  # noqa: BLE001 - a side measurement must not take the headline line down
        res["native-c-env" + tag] = {"error": f"{type(ex).__name__}: {ex}"}

    return res


def host_path_measurements(args, job):
    """north_star's own data path in front of the driver: host-core environments -> pinned staging -> GPU policy ->
    actions back, at the headline shape (doggo 58 / 12, 4096 envs, 2x256).  Reference: the VecEnv loop SB3 drives through
    /root/reference/src/mobrob/rl_control/ppo.py:30-48 over EnvWrapper.step (/root/reference/src/mobrob/envs/wrapper.py:156-201).

    Three numbers per collector, per vector step of all environments, so that collector overhead is a figure, not a guess:
      host_sim_alone   the environments stepped with fixed actions, no GPU work
      gpu_alone        act + store on static staging buffers, no environment stepped
      pipelined        the collector as it runs (mobrob_ppo_collect_host for the native env, PPO's part pipeline for ShmVecEnv)
    and, for the native env, the `host_parts` sweep 1 / 2 / 4 / 8 plus the whole iteration (rollout + update) at the best setting."""
    import torch
    res = {}
    w = WORKLOADS["doggo-4096env-2x256-hostenv"]
    D, A, H, N, T, E, B = w["D"], w["A"], w["H"], w["N"], w["T"], w["E"], w["B"]

    def timed(fn, sync):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        return time.perf_counter() - t0

    # ---- (1) native C env (csrc/host_env.c, OpenMP), pinned zero-copy staging: the headline shape, then BASELINE configs[1]'s ----
    res.update(native_env_legs(args, job, "doggo-4096env-2x256-hostenv"))
    res.update(native_env_legs(args, job, "point-1024env-2x64-hostenv", parts_sweep=(2, 4, 1), launch_sweep=(2,)))
    res.update(native_env_legs(args, job, "doggo-ref-16env-2x64-hostenv", parts_sweep=(1,), launch_sweep=()))
    # ---- (2) ShmVecEnv (`vec_env_type: subproc`): Python EnvWrapper instances in worker processes over a GPU-registered block ----
    try:
        from mobrob_amd.rl_control.ppo import PPOCtrl, BaseCallback
        Ts = 32                                              # a bounded sample: 4096 Python envs take milliseconds per vector step
        cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": Ts, "batch_size": B, "n_epochs": E, "gamma": 0.99, "gae_lambda": 0.95,
                              "ent_coef": 0.01, "clip_range": 0.2, "policy_kwargs": {"net_arch": {"pi": [H, H], "vf": [H, H]}}},
               "env_name": "doggo", "time_limit": w["tl"], "n_envs": N, "vec_env_type": "subproc", "enable_gui": False, "seed": 0}
        ppo = PPOCtrl.from_config(cfg).ppo
        env, e2 = ppo.env, ppo.engine
        ppo.learn(total_timesteps=N * Ts)                   # warm-up iteration: staging registered, workers running
        b = ppo._host_bufs
        cb = BaseCallback()
        cb.init_callback(ppo)

        rng = np.random.default_rng(0)
        pool = [np.ascontiguousarray(rng.uniform(-1, 1, (N, A)), np.float32) for _ in range(16)]
        parts2 = ppo.host_parts
        bounds = [(N * q // parts2, N * (q + 1) // parts2) for q in range(parts2)]

        def shm_sim():   # the same commands the pipelined collector sends (one per row range), no GPU work between them
            for t in range(Ts):
                b["clip"][:] = pool[t & 15]
                for i0, i1 in bounds:
                    env.step_range(i0, i1, b["clip"])

        def shm_gpu():
            e2.rollout_begin()
            for _ in range(Ts):
                e2.act(b["obs"], out_clipped=b["clip"], want_all=False)
                e2.store(b["rew"], b["done"], None, None)
            e2.finish_rollout(b["obs"], b["done"])
        t_sim = min(timed(shm_sim, e2.synchronize) for _ in range(2))
        t_gpu = min(timed(shm_gpu, e2.synchronize) for _ in range(2))
        t_pipe = min(timed(lambda: ppo._collect_rollouts(cb), e2.synchronize) for _ in range(2))
        t0 = time.perf_counter()
        ppo.learn(total_timesteps=2 * N * Ts, reset_num_timesteps=False)
        e2.synchronize()
        t_learn = (time.perf_counter() - t0) / 2
        us2 = lambda t: 1e6 * t / Ts   # noqa: E731
        res["ShmVecEnv (vec_env_type: subproc), 4096 Python EnvWrapper instances"] = {
            "envs": N, "steps_per_rollout": Ts, "workers": getattr(env, "n_workers", None), "host_parts": ppo.host_parts,
            "us_per_vector_step": {"host_sim_alone": us2(t_sim), "gpu_act_store_alone": us2(t_gpu), "pipelined": us2(t_pipe)},
            # (the simulator-alone leg -- the same per-range commands, uniform random actions, nothing between them -- has read
            #  LONGER than the pipelined rollout on every box: the Python environments' step time depends on what they are doing
            #  and the workers' wake-ups on the command cadence; an "overhead" is quoted only when the legs bracket the pipeline)
            "collector_overhead_us_per_step": (us2(t_pipe) - max(us2(t_sim), us2(t_gpu))) if t_pipe >= max(t_sim, t_gpu) else None,
            "rollout_only_env_steps_per_s": N * Ts / t_pipe,
            "learn_env_steps_per_s": N * Ts / t_learn,
            "note": "kinematic stand-in behind the EnvWrapper surface (no MuJoCo / Bullet on the box); bounded sample of 32 vector steps per rollout",
        }
        env.close()
        e2.close()
    except Exception as ex:  # noqa: BLE001
        res["ShmVecEnv"] = {"error": f"{type(ex).__name__}: {ex}"}
    return res


def also_measured(args, job):
    """The other BASELINE configurations that fit one GPU, three steps each with the accounting of `value` (config 2,
    the config-5 fleet, and the reference's own doggo YAML shape): the driver's record carries them too."""
    res = {}
    for name in ("point-1024env-2x64", "fleet-car-drone-turtlebot3-2x64", "doggo-ref-16env-2x64"):
        fn = bench_fleet if "segments" in WORKLOADS[name] else bench_single
        try:
            o = fn(args, name, 3, 1, job, False)
            res[name] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": 3, "warmup": 1,
                         "roofline": {k: o["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches")},
                         "config": {k: v for k, v in o["config"].items() if k in ("envs_per_gpu", "n_steps", "n_epochs", "minibatch_per_gpu", "net_arch", "obs_dim", "act_dim")}}
        except Exception as ex:  # noqa: BLE001 - the headline line must not die with a side measurement
            res[name] = {"error": f"{type(ex).__name__}: {ex}"}
    # the headline workload with EVERY matrix product on v_mfma_f32 (forward_x3 off): what the split-bf16 forward passes buy
    try:
        import copy
        a32 = copy.copy(args)
        a32.f32_pipe = True
        o = bench_single(a32, "doggo-4096env-2x256", 3, 1, job, False)
        res["doggo-4096env-2x256, forward_x3 off (all products on v_mfma_f32)"] = {
            "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": 3, "warmup": 1,
            "roofline": {k: o["roofline"].get(k) for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "launches", "power")}}
    except Exception as ex:  # noqa: BLE001
        res["doggo-4096env-2x256, forward_x3 off (all products on v_mfma_f32)"] = {"error": f"{type(ex).__name__}: {ex}"}
    # the headline workload through the GENERIC GEMM chain (what every net_arch / activation_fn / use_sde outside the fused families
    # runs on): one step is ~0.6 s
    key = "doggo-4096env-2x256, generic GEMM chain (fast_kernels off)"
    try:
        import copy
        ag = copy.copy(args)
        ag.generic = True
        o = bench_single(ag, "doggo-4096env-2x256", 1, 1, job, False)
        res[key] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": 1, "warmup": 1}
    except Exception as ex:  # noqa: BLE001
        res[key] = {"error": f"{type(ex).__name__}: {ex}"}
    if not getattr(args, "no_host_path", False):
        res["host_env_streaming_path"] = host_path_measurements(args, job)
    return res


_RESULT_FD = None


def emit(obj):
    """The ONE JSON line of the contract, written to the process's original stdout."""
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (json.dumps(obj) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="doggo-4096env-2x256", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the also_measured side configurations")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock samples (roofline.power; ~4 s after the timed region)")
    ap.add_argument("--no-host-path", action="store_true", help="also_measured: skip the host-env streaming path (native C env, ShmVecEnv)")
    ap.add_argument("--only-host-path", action="store_true", help="print only the host-env streaming-path measurements (one JSON line)")
    ap.add_argument("--generic", action="store_true", help="force the generic (unfused) kernels")
    ap.add_argument("--f32-pipe", action="store_true",
                    help="forward_x3 off: every matrix product on v_mfma_f32 (default: the forward passes of 256-wide nets run as six "
                         "bf16 products of three-way split float32 operands)")
    ap.add_argument("--phases", action="store_true",
                    help="bracket every phase with HIP events (phase_ms_per_step; costs ~4 %% of the throughput); "
                         "by default only the dominant kernel is bracketed")
    ap.add_argument("--no-allreduce-phase", action="store_true",
                    help="data-parallel runs: do not bracket the all-reduces with HIP events (640 event records per iteration "
                         "at the headline shape)")
    ap.add_argument("--host-python-loop", action="store_true",
                    help="host-env workloads: drive the pipelined rollout from Python instead of mobrob_ppo_collect_host")
    ap.add_argument("--host-env-threads", type=int, default=0, help="host-env workloads: OpenMP team of the native env (0: its default)")
    ap.add_argument("--host-parts", type=int, default=2,
                    help="host-env workloads: row ranges of the pipelined rollout (1 = whole batch per step)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="tests: exercise the rank launcher / rendezvous / one-line contract with gloo on CPU (no PPO work)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # not under torchrun: start the ranks ourselves
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # (no GPU call has happened in this process)
    # RCCL prints a version banner to the C stdout of every rank (flushed at exit, i.e. after the result line):
    # keep the real stdout for the JSON line only and send everything else written to fd 1 to stderr.
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    w = WORKLOADS[args.workload]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    job = Job(args, rank, local_rank, world)
    torch.cuda.set_device(job.device_id)
    job.init()

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    job.barrier()

    if args.only_host_path:
        if rank == 0:
            emit({"host_env_streaming_path": host_path_measurements(args, job)})
        job.finish()
        return
    fn = bench_fleet if "segments" in w else bench_single
    args.power_leg = world == 1 and not args.no_power and not args.phases and not args.generic
    out = fn(args, args.workload, args.steps, args.warmup, job, args.phases)
    if rank == 0:
        if world == 1 and not job.use_dp and args.workload == "doggo-4096env-2x256" and not args.no_also and not args.generic:
            out["also_measured"] = also_measured(args, job)
        if not args.no_cpu_baseline and world == 1 and "segments" not in w:  # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(w, workload_name=args.workload)
        emit(out)
    job.finish()


if __name__ == "__main__":
    main()
