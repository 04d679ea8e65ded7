#!/usr/bin/env python3
"""Generate tests/golden/reference_counters.json from the reference checkpoints (run in the BUILD container only).

Reference-HELD outputs of SB3's learn loop (`OnPolicyAlgorithm.learn` / `PPO.train`, reached through
/root/reference/src/mobrob/rl_control/ppo.py:73-74 from /root/reference/examples/train.py:42-46): the counters each
`data/policies/<env>-ppo.zip` was saved with -- `num_timesteps`, `_total_timesteps`, `_n_updates`,
`_current_progress_remaining` (the `data` JSON) and Adam's `step` (`policy.optimizer.pth`) -- next to the hyper-parameters that
produced them (`n_steps`, `n_envs`, `batch_size`, `n_epochs`).  They pin the loop's BOOKKEEPING:
what counts as a timestep, when the loop stops (drone overshoots its 1 000 000 by one rollout), that `_n_updates` counts epochs
and Adam counts minibatches, and that `_current_progress_remaining` is the value of the last `train()` (doggo was saved by the
CheckpointCallback during the collection of rollout 1875, before that rollout's update).  Data only: five small dicts."""
import io
import json
import os
import zipfile

import torch

REF = "/root/reference/data/policies"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_counters.json")
KEYS = ["num_timesteps", "_total_timesteps", "_n_updates", "n_steps", "n_envs", "batch_size", "n_epochs",
        "_num_timesteps_at_start", "_current_progress_remaining"]
out = {}
for env in ("point", "car", "doggo", "drone", "turtlebot3"):
    zf = zipfile.ZipFile(os.path.join(REF, f"{env}-ppo.zip"))
    d = json.loads(zf.read("data"))
    opt = torch.load(io.BytesIO(zf.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
    steps = {int(s["step"]) for s in opt["state"].values()}
    assert len(steps) == 1
    out[env] = {k: d[k] for k in KEYS}
    out[env]["adam_step"] = steps.pop()
json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
print(open(OUT).read())
