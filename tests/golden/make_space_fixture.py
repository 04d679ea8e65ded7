"""Extract the `observation_space` / `action_space` entries of the five reference checkpoints
(/root/reference/data/policies/<env>-ppo.zip, member `data`) into tests/golden/reference_spaces.json.

These are reference-HELD bytes: gymnasium 0.28.1 `Box` objects pickled by value by SB3 2.0.0's `PPO.save`
(/root/reference/src/mobrob/rl_control/ppo.py:76-77), plus the str() of every state item SB3 writes next to the
pickle.  They pin what the checkpoint writer must emit for each robot (bounds included: the drone and turtlebot3
spaces are finite).  Run in the build container only:  python tests/golden/make_space_fixture.py
"""
import json
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/policies"


def main():
    out = {}
    for env in ("point", "car", "doggo", "drone", "turtlebot3"):
        d = json.loads(zipfile.ZipFile(f"{REF}/{env}-ppo.zip").read("data"))
        out[env] = {k: d[k] for k in ("observation_space", "action_space")}
    with open(os.path.join(HERE, "reference_spaces.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
