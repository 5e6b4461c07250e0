"""Records which vector math library stood behind `torch.sin / torch.cos / torch.atan2` when the fixtures of this directory were
generated: sha256 of their fp32 results over a fixed grid (host_libm_canary.json).  The appended reference headings of the
veh3dofconti fixtures amplify the last bit of those functions ~1e3 times (gops/env/env_ocp/resources/ref_traj_model.py:144-148), so
bit-level statements about them (tests/test_ref_traj_host_cpu.py) are made only on a host whose canary matches.

    python tests/golden/make_libm_canary.py        # run in the container that ran make_golden.py
"""
import hashlib
import json
import os

import torch


def canary() -> dict:
    x = torch.linspace(0.0, 40.0, 1 << 20, dtype=torch.float32)
    y = torch.linspace(-1e-3, 1e-3, 1 << 20, dtype=torch.float32)
    h = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()   # noqa: E731
    return {"sin": h(torch.sin(x)), "cos": h(torch.cos(x)), "atan2": h(torch.atan2(y, 5e-3 + 0 * y))}


if __name__ == "__main__":
    out = dict(canary(), torch=torch.__version__, cpu_capability=torch.backends.cpu.get_cpu_capability())
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_libm_canary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)
