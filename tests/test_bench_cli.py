"""bench.py's launcher logic that can be checked without a GPU: a multi-GPU request on a box with fewer devices must fail
loudly instead of running one rank and reporting it as the requested job."""
import os
import subprocess
import sys

import torch

from conftest import ROOT


def test_multi_gpu_request_without_devices_fails_loudly():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                                            # a real multi-GPU box: the request is legitimate there
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout


def test_world_size_must_match_the_request():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
