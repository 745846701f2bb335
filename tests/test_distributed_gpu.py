"""GPU: the RCCL leg of bench.py on a one-GPU box -- a single-rank process group (GMD_BENCH_FORCE_DIST=1) drives
init_process_group("nccl"), both broadcasts, the barrier and the max-reduce; the sharded run must give bit-identical
HDR codes to the plain run (sharding must not change per-sample results)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--unet", "tiny", "--res", "128", "--batch", "2", "--inference-steps", "4", "--steps", "1", "--warmup", "0",
        "--no-cpu-baseline", "--no-kernel-timing", "--no-drift", "--checksum"]


def _run(env_extra, args):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


# batch 2: the small case; batch 8 + bf16: the per-rank share of BASELINE config 3 (--global-batch 64 over 8 GPUs) in the dtype
# the hidden states are broadcast in (rounded on rank 0, not by the UNet's prepare_context)
@pytest.mark.parametrize("extra", [[], ["--batch", "8", "--dtype", "bf16"], ["--global-batch", "8", "--dtype", "f16"]])
def test_single_rank_rccl_run_equals_plain_run(extra):
    args = ARGS + extra
    plain = _run({}, args)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    sharded = _run({"GMD_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port)}, args)
    assert sharded["config"]["per_gpu_batch"] == (8 if extra else 2)
    assert plain["config"]["rccl_world_size"] is None and sharded["config"]["rccl_world_size"] == 1
    assert plain["outputs_finite"] and sharded["outputs_finite"]
    assert plain["output_sha256"] == sharded["output_sha256"]


def test_single_rank_rccl_run_at_sd15_width_equals_plain_run():
    """The SD-1.5-width models through the RCCL leg once on hardware (round-3 review: the group had only ever seen the tiny UNets):
    full-width synthetic weights, init_process_group("nccl"), the bf16 hidden-state broadcast at cross-attention width 768, the
    float32 latent broadcast, barrier and max-reduce; 256x256, batch 2, 2 PNDM steps -> HDR codes bit-identical to the plain run."""
    args = ["--unet", "sd15", "--res", "256", "--batch", "2", "--inference-steps", "2", "--steps", "1", "--warmup", "0", "--dtype", "bf16",
            "--no-cpu-baseline", "--no-kernel-timing", "--no-drift", "--no-tolerance-path", "--checksum"]
    plain = _run({}, args)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    sharded = _run({"GMD_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port)}, args)
    assert "SD-v1-5" in sharded["config"]["workload"] and sharded["config"]["per_gpu_batch"] == 2
    assert plain["config"]["rccl_world_size"] is None and sharded["config"]["rccl_world_size"] == 1
    assert plain["outputs_finite"] and sharded["outputs_finite"]
    assert plain["output_sha256"] == sharded["output_sha256"]
