"""The one-shot all-reduce (csrc/comm.hip) with TWO processes sharing the one GPU of the test box: same hipIpc
handle exchange, same flag protocol, same kernels as on a multi-GPU node (peer memory then sits behind xGMI instead
of the local HBM).  Runs the helper that the engine also uses to validate the path before enabling it."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("proto", ["granule", "flag"])
def test_one_shot_allreduce_two_processes_one_gpu(proto):
    """proto: small messages as data-tagged granules (round 4, the default) or everything on the stage -> flag -> peer-read
    protocol.  The helper interleaves plain, fused and gather
    calls of many sizes eagerly and under hipGraph replay, with one rank delayed, and compares bit for bit; then 300 rounds of
    randomly sized calls in random order (SSD_AR_STRESS) on both sides of the granule / flag size boundary."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 29700 + os.getpid() % 200 + {"granule": 0, "flag": 211}[proto]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", SSD_AR_DEVICE="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, SSD_AR_PROTO=proto, SSD_AR_STRESS="300")
        procs.append(subprocess.Popen([sys.executable, "-m", "ssd_amd.utils.custom_ar"], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} exit {p.returncode}:\n{out[-3000:]}"
