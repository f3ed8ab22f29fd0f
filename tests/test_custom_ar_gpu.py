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


def test_one_shot_allreduce_two_processes_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 29700 + os.getpid() % 200
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", SSD_AR_DEVICE="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
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
