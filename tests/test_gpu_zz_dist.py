"""Two ranks of the facade ranking_eval on ONE GPU (gloo for the collectives; on a multi-GPU node the same code runs over
RCCL with one rank per GPU): every rank builds the candidate sets of ITS user range -- on the host and through the
device builder, row0 > 0 on rank 1 -- and an engine over its rows only, and every rank returns the reference's per-user
values and means (tools/dist_facade_gpu.py asserts them to 1e-6 on every user)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_the_users_and_agree_with_the_reference():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "dist_facade_gpu.py")]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    text = out.stdout                                   # (the two ranks' lines may interleave: count the messages)
    assert text.count(": ok,") == 8, text[-2000:]
    assert text.count("device-built") == 4 and text.count("host-built") == 4 and "from row 288" in text
