"""Multi-process sharding on CPU (gloo, world sizes 2, 3 and 8): every rank ends up with the
detections of every image, bit-identical to a single process over the whole batch."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (8, 256) and (8, 250) are the partition of the eight-GPU configuration (256 images, 32 per
# rank) and an uneven variant of it
@pytest.mark.parametrize("world,batch", [(2, 6), (2, 5), (3, 7), (8, 256), (8, 250)])
def test_sharded_detections_all_gather(tmp_path, world, batch):
    out = str(tmp_path / "result")
    port = str(free_port())
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out, str(batch)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for rank, p in enumerate(procs):
        assert p.returncode == 0, logs[rank]
        assert open("%s.%d" % (out, rank)).read() == "ok", logs[rank]
