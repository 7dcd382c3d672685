"""world_size-2 gloo test of the multi-GPU driver logic (sharding by windows, MAX-reduce of the wall time,
all-gather of the per-rank timing records) — SURVEY.md §8e.  No GPU needed."""
import json
import os
import socket
import subprocess
import sys

from okvis_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_helpers():
    a, b = D.shard_seeds(0, 2, 4), D.shard_seeds(1, 2, 4)
    assert len(set(a) | set(b)) == 8 and not (set(a) & set(b))
    parts = [D.shard_windows(64, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == list(range(64)) and all(len(p) == 8 for p in parts)
    assert D.max_over_ranks(None, 1.5) == 1.5 and D.gather_records(None, [1, 2]) == [[1.0, 2.0]]


def test_two_ranks_gloo(tmp_path):
    out = tmp_path / "r.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), "3"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, timeout=300, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = json.load(open(out))
    assert r["world"] == 2
    assert r["wall"] == 0.75                                   # MAX over ranks
    recs = sorted(r["records"])
    assert [x[0] for x in recs] == [0.0, 1.0] and all(x[1] == 3.0 for x in recs)
    assert recs[0][3] == 0.5 and recs[1][3] == 0.75
    assert r["seeds0"] == D.shard_seeds(0, 2, 3)
    # configs[3]-style job: every window exactly once, its record carried by the one all-gather
    wr = sorted(r["window_records"])
    assert [int(x[0]) for x in wr] == list(range(7))
    assert all(x[1] == 5.0 and x[2] == 100.0 + x[0] for x in wr)
    assert {x[3] for x in wr} == {0.5, 0.75}                    # rank 0 / rank 1 batch seconds
