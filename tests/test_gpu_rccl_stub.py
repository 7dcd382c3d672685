"""GPU: the C++ record all-gather (okvis_amd/csrc/dist_capi.hip: okvis_ba_gather_records, okvis_ba_batch_run_gathered) with MORE THAN
ONE rank on a one-GPU box.  The real librccl.so cannot form a communicator of two ranks on one device, so the ranks bind
tests/stub_rccl/libstub_rccl.so (OKVIS_BA_RCCL_LIB; the four nccl entry points over a Unix socket, device buffers staged with
hipMemcpy): everything on OUR side of the nccl calls — the id-file hand-over, rank order, padding records, the device staging, the
time-outs — runs as it will on eight GPUs (window i on rank i mod G, one all-gather of 24-byte records).
Test infrastructure; the product loads librccl.so."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "stub_rccl")
STUB = os.path.join(STUB_DIR, "libstub_rccl.so")


@pytest.fixture(scope="module")
def stub():
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-s", "-C", STUB_DIR])
    return STUB


_GATHER = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, sys.argv[1])
from okvis_amd import _lib
from okvis_amd.dist import WindowRecordC
rank, world, idf, n, timeout = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), float(sys.argv[6])
L = _lib.lib()
L.okvis_ba_gather_records.argtypes = [C.c_int32, C.c_int32, C.c_int, C.c_char_p, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]
mine, out = (WindowRecordC * n)(), (WindowRecordC * (n * world))()
for i in range(n):
    last = (i == n - 1 and rank == world - 1)          # the last rank pads its last slot (window_id 0xffffffff)
    mine[i].window_id = 0xffffffff if last else rank + world * i
    mine[i].iterations = 0 if last else 10 + rank
    mine[i].final_cost = 0.0 if last else 1000.0 * rank + i + 0.25
    mine[i].seconds = 0.0 if last else 1e-3 * (rank + 1)
rc = L.okvis_ba_gather_records(rank, world, 0, os.fsencode(idf), timeout, mine, n, out)
print(json.dumps({"rc": rc, "records": [[r.window_id, r.iterations, r.final_cost, r.seconds] for r in out] if rc == 0 else []}))
'''


def _spawn(code, args, stub, **env):
    e = dict(os.environ, OKVIS_BA_RCCL_LIB=stub, **{k: str(v) for k, v in env.items()})
    return subprocess.Popen([sys.executable, "-c", code, ROOT] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def _result(p, timeout=120):
    import json
    out, err = p.communicate(timeout=timeout)
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert lines, (out[-2000:], err[-2000:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("world", [2, 3])
def test_gather_records_in_rank_order_with_padding(stub, tmp_path, world):
    idf = tmp_path / "nccl_id"
    n = 3
    ps = [_spawn(_GATHER, [r, world, idf, n, 30.0], stub) for r in reversed(range(world))]   # (rank 0 started last: the others wait for its id)
    res = [_result(p) for p in ps][::-1]
    for r in range(world):
        assert res[r]["rc"] == 0, res[r]
        assert res[r]["records"] == res[0]["records"]                    # every rank holds the same table
    recs = res[0]["records"]
    assert len(recs) == n * world
    for r in range(world):
        for i in range(n):
            rec = recs[r * n + i]                                          # rank-major: rank r's records at [r * n, (r + 1) * n)
            if r == world - 1 and i == n - 1:
                assert rec == [0xffffffff, 0, 0.0, 0.0]
            else:
                assert rec == [r + world * i, 10 + r, 1000.0 * r + i + 0.25, 1e-3 * (r + 1)]
    assert not idf.exists()                                                # rank 0 removed the id once the communicator stood


def test_a_stale_id_file_is_replaced_and_a_missing_rank_times_out(stub, tmp_path):
    idf = tmp_path / "nccl_id"
    idf.write_bytes(b"\x55" * 128)                                         # a crashed job's left-over
    p0 = _spawn(_GATHER, [0, 2, idf, 2, 30.0], stub)
    t0 = time.time()
    while idf.exists() and idf.read_bytes() == b"\x55" * 128 and time.time() - t0 < 20:   # rank 0 clears it before it publishes its own
        time.sleep(0.01)
    p1 = _spawn(_GATHER, [1, 2, idf, 2, 30.0], stub)
    r0, r1 = _result(p0), _result(p1)
    assert r0["rc"] == 0 and r1["rc"] == 0 and r0["records"] == r1["records"] and len(r0["records"]) == 4
    # rank 1 never comes: rank 0 gives up inside the communicator set-up (the stub's bound, like NCCL_TIMEOUT), reports a state
    # error and leaves no id behind
    p0 = _spawn(_GATHER, [0, 2, idf, 2, 30.0], stub, STUB_RCCL_TIMEOUT_S=1.0)
    r0 = _result(p0)
    assert r0["rc"] == -2 and not idf.exists()
    # rank 0 never comes: rank 1 gives up after ITS time-out (the id file never appears)
    t0 = time.time()
    r1 = _result(_spawn(_GATHER, [1, 2, idf, 2, 0.5], stub))
    assert r1["rc"] == -2 and 0.4 < time.time() - t0 < 60


_RUN = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
from okvis_amd import dist as D, synthetic
from okvis_amd.window import default_options
rank, world, idf, n = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
ws = [synthetic.small_window(seed=70 + i, K=4, L=40) for i in range(n)]
recs = D.batch_run_gathered(ws, rank, world, 0, 5, idf, default_options())
print(json.dumps({"rc": 0, "records": [[int(r[0]), int(r[1]), float(r[2]), float(r[3])] for r in recs]}))
'''


def test_batch_run_gathered_with_three_ranks_on_one_gpu(stub, tmp_path, oracle):
    """okvis_ba_batch_run_gathered with world = 3 and five windows: window i runs on rank i mod 3, ranks 1 and 2 pad their second
    slot..., every rank returns all five records in WINDOW order, equal to a one-rank run of the same five windows"""
    from okvis_amd import solver, synthetic
    from okvis_amd.window import default_options
    idf = tmp_path / "nccl_id"
    n, world = 5, 3
    ps = [_spawn(_RUN, [r, world, idf, n], stub) for r in reversed(range(world))]
    res = [_result(p, 300) for p in ps][::-1]
    ws = [synthetic.small_window(seed=70 + i, K=4, L=40) for i in range(n)]
    b = solver.WindowBatch(ws, options=default_options())
    sg = b.optimize(5)
    b.close()
    for r in range(world):
        recs = res[r]["records"]
        assert [x[0] for x in recs] == list(range(n))
        for i in range(n):
            assert recs[i][1] == sg[i]["iterations"] and abs(recs[i][2] - sg[i]["final_cost"]) <= 1e-10 * sg[i]["final_cost"] and recs[i][3] > 0
        assert [x[:3] for x in recs] == [x[:3] for x in res[0]["records"]]
