"""Worker of tests/test_distributed_cpu.py: one process per (fake) GPU, gloo backend, no GPU needed."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import dist as D, solver, synthetic  # noqa: E402


def main(out_path, windows_per_gpu):
    r = D.Rank.from_env()
    dist = D.init("gloo")
    seeds = D.shard_seeds(r.rank, r.world, windows_per_gpu)
    wins = [synthetic.small_window(seed=s, K=3, L=20) for s in seeds]
    stats = [solver.check_window(w) for w in wins]           # host-side structure building, no device
    seconds = 0.5 + 0.25 * r.rank                            # fake per-rank wall time
    D.barrier(dist)
    wall = D.max_over_ranks(dist, seconds)
    recs = D.gather_records(dist, [r.rank, len(wins), 10.0, seconds, float(sum(s["D"] for s in stats))])
    # strong-scaling job (BASELINE configs[3]): 7 windows in total, window i on rank i mod world, ONE all-gather of the
    # per-window records {window_id, iterations, final_cost, seconds} padded to the largest share
    ids = D.shard_windows(7, r.rank, r.world)
    n_rec = (7 + r.world - 1) // r.world
    rec = []
    for k in range(n_rec):
        rec += [float(ids[k]), 5.0, 100.0 + ids[k], seconds] if k < len(ids) else [-1.0, 0.0, 0.0, 0.0]
    g = D.gather_records(dist, rec)
    window_records = [x[4 * k:4 * k + 4] for x in g for k in range(n_rec) if x[4 * k] >= 0]
    if r.rank == 0:
        json.dump({"wall": wall, "records": recs, "world": r.world, "seeds0": seeds, "window_records": window_records},
                  open(out_path, "w"))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
