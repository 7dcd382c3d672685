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
    if r.rank == 0:
        json.dump({"wall": wall, "records": recs, "world": r.world, "seeds0": seeds}, open(out_path, "w"))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
