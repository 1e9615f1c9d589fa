"""Stand-in for bench.py in tests/test_launch_cpu.py: same launch protocol (multigridsolver_amd/launch.py), no GPU.
argv: <ok_from_generation> <mode: exit|hang|crash_after_done|die_before_print>.  A worker of an earlier generation fails the way `mode` says;
from generation `ok_from` on the ranks meet in a gloo process group, all-reduce their ranks and rank 0 prints one JSON line."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multigridsolver_amd import launch  # noqa: E402


def main():
    ok_from, mode = int(sys.argv[1]), sys.argv[2]
    if os.environ.get("MGS_BENCH_WORKER") != "1":
        world_env = int(os.environ.get("WORLD_SIZE", "1"))
        if world_env > 1:
            sys.exit(launch.supervise_rank(sys.argv))
        sys.exit(launch.spawn_ranks(sys.argv, int(sys.argv[3])))
    gen, rank = int(os.environ["MGS_BENCH_GEN"]), int(os.environ["RANK"])
    if gen >= ok_from:
        import torch                       # before the watchdog starts, as in bench.py: the first import on a cold machine can take longer than the tests' 4 s period
        import torch.distributed as dist
    wd = launch.Watchdog()
    if gen < ok_from:
        if mode == "hang":
            if rank == 1:
                time.sleep(3600)          # the watchdog must end this
            sys.exit(launch.EXIT_RETRY)
        sys.exit(3 if rank == 0 else launch.EXIT_RETRY)
    wd.beat("init")
    launch.init_process_group("gloo")
    t = torch.tensor([float(dist.get_rank())])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        line = json.dumps({"generation": gen, "name": os.environ["MGS_BENCH_GEN_NAME"], "sum": float(t[0]), "world": dist.get_world_size(),
                           "native": os.environ.get("MGS_NATIVE_RCCL"), "graph": os.environ.get("MGS_NATIVE_GRAPH"),
                           "degraded": bool(launch.abandoned_generations()), "abandoned_generations": launch.abandoned_generations()})
        launch.persist_result(line)        # before any teardown
        launch.mark_done()
        if mode == "die_before_print":
            os._exit(9)                    # the measurement is finished and persisted: the supervisor prints it
        print(line, flush=True)
        launch.mark_printed()
    dist.destroy_process_group()
    launch.mark_done()
    wd.stop()
    if mode == "crash_after_done":
        os._exit(9)                        # teardown crash after the work is done: must not restart anything


if __name__ == "__main__":
    main()
