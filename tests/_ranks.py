"""Test helper: start the ranks of a multi-process test on a fresh rendezvous port."""
import socket

_NET_ERRORS = ("Address already in use", "EADDRINUSE", "Connection refused", "Connection reset", "timed out", "TCPStore",
               "DistNetworkError", "DistStoreError")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(worker, nprocs, args_of_port, attempts=3):
    """`mp.spawn(worker, args_of_port(port), nprocs)`.  The port is probed and released before the ranks bind it, so another
    process can take it in between (seen once in ~10 full runs of the GPU suite): a rendezvous that fails on the network is
    started again on another port; an exception out of the worker's own work (an assertion, a kernel error) is raised as it is."""
    import torch.multiprocessing as mp
    for k in range(attempts):
        try:
            mp.spawn(worker, args=args_of_port(free_port()), nprocs=nprocs, join=True)
            return
        except Exception as e:   # ProcessRaisedException carries the rank's traceback as text
            if k + 1 == attempts or not any(w in str(e) for w in _NET_ERRORS):
                raise
            print(f"[spawn_ranks] rendezvous failed ({str(e).strip().splitlines()[-1]}); new port")
