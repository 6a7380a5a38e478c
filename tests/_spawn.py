"""Starting the ranks of a multi-process test: torch.multiprocessing.spawn with a rendezvous port that is free NOW.

A port found by binding to 0 and closing the socket can be taken by anybody before the ranks' store listens on it -- the
multiprocessing.Manager the caller creates for the results is such a somebody (seen once on a GPU box: EADDRINUSE in rank 0's
TCPStore), so the caller creates its Manager FIRST, and a start that still loses the race is repeated with another port."""
import socket


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(worker, make_args, nprocs, out=None, attempts=4):
    """mp.spawn(worker, args=make_args(port), nprocs=nprocs); `out` (a Manager dict of results) is cleared before a repeat."""
    import torch.multiprocessing as mp
    for attempt in range(attempts):
        port = free_port()
        try:
            mp.spawn(worker, args=make_args(port), nprocs=nprocs, join=True)
            return port
        except Exception as exc:      # noqa: BLE001
            text = str(exc)
            if ("EADDRINUSE" in text or "address already in use" in text.lower()) and attempt + 1 < attempts:
                if out is not None:
                    out.clear()
                continue
            raise
