"""Rendezvous for the multi-process tests: torch.distributed's FileStore on a path of the test's own (no TCP port at all:
a port chosen with bind(0); close() and handed to spawned ranks seconds later can be taken in between -- EADDRINUSE ended
the driver's round-5 GPU run at item 196 and hid everything collected behind it)."""
import os
import tempfile
import uuid


def new_rendezvous(tmp=None):
    """init_method string for dist.init_process_group: a file that does not exist yet, in a directory of its own."""
    d = str(tmp) if tmp is not None else tempfile.mkdtemp(prefix="sf_rdzv_")
    return "file://" + os.path.join(d, "rdzv_" + uuid.uuid4().hex)
