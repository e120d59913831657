"""Child processes a test waits for (torchrun ranks, the C host, a bare-shell bench): run() is subprocess.run with the wall time
kept, so that a slow session says WHERE it waited (tests/conftest.py writes gpurun_out/test_durations.txt)."""
import subprocess
import time

LOG = []            # (seconds, what)


def run(cmd, **kw):
    t0 = time.time()
    try:
        return subprocess.run(cmd, **kw)
    finally:
        tail = " ".join(str(c) for c in cmd[-6:])
        LOG.append((time.time() - t0, tail[-160:]))
