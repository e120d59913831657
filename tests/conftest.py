import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import vbnn_oracle as vo
    vo.build()
    vo.lib()
    return vo


# ---- where the time of a test session went (one of twenty GPU runs of round 3 took 654 s instead of ~85 s
# and left nothing that said which test -- or which child process, or the box -- ate nine minutes). Every session now leaves
# gpurun_out/test_durations.txt: the import time of torch in this process (a slow box shows there first), every test above
# one second, and the wall time of every child process a test waited for (tests/_children.py).
import time as _time

_T0 = _time.time()
try:                                                    # (every test module imports torch anyway: time it here, first)
    import torch as _torch
    _torch._vbnn_import_seconds = round(_time.time() - _T0, 2)
except Exception:                                       # pragma: no cover
    pass
_durations = []


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.duration > 1.0):
        _durations.append((report.duration, report.when, report.nodeid, report.outcome))


def pytest_sessionfinish(session, exitstatus):
    try:
        from tests import _children
        kids = list(_children.LOG)
    except Exception:
        kids = []
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        t_imp = None
        if "torch" in sys.modules:
            t_imp = getattr(sys.modules["torch"], "_vbnn_import_seconds", None)
        lines = [f"session: {_time.time() - _T0:.1f} s wall, exit status {int(exitstatus)}, {len(_durations)} test phases recorded",
                 f"import torch in the test process: {t_imp if t_imp is not None else 'not timed'} s"]
        lines.append(f"child processes waited for: {len(kids)}, {sum(k[0] for k in kids):.1f} s in all")
        for el, what in sorted(kids, reverse=True)[:20]:
            lines.append(f"  child {el:8.2f} s  {what}")
        lines.append("tests above 1 s:")
        for d, when, nodeid, outcome in sorted(_durations, reverse=True):
            if d < 1.0:
                break
            lines.append(f"  {d:8.2f} s  {when:5s} {outcome:7s} {nodeid}")
        with open(os.path.join(out, "test_durations.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
