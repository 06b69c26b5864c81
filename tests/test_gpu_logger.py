"""-m gpu: the engine's logger honours the reference's switches (utils/logger.cc:22-56): FT_LOG_LEVEL names, the warning for an
unknown name, "[FT][LEVEL] message" lines on stderr (stdout stays the caller's)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = """
import sys
sys.path.insert(0, %r)
from tests import gpu_helpers as gh
from tests.helpers import load_tiny
cfg, w, z = load_tiny()
op = gh.make_op(cfg, w)
gh.run_op(op, z["prompt"][None, :], [16], 4, cfg["vocab_size"], top_k=1)
print("done")
""" % ROOT


def _run(level):
    env = dict(os.environ)
    env.pop("FT_LOG_LEVEL", None)
    if level is not None:
        env["FT_LOG_LEVEL"] = level
    r = subprocess.run([sys.executable, "-c", PROG], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "done"
    return [l for l in r.stderr.splitlines() if l.startswith("[FT]")]


def test_ft_log_level():
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    default = _run(None)  # INFO, like a release build of the reference
    assert any(l.startswith("[FT][INFO] GptNeoX engine on device") for l in default)
    assert not any(l.startswith("[FT][DEBUG]") or l.startswith("[FT][TRACE]") for l in default)
    trace = _run("TRACE")
    assert any(l.startswith("[FT][DEBUG] decoder of this request: persistent layers") for l in trace)
    assert any(l.startswith("[FT][TRACE] begin: batch 1") for l in trace)
    assert _run("ERROR") == []
    bogus = _run("LOUD")
    assert bogus and bogus[0].startswith("[FT][WARNING] Invalid logger level FT_LOG_LEVEL=LOUD")
    assert any(l.startswith("[FT][INFO]") for l in bogus)  # ... and the default level applies
