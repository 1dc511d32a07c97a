"""The right to launch co-resident workgroups (banded / slab chain forms) is ONE per device across processes
(_native.coresident_right: an advisory file lock keyed by the GPU's identity), and a module on a device that stays
shared backs off instead of re-trying the banded form every 256 forwards.  CPU only: the
device identity is stubbed, the lock itself is the real one."""
import os
import subprocess
import sys
import warnings

from conftest import ROOT

_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from multi_view_stereonet_amd import _native
_native._device_key = lambda index: "test-gpu-%d" % index           # (no GPU here: stub the identity, keep the lock)
got = _native.coresident_right(0)
again = _native.coresident_right(0)                                  # cached: asking twice never blocks or flips
other = _native.coresident_right(1)                                  # another device has its own lock
print("RIGHT", int(got), int(again), int(other), flush=True)
ready, go = sys.argv[2], sys.argv[3]
open(ready, "w").close()
t0 = time.time()
while not os.path.exists(go) and time.time() - t0 < 60:
    time.sleep(0.02)
"""


def _spawn(tmp, name, env):
    ready, go = os.path.join(tmp, name + ".ready"), os.path.join(tmp, "go")
    p = subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, ready, go], env=env, stdout=subprocess.PIPE, text=True)
    return p, ready


def _wait(path, proc, timeout=120):
    import time
    t0 = time.time()
    while not os.path.exists(path):
        assert proc.poll() is None and time.time() - t0 < timeout, "worker did not come up"
        time.sleep(0.02)


def test_only_one_process_per_device_gets_the_coresident_right(tmp_path):
    tmp = str(tmp_path)
    env = dict(os.environ, MVSN_LOCK_DIR=tmp)
    env.pop("MVSN_CORESIDENT_LOCK", None)
    a, a_ready = _spawn(tmp, "a", env)
    _wait(a_ready, a)                                  # A holds device 0 and device 1
    b, b_ready = _spawn(tmp, "b", env)
    _wait(b_ready, b)
    c, c_ready = _spawn(tmp, "c", dict(env, MVSN_CORESIDENT_LOCK="0"))      # the switch: granted without asking
    _wait(c_ready, c)
    open(os.path.join(tmp, "go"), "w").close()
    outs = [p.communicate(timeout=60)[0].strip() for p in (a, b, c)]
    assert outs == ["RIGHT 1 1 1", "RIGHT 0 0 0", "RIGHT 1 1 1"], outs
    # the owner is gone: the next process gets the right (flock dies with its holder, no stale-file problem)
    os.remove(os.path.join(tmp, "go"))
    d, d_ready = _spawn(tmp, "d", env)
    _wait(d_ready, d)
    open(os.path.join(tmp, "go"), "w").close()
    assert d.communicate(timeout=60)[0].strip() == "RIGHT 1 1 1"
    assert sorted(f for f in os.listdir(tmp) if f.endswith(".lock")) == ["mvsn_coresident_test-gpu-0.lock",
                                                                         "mvsn_coresident_test-gpu-1.lock"]


def test_release_and_unwritable_lock_directory(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    from multi_view_stereonet_amd import _native
    monkeypatch.setattr(_native, "_device_key", lambda index: "test-gpu-%d" % index)
    monkeypatch.setattr(_native, "_coresident", {})
    monkeypatch.setenv("MVSN_LOCK_DIR", str(tmp_path))
    monkeypatch.delenv("MVSN_CORESIDENT_LOCK", raising=False)
    assert _native.coresident_right(0) and _native._coresident[0][1] is not None
    assert open(_native.coresident_lock_path(0)).read().strip() == str(os.getpid())
    _native.release_coresident_right(0)
    assert 0 not in _native._coresident
    monkeypatch.setenv("MVSN_LOCK_DIR", os.path.join(str(tmp_path), "does", "not", "exist"))
    assert _native.coresident_right(0) and _native._coresident[0][1] is None     # granted: the pre-lock behaviour
    _native.release_coresident_right()


def test_latch_backs_off_on_a_device_that_stays_shared():
    """the latch expired every 256 forwards for ever (a multi-second time-out + a warning each time on a
    permanently shared device).  Now every re-latch doubles the wait (capped) and the warning is issued once."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from multi_view_stereonet_amd.multi_view_stereonet import _SharedState
    st = _SharedState()
    st.words_np = np.zeros(4, dtype=np.int32)

    def repaired_forward():
        st.words_np[0] |= 3
        st.words_np[1] += 1
        return st.poll()

    def forwards_until_rearmed():
        n = 0
        while st.banded_latched:
            st.tick()
            n += 1
        return n

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        waits = []
        for _ in range(10):
            assert repaired_forward() == 1 and st.banded_latched
            waits.append(forwards_until_rearmed())
    assert waits[:4] == [256, 512, 1024, 2048] and waits[-1] == _SharedState.LATCH_CAP
    assert len([w for w in caught if issubclass(w.category, RuntimeWarning)]) == 1
    st.rearm()
    assert (st.latch_forwards, st.relatches, st.warned, st.banded_latched) == (256, 0, False, False)
