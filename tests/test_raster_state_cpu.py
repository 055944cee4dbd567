"""The rasteriser's host state (guassianhand_amd/rasterizer.py, round 5): one _DeviceState per device behind a re-entrant lock,
process-wide switches in _Policy, the GH_FLAG_DEPTH24 verdict learned from read-backs only. No GPU: the state objects and the
pure host logic."""
import threading

import pytest

from guassianhand_amd import _abi
from guassianhand_amd import rasterizer as R


def test_states_are_per_device_and_locked():
    a, b = R._state(0), R._state(1)
    assert a is not b and a is R._state(0) and a.index == 0 and b.index == 1
    a.capacity[("shape",)] = 7
    assert ("shape",) not in b.capacity                 # two devices in one process share nothing
    a.capacity.pop(("shape",))
    assert isinstance(a.lock, type(threading.RLock()))
    with a.lock:                                        # re-entrant: check_overflow() is called from inside raster_forward()
        with a.lock:
            pass


def test_legacy_module_attributes_resolve_to_the_current_device():
    st = R._state()
    assert R._capacity is st.capacity and R._depth24 is st.depth24 and R._graph_counters is st.graph_counters
    assert R._split_policy is R._policy.split and R._graph_mode is R._policy.graph_mode
    with pytest.raises(AttributeError):
        R._no_such_attribute


def test_depth24_is_used_only_after_it_was_observed_to_hold():
    st = R._DeviceState(99)
    key = (10, 1, 16, 16, False)
    assert st.depth24.get(key) is not True              # unknown: four passes
    R._learn_depth24(st, key, 0)                        # a complete call whose top byte varied (no information bit)
    assert st.depth24[key] is False
    st.depth24.pop(key)
    R._learn_depth24(st, key, 1)                        # lists truncated by an overflow prove nothing
    assert key not in st.depth24
    R._learn_depth24(st, key, _abi.GH_COUNTER_DEPTH24_OK)
    assert st.depth24[key] is True
    R._learn_depth24(st, key, 8 | _abi.GH_COUNTER_DEPTH24_OK)     # the flag was used and did not hold
    assert st.depth24[key] is False
    R._learn_depth24(st, key, _abi.GH_COUNTER_DEPTH24_OK)         # a later narrow scene does not flip a shape back
    assert st.depth24[key] is False


def test_concurrent_state_updates_from_two_threads():
    """Autograd runs `backward` on its own thread: capacity updates and the pending list are touched from two threads. Hammer the
    read-modify-write sequences the lock protects."""
    st = R._DeviceState(98)
    key = (1, 1, 1, 1, False)
    st.capacity[key] = 0
    n = 20000

    def bump():
        for _ in range(n):
            with st.lock:
                st.capacity[key] = st.capacity.get(key, 0) + 1
                st.pending.append(None)
                st.pending = [p for p in st.pending if p is not None][-4:]

    ts = [threading.Thread(target=bump) for _ in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert st.capacity[key] == 2 * n


def test_report_counter_word_learns_and_raises():
    st = R._state(97)
    key = (5, 2, 32, 32, False)
    R.report_counter_word(key, _abi.GH_COUNTER_DEPTH24_OK, 100, 1000, dev=97)          # information only: no error
    assert st.depth24[key] is True and st.last_D == 100
    with pytest.raises(R.GhOverflowError, match="exceeded max_instances"):
        R.report_counter_word(key, 1, 5000, 1000, dev=97)
    assert st.capacity[key] >= 5000
    with pytest.raises(R.GhStaleGeometryError):
        R.report_counter_word(key, 2, 100, 1000, dev=97)
    with pytest.raises(R.GhOverflowError, match="24 key bits"):
        R.report_counter_word(key, 8, 100, 1000, dev=97)
    assert st.depth24[key] is False
