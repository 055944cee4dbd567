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


def test_depth24_is_not_learned_from_words_that_say_nothing():
    """a counter word without bit 4 marked a shape False for good even when no depth sort had run (a refresh over static
    lists, a call with nothing to project: counters memset to 0). Only a full forward with D > 0 decides."""
    st = R._DeviceState(97)
    key = (10, 1, 16, 16, False)
    R._learn_depth24(st, key, 0, 0)                     # nothing was listed: no key took part in the (OR, AND)
    assert key not in st.depth24
    R._learn_depth24(st, key, _abi.GH_COUNTER_DEPTH24_OK, 0)      # "the top byte did not vary" over zero keys is vacuous
    assert key not in st.depth24
    R._learn_depth24(st, key, 0, 5)
    assert st.depth24[key] is False
    # a graph-mode refresh registers its counters with learn24 = False: its word (no depth sort: bit 4 absent) leaves the shape alone
    st2 = R._state(0)
    key2 = (123456, 1, 16, 16, False)
    st2.depth24.pop(key2, None)
    R.report_counter_word(key2, 0, 7, 100, dev=0, learn24=False)
    assert key2 not in st2.depth24
    R.report_counter_word(key2, _abi.GH_COUNTER_DEPTH24_OK, 7, 100, dev=0)
    assert st2.depth24.pop(key2) is True


def test_check_overflow_reports_the_most_severe_record_and_keeps_the_others():
    """with several bad records pending, check_overflow raised the LAST one and dropped the rest — a capacity overflow
    behind a bound miss was never reported. Now: one error per call, the most severe first, the others stay for the next check."""
    st = R._DeviceState(96)

    def done(**kw):
        pc = R._Pending(st, None, None, 100, (1, 1, 16, 16, False))
        pc.done, pc.over = True, True
        for k, v in kw.items():
            setattr(pc, k, v)
        return pc

    miss, cap, stale = done(miss=True, d=10), done(d=500), done(stale=True, d=10)
    st.pending = [cap, miss, stale]
    R._states[96] = st
    try:
        with pytest.raises(R.GhStaleGeometryError):
            R.check_overflow(dev=96)
        assert len(st.pending) == 2
        with pytest.raises(R.GhOverflowError) as e:
            R.check_overflow(dev=96)
        assert not isinstance(e.value, (R.GhDepthBoundMiss, R.GhStaleGeometryError)) and "D=500" in str(e.value)
        with pytest.raises(R.GhDepthBoundMiss):
            R.check_overflow(dev=96)
        assert st.pending == []
        R.check_overflow(dev=96)                           # nothing left
    finally:
        R._states.pop(96, None)


def test_camera_record_cache_is_keyed_by_identity_and_version():
    import torch
    from guassianhand_amd import camera as Cm
    Cm.clear_pack_cache()
    V, Pm, cp, bg0, bg1 = torch.eye(4), torch.eye(4) * 2, torch.zeros(3), torch.zeros(3), torch.ones(3)
    a = Cm.pack_camera(V, Pm, cp, 0.1, 0.2, bg0)
    b = Cm.pack_camera(V, Pm, cp, 0.1, 0.2, bg1)            # the mask call's other bg: a second entry, not an eviction
    assert Cm.pack_camera(V, Pm, cp, 0.1, 0.2, bg0) is a and Cm.pack_camera(V, Pm, cp, 0.1, 0.2, bg1) is b
    assert a.shape == (1, _abi.GH_CAM_FLOATS) and torch.equal(a[0, 37:], bg0) and torch.equal(b[0, 37:], bg1) and float(a[0, 35]) == pytest.approx(0.1)
    bg0.add_(0.5)                                           # an in-place update moves the version: a new record
    c = Cm.pack_camera(V, Pm, cp, 0.1, 0.2, bg0)
    assert c is not a and torch.equal(c[0, 37:], bg0)
    assert Cm.pack_camera(V.clone(), Pm, cp, 0.1, 0.2, bg0) is not c      # another object, same values: identity decides
    for k in range(2 * Cm._PACK_MAX):                       # bounded
        Cm.pack_camera(V, Pm, cp, 0.1 + k, 0.2, bg1)
    assert len(Cm._pack_cache) <= Cm._PACK_MAX
