"""Assertions about ROUND COUNTS (and printed seconds) -- not about bytes.  They live here, collected after every parity test
(the file name sorts last and conftest.py moves anything marked `perf` to the end), so that under `pytest -x` a
performance regression can never again keep parity tests from running (one such assertion inside
test_rzip_gpu.py hid 83 tests)."""
import ctypes as C
import time

import pytest

import datagen

pytestmark = [pytest.mark.gpu, pytest.mark.perf]


@pytest.mark.parametrize("kind", ["few", "phrases"])
def test_serial_stretches_replace_rounds_that_do_not_pay(B, O, kind):
    """An input whose tags crowd into a few buckets ('few': every candidate conflicts with its neighbours), or one that is a
    short match every 25 candidates ('phrases'), commits two or three candidates per round: the resolver must not pay a
    35 us round for them.  The verdict counts candidates (no clock), so the figures are the same on every box."""
    from test_chunks_gpu import _bench, _profile
    from test_rzip_gpu import _check
    data = datagen.KINDS[kind]((2 << 20) + 99, seed=9)
    B.lib().lrzgpu_profile_reset()
    st = _check(B, O, data, level=7)
    p = _profile(B, _bench())
    rounds, committed, exact = (int(v) for v in p.resolve_dbg[:3])
    assert committed + exact <= st.lookups  # (a candidate that stands again after its match was emitted is looked up twice)
    assert rounds < 2500 or committed > 0.9 * st.lookups, (rounds, committed, exact)


@pytest.mark.parametrize("kind", ["few", "phrases"])
def test_degenerate_inputs_are_off_their_cliff(B, kind):
    """5 MiB of a four-letter alphabet / of 50 random phrases through the scan at level 7: 1.1 / 2.2 s of k_resolve since
    the dense variant of the resolver (5.0 / 6.1 s in round 5, when every candidate was a 2 - 4 us exact
    step).  What is ASSERTED is a count -- at most one candidate in ten takes the exact step -- because seconds depend on
    the box and on what the process did before (behind 410 other tests the same call once took 5.0 s instead of 2.9:
    workspaces allocated beside 100 GB of parked pools); the seconds are printed."""
    from test_chunks_gpu import _bench, _profile
    data = datagen.KINDS[kind]((5 << 20) + 123, seed=9)
    B.hash_search(data[: 1 << 20], level=7)  # (workspaces, code objects)
    B.lib().lrzgpu_profile_reset()
    t0 = time.time()
    _, _, st, _, _ = B.hash_search(data, level=7)
    dt = time.time() - t0
    p = _profile(B, _bench())
    rounds, committed, exact = (int(v) for v in p.resolve_dbg[:3])
    print("%s: %.2f s (k_resolve %.0f ms), %d rounds, %d committed, %d exact steps of %d lookups" % (kind, dt, p.resolve_ms, rounds, committed, exact, st.lookups))
    assert exact < 0.1 * st.lookups and committed > 0.9 * st.lookups, (rounds, committed, exact, st.lookups)


def test_block_above_the_ceiling_is_refused_at_once(B):
    """LRZGPU_E_BLOCK_TOO_LARGE comes before anything is scanned (the refusal itself is asserted in
    test_configs_gpu.py): no kernel of the scan has run when the call returns -- a count, not a stopwatch."""
    import torch
    from test_chunks_gpu import _bench, _profile
    n = 3 << 30
    buf = torch.zeros(n + 256, dtype=torch.uint8, device="cuda")
    B.lib().lrzgpu_profile_reset()
    t0 = time.time()
    with pytest.raises(RuntimeError, match="rc=-108"):
        B.compress_device(buf.data_ptr(), n, level=7, threads=1, processors=1, ramsize=64 << 30, host_threads=4)
    p = _profile(B, _bench())
    print("refused after %.2f s" % (time.time() - t0))
    assert p.tag_scan_launches == 0 and p.resolve_launches == 0 and p.mf_launches == 0
