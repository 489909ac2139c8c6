"""The node-local gradient exchange's sequence / parity protocol, on the CPU: moc_amd/csrc/moc_p2p_proto.h -- the very
code the step kernels compile (p2p_push, p2p_signal_wait, p2p_sum) -- built against GCC atomics and threads
(tests/native/p2p_protocol_host.cpp): ranks with random pauses and a deliberately late rank must sum every step's
pushes exactly (two buffer parities, monotonic sequence numbers, never a slot overwritten under a reader); a rank
that goes silent must cost the others one bounded wait, name itself in their error word, and make every later wait
return at once.  What this cannot show -- the visibility of posted xGMI stores across devices -- is listed in
DESIGN.md section 6."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = tmp_path_factory.mktemp("p2p") / "p2p_host"
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "moc_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "p2p_protocol_host.cpp"), "-o", str(exe)])
    return str(exe)


def _run(exe, *args):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=120)
    kv = dict(re.findall(r"(\w+)=(\w+)", r.stdout))
    return r.returncode, kv


@pytest.mark.parametrize("world,steps", [(2, 200), (4, 60), (8, 30)])
def test_late_and_jittery_ranks_sum_exactly(harness, world, steps):
    rc, kv = _run(harness, world, steps, "late")
    assert rc == 0 and kv["bad"] == "0" and kv["timeouts"] == "0"
    assert int(kv["updates"]) == world * 2 * steps            # every (rank, channel) completed every step


def test_a_silent_rank_costs_one_bounded_wait_and_is_named(harness):
    world, steps = 3, 8
    rc, kv = _run(harness, world, steps, "silent")
    assert rc == 0 and kv["bad"] == "0"
    assert int(kv["updates"]) == world * 2 * 2                # steps 1 and 2 complete everywhere
    live = (world - 1) * 2                                    # (rank, channel) pairs left waiting
    assert int(kv["timeouts"]) == live * (steps - 2)
    assert int(kv["fast_returns"]) == live * (steps - 3)      # only the FIRST wait runs into the time-out
    assert int(kv["error_words_naming_silent_rank"]) == world - 1
