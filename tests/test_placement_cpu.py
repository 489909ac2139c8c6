"""Host side of the score pass's placement (include/moc_hip.h: compute units kept free of the score pass): which slots
the reserved table names, given what moc_cu_census found.  No GPU: the census result is made up here."""
import pytest

from moc_amd import engine as E


def _mi355x_like(harvest=()):
    """8 XCDs x 4 shader engines x 8 CUs, HW_ID[15:8] = SE << 5 | SH << 4 | CU; `harvest`: slots that do not exist."""
    slots = [(x, (se << 5) | cu) for x in range(8) for se in range(4) for cu in range(9)]
    gone = set(harvest) | {(x, (se << 5) | 8) for x in range(8) for se in range(4)}     # nine physical, eight active
    return sorted(t for t in slots if t not in gone)


@pytest.mark.parametrize("n", [8, 32, 64, 72, 100])
def test_an_equal_share_of_every_xcd_spread_over_its_shader_engines(n):
    slots = _mi355x_like()
    assert len(slots) == 256
    chosen = E.choose_reserved_slots(slots, n)
    assert len(chosen) == n == len(set(chosen)) and set(chosen) <= set(slots)
    per_xcd = [sum(1 for x, _ in chosen if x == k) for k in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 1 and sum(per_xcd) == n
    for k in range(8):
        per_se = [sum(1 for x, s in chosen if x == k and s >> 5 == se) for se in range(4)]
        assert max(per_se) - min(per_se) <= 1, (k, per_se)
    # the highest CU ids of an engine go first
    for x, s in chosen:
        higher = [t for t in slots if t[0] == x and t[1] >> 4 == s >> 4 and t[1] > s]
        assert all(t in chosen for t in higher)


def test_irregular_harvest_and_bitmap():
    slots = _mi355x_like(harvest={(0, (1 << 5) | 7), (0, (1 << 5) | 6), (5, (3 << 5) | 0)})
    chosen = E.choose_reserved_slots(slots, 64)
    assert len(set(chosen)) == 64 and set(chosen) <= set(slots)
    words = E.reserved_words(chosen)
    assert len(words) == 128 and sum(bin(w).count("1") for w in words) == 64
    for x, s in chosen:
        bit = x * 256 + s
        assert words[bit >> 5] >> (bit & 31) & 1
    assert len(slots) * 3 // 4 - 8 <= len(E.choose_reserved_slots(slots, len(slots))) <= len(slots) * 3 // 4     # clamped to three quarters of the chip (of every XCD)
    with pytest.raises(AssertionError):
        E.choose_reserved_slots(slots, 0)


def test_an_incomplete_census_is_not_fatal():
    """Other processes may hold whole compute units while the census runs: the table then names fewer units (never more
    than three quarters of what an XCD showed), it does not raise."""
    slots = [t for t in _mi355x_like() if not (t[0] == 3 and (t[1] & 15) >= 2)]      # XCD 3 showed 8 of its 32 units
    chosen = E.choose_reserved_slots(slots, 64)
    assert sum(1 for x, _ in chosen if x == 3) == 6 and len(chosen) == 62 and set(chosen) <= set(slots)


def test_a_small_device_or_partition_does_not_crash_the_default_plan():
    """MOC_RESERVE_CUS = 64 on a device that shows 32 compute units (a CPX partition, HSA_CU_MASK) used to die
    with an AssertionError at plan build; it is clamped to three quarters of what exists, and a device too small to give any
    unit away keeps the static whole-chip walk."""
    part = [(0, (se << 5) | cu) for se in range(4) for cu in range(8)]            # one XCD, 32 units
    chosen = E.choose_reserved_slots(part, 64)
    assert len(chosen) == 24 and set(chosen) <= set(part)
    assert [sum(1 for _, s in chosen if s >> 5 == se) for se in range(4)] == [6, 6, 6, 6]
    assert E.choose_reserved_slots(part[:1], 64) == []
