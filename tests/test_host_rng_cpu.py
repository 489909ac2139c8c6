"""The library's host-side mask generator must reproduce torch's CPU generator bit for bit
(main_moc.py:330 draws `torch.rand(N) > 0.5` there) and leave it in the same state."""
import torch

from moc_amd import engine


def _positions():
    for seed in (0, 1, 123, 99999, 2 ** 31 + 5):
        for pre in (0, 1, 623, 624, 625, 1000):
            yield seed, pre


def test_masks_equal_torch_rand_stream():
    for seed, pre in _positions():
        sizes = [5, 619, 1, 1, 3, 624, 625, 1247, 15000, 7]
        torch.manual_seed(seed)
        torch.rand(pre)
        ref = [torch.rand(n) > 0.5 for n in sizes]
        after_ref = torch.rand(4)
        torch.manual_seed(seed)
        torch.rand(pre)
        got = [engine.draw_row_masks(n) for n in sizes]
        after = torch.rand(4)
        for r, (g, k) in zip(ref, got):
            assert g.dtype == torch.uint8 and torch.equal(r, g.bool()) and int(r.sum()) == k
        assert torch.equal(after_ref, after), "generator left in a different state than torch.rand would"


def test_one_call_equals_per_slide_calls():
    sizes = [15000, 14321, 9, 60000, 2000]
    torch.manual_seed(42)
    ref = torch.cat([torch.rand(n) > 0.5 for n in sizes])
    torch.manual_seed(42)
    allm, kept = engine.draw_row_masks(sum(sizes))
    assert torch.equal(ref, allm.bool()) and kept == int(ref.sum())


def test_other_draws_interleave_correctly():
    torch.manual_seed(7)
    a1, n1, a2 = torch.rand(10) > 0.5, torch.randn(5), torch.rand(700) > 0.5
    torch.manual_seed(7)
    b1, _ = engine.draw_row_masks(10)
    m1 = torch.randn(5)
    b2, _ = engine.draw_row_masks(700)
    assert torch.equal(a1, b1.bool()) and torch.equal(n1, m1) and torch.equal(a2, b2.bool())


def test_falls_back_under_float64_default():
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(3)
        ref = torch.rand(100) > 0.5
        torch.manual_seed(3)
        got, _ = engine.draw_row_masks(100)
        assert torch.equal(ref, got.bool())
    finally:
        torch.set_default_dtype(torch.float32)


def test_rejects_a_state_with_nothing_left():
    """`left` == 0 never leaves at::mt19937 (it regenerates AT zero); the replay refuses it instead of indexing
    before its output (the caller then lets torch draw)."""
    import ctypes as C
    from moc_amd._lib import lib, ptr
    torch.manual_seed(11)
    st = torch.get_rng_state()
    st[8:12] = 0                                    # i32 left = 0
    out = torch.empty(16, dtype=torch.uint8)
    assert lib().moc_host_draw_masks(ptr(st), st.numel(), 16, ptr(out)) < 0
    assert b"unexpected generator state" in C.cast(lib().moc_last_error(), C.c_char_p).value


def test_max_kept_is_the_largest_per_slide_count():
    import ctypes as C
    import numpy as np
    from moc_amd._lib import lib
    rng = np.random.default_rng(3)
    sizes = [1, 17, 4096, 15000, 9, 8193]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    mask = (rng.random(off[-1]) > 0.5).astype(np.uint8)
    mask[off[2]:off[3]] = 1                                   # one slide keeps everything
    got = lib().moc_host_max_kept(mask.ctypes.data, off.ctypes.data, len(sizes))
    assert got == max(int(mask[off[b]:off[b + 1]].sum()) for b in range(len(sizes)))
    mask[off[2]:off[3]] = 0
    mask[off[3]:off[4]] = 0
    got = lib().moc_host_max_kept(mask.ctypes.data, off.ctypes.data, len(sizes))
    assert got == max(int(mask[off[b]:off[b + 1]].sum()) for b in range(len(sizes)))
    assert lib().moc_host_max_kept(None, off.ctypes.data, 3) == -1


def test_mask_drawer_hands_out_torchs_bits_whether_or_not_its_speculation_holds():
    """engine.MaskDrawer draws the NEXT pass's flags on a helper thread; whatever happens to the generator in between,
    what it hands out for a state are the bits torch.rand(N) > 0.5 gives from that state."""
    import ctypes as C
    sizes = [700, 1300, 5, 2048]
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    row_off_c = (C.c_int64 * len(off))(*off)
    d = engine.MaskDrawer(off[-1], row_off_c, len(sizes), pinned=False)

    class Ev:                       # stands in for the CUDA event attached to a buffer
        def __init__(self):
            self.done = False

        def query(self):
            return self.done

        def synchronize(self):
            self.done = True

    def reference(state):
        torch.set_rng_state(state)
        m = torch.cat([torch.rand(n) > 0.5 for n in sizes])
        return m, torch.get_rng_state()

    torch.manual_seed(2024)
    state = torch.get_rng_state()
    used = []
    for step in range(12):
        if step in (4, 9):                                   # somebody else draws: the speculation must be dropped
            torch.set_rng_state(state)
            torch.rand(3)
            state = torch.get_rng_state()
        buf, kept, mk, after, i = d.take(state)
        ref, ref_after = reference(state)
        assert torch.equal(buf.bool(), ref) and kept == int(ref.sum())
        assert mk == max(int(ref[off[b]:off[b + 1]].sum()) for b in range(len(sizes)))
        assert torch.equal(after, ref_after)
        ev = Ev()
        d.attach(i, ev)
        used.append((i, ev, buf.clone(), buf))
        if len(used) > 2:                                    # the GPU finishes with a buffer two passes later
            used[-3][1].done = True
        for j, e, snap, b in used[-2:]:
            assert torch.equal(snap, b), "a buffer still in use was overwritten"
        state = after
