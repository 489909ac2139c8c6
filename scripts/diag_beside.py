"""What costs the meta-steps their speed when something runs beside them on a second stream: (a) nothing, (b) a tiny
kernel that only HOLDS a queue busy, (c) a copy that saturates HBM, (d) the real phase A of another work-array set."""
import os, sys, time, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moc_amd import engine, main_moc as M, synth
from moc_amd._lib import lib, check, ptr
dev = torch.device("cuda:0")
Cc, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, Cc, i % Cc, dev, torch.float32) for i in range(32)]
res = M.ResidentBags(bags, [i % Cc for i in range(32)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(Cc, Cc + 4, j, K, [])
batch, other, lab = plan["batches"][0], plan["batches"][1], plan["labels"]
for b in (batch, other):
    m, kept = engine.draw_row_masks(b.total); b.set_mask(m, kept); b.phase_a(bank)
meta = engine.MetaState(model, opt)
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
hist = torch.zeros(16 * 256, dtype=torch.int32, device=dev)
for _ in range(5):
    engine.train_steps(batch, meta, lab, 0, 32, 15)
torch.cuda.synchronize()


def run(label, beside, reps=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_side = 0
    e0.record()
    for r in range(reps):
        engine.train_steps(batch, meta, lab, 0, 32, 15)
        if beside is not None:
            with torch.cuda.stream(side):
                beside()
            n_side += 1
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{label:44s} {ms * 1e3 / (reps * 32):6.2f} us/step  ({reps * 32 / (ms * 1e-3):7.0f} steps/s), {n_side} side launches, {ms * 1e3 / reps:7.1f} us per pass")


if len(sys.argv) > 1 and sys.argv[1] == "cont":
    # score passes back to back on the side stream for the whole measurement: the rate of the steps UNDER a score pass
    def scores_only(b):
        check(lib().moc_scores(C.byref(b.c), ptr(bank.image), engine._stream()), "moc_scores")
    for label, prep in (("reserved 64, tickets", lambda: other.reserve_cus(64)), ("whole chip, static", lambda: other.reserve_cus(0))):
        prep()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            s0.record(side)
            for _ in range(60):
                scores_only(other)
            s1.record(side)
        e0.record()
        for _ in range(10):
            engine.train_steps(batch, meta, lab, 0, 32, 15)
        e1.record()
        torch.cuda.synchronize()
        print(f"{label}: steps {e0.elapsed_time(e1) * 1e3 / 320:.2f} us/step while score passes run beside at {s0.elapsed_time(s1) * 1e3 / 60:.1f} us each")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "split":
    # the score pass IN LINE on the main stream at the pass boundary (whole chip, static walk, full speed), the rest of
    # phase A on the side stream beside the steps -- against the whole phase A on the side stream (what the loop does)
    other.reserve_cus(0)
    def one_pass_split():
        engine.train_steps(batch, meta, lab, 0, 32, 15)
        check(lib().moc_scores(C.byref(other.c), ptr(bank.image), engine._stream()), "moc_scores")     # main stream, after the steps
        ev = torch.cuda.Event(); ev.record(engine.stream_obj())
        with torch.cuda.stream(side):
            side.wait_event(ev)
            other.select(); other.gather_candidates()
            check(lib().moc_mask_compact(C.byref(other.c), engine._stream()), "mc")                   # (the NEXT pass's flags)
    def one_pass_side(reserve):
        other.reserve_cus(reserve)
        def f():
            engine.train_steps(batch, meta, lab, 0, 32, 15)
            with torch.cuda.stream(side):
                other.phase_a(bank)
        return f
    for label, fn in (("steps only", lambda: engine.train_steps(batch, meta, lab, 0, 32, 15)), ("score in line + rest beside", one_pass_split),
                      ("phase A beside, 64 CUs reserved", one_pass_side(64)), ("phase A beside, whole chip", one_pass_side(0)),
                      ("score in line + rest beside", one_pass_split)):
        if label.startswith("score"):
            other.reserve_cus(0)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{label:40s} {e0.elapsed_time(e1) * 1e3 / 40:7.1f} us per pass ({40 * 32 / (e0.elapsed_time(e1) * 1e-3):7.0f} steps/s)")
    sys.exit(0)
run("nothing beside", None)
if len(sys.argv) > 1 and sys.argv[1] == "hold":
    for nwg, hold in ((1, 300), (8, 100), (8, 300), (8, 1000), (64, 300), (256, 300), (2048, 300), (2048, 100), (16384, 300)):
        run(f"holding {nwg} workgroups for {hold} us", lambda: check(lib().moc_cu_census(ptr(hist), nwg, hold, engine._stream()), "census"))
    sys.exit(0)
run("a kernel holding 8 workgroups for 300 us", lambda: check(lib().moc_cu_census(ptr(hist), 8, 300, engine._stream()), "census"))
run("a kernel holding 2048 workgroups for 100 us", lambda: check(lib().moc_cu_census(ptr(hist), 2048, 100, engine._stream()), "census"))
run("256 MB device copy (HBM)", lambda: big_b.copy_(big_a))
run("phase A of the other set (reserved CUs)", lambda: other.phase_a(bank))
other.reserve_cus(0)
run("phase A of the other set (whole chip)", lambda: other.phase_a(bank))
other.reserve_cus(64)
def only_scores():
    check(lib().moc_scores(C.byref(other.c), ptr(bank.image), engine._stream()), "moc_scores")
run("score pass only (reserved CUs)", only_scores)
def no_scores():
    check(lib().moc_mask_compact(C.byref(other.c), engine._stream()), "mc"); other.select(); other.gather_candidates()
run("phase A without the score pass", no_scores)
