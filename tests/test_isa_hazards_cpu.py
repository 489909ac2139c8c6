"""Static check of the kernels that issue vector loads through inline asm (moc_scores.hip): between an
asm `global_load_dwordx4 v[a:b], ...` and the hand-written `s_waitcnt vmcnt(N)` that covers it, no
instruction may read or write v[a:b] -- the compiler does not know the data is still in flight, and
under register pressure it has been seen to copy such registers away and re-use them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "moc_amd", "csrc", "moc_scores.hip")


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _check_kernel(name, lines):
    pending, in_asm, n_loads, n_waits = set(), False, 0, 0
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if in_asm and t.startswith("global_load_dwordx4"):
            pending |= _regs(toks[0])
            n_loads += 1
            continue
        if in_asm and t.startswith("s_waitcnt vmcnt"):
            pending.clear()          # every site issues the newer loads BEFORE this wait; those are
            n_waits += 1             # re-armed below by the next asm loads, older ones are complete
            continue
        used = set()
        for tok in toks:
            used |= _regs(tok)
        assert not (used & pending), f"{name}: `{t}` touches in-flight load registers {sorted(used & pending)[:8]}"
    return n_loads, n_waits


@pytest.mark.timeout(300)
def test_no_instruction_touches_inflight_asm_load_registers(tmp_path):
    asm = tmp_path / "scores.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                           "--cuda-device-only", "-o", str(asm), SRC], stderr=subprocess.DEVNULL)
    text = asm.read_text().splitlines()
    kernels, cur, name = {}, None, None
    for ln in text:
        m = re.match(r"^(_ZN\S*scores_(?:stream|rows)_kernel\S*):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln)
            if "s_endpgm" in ln:
                kernels[name] = cur
                cur = None
    assert len(kernels) >= 4, f"expected the asm-load kernels in the ISA, found {list(kernels)}"
    for k, lines in kernels.items():
        n_loads, n_waits = _check_kernel(k, lines)
        assert n_loads >= 16 and n_waits >= 2, (k, n_loads, n_waits)
