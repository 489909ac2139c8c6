"""Static check of the kernels that issue loads through inline asm (moc_scores.hip): between an asm
`global_load_dwordx4 v[a:b], ...` / `ds_read_b128 v[a:b], ...` and the hand-written `s_waitcnt
vmcnt(N)` / `lgkmcnt(N)` that covers it, no instruction may read or write v[a:b] -- the compiler does
not know the data is still in flight, and under register pressure it has been seen to copy such
registers away and re-use them.  Both counters retire in order, so a wait for N leaves
exactly the N youngest loads of its kind in flight."""
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


def _check_kernel(name, lines, precise=True):
    """precise: a wait for N keeps the N youngest loads pending (valid where the textual order of the
    ISA is the execution order between an issue and its wait: the streaming kernels).  Otherwise a wait
    retires everything issued textually before it (the row kernel's rotated loop)."""
    pend = {"vm": [], "lgkm": []}            # destination register sets, oldest first
    in_asm, n_loads, n_waits = False, {"vm": 0, "lgkm": 0}, {"vm": 0, "lgkm": 0}
    alt_open = False
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm, alt_open = True, False
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if t.startswith("global_load_lds"):
            # an LDS-DMA (the compiler's builtin, not asm): no destination register, but it takes a place in the in-order
            # vector-memory queue that the counted waits retire -- the wide ring kernel mixes them with register loads
            in_flight = set().union(*pend["vm"], *pend["lgkm"]) if (pend["vm"] or pend["lgkm"]) else set()
            addr = set().union(*[_regs(x) for x in toks]) if toks else set()
            assert not (addr & in_flight), f"{name}: `{t}` takes its address from in-flight registers"
            pend["vm"].append(set())
            continue
        # (global_atomic_add with return: the streaming kernels' tile tickets -- same in-order queue as the loads)
        # global_load_dword: the dummy ticket operation of a unit that is not a tile boundary
        kind = "vm" if (t.startswith("global_load_dword") or t.startswith("global_atomic_add")) else \
            "lgkm" if t.startswith("ds_read_b128") else None
        if in_asm and kind:
            in_flight = set().union(*pend["vm"], *pend["lgkm"]) if (pend["vm"] or pend["lgkm"]) else set()
            addr = set().union(*[_regs(x) for x in toks[1:]]) if len(toks) > 1 else set()
            assert not (addr & in_flight), f"{name}: `{t}` takes its address from in-flight registers"
            assert not (_regs(toks[0]) & in_flight), f"{name}: `{t}` overwrites registers still in flight"
            if t.startswith("global_load_dword ") and alt_open:
                # the ticket operation: ONE asm statement holds the request and, behind a branch, the dummy load that runs
                # instead of it -- one place in the queue, either destination may be the one in flight
                pend[kind][-1] = pend[kind][-1] | _regs(toks[0])
                alt_open = False
                continue
            alt_open = t.startswith("global_atomic_add")
            pend[kind].append(_regs(toks[0]))
            n_loads[kind] += 1
            continue
        m = re.match(r"s_waitcnt (vmcnt|lgkmcnt)\((\d+)\)", t)
        if in_asm and m:
            k = "vm" if m.group(1) == "vmcnt" else "lgkm"
            keep = int(m.group(2))
            pend[k] = pend[k][-keep:] if (keep and precise) else []
            n_waits[k] += 1
            continue
        used = set()
        for tok in toks:
            used |= _regs(tok)
        for k in pend:
            for regs in pend[k]:
                assert not (used & regs), f"{name}: `{t}` touches in-flight {k} load registers {sorted(used & regs)[:8]}"
    return n_loads, n_waits


@pytest.mark.timeout(300)
def test_no_instruction_touches_inflight_asm_load_registers(tmp_path):
    asm = tmp_path / "scores.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                           "--cuda-device-only", "-o", str(asm), SRC], stderr=subprocess.DEVNULL)
    text = asm.read_text().splitlines()
    kernels, cur, name = {}, None, None
    for ln in text:
        m = re.match(r"^(_ZN\S*scores_(?:stream|wide|wide_ring)_kernel\S*):", ln)
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
        n_loads, n_waits = _check_kernel(k, lines, precise=True)
        if "scores_stream_kernel" in k:
            assert n_loads["vm"] >= 16 and n_waits["vm"] >= 2, (k, n_loads, n_waits)
        assert n_loads["lgkm"] >= 8 and n_waits["lgkm"] >= 6, (k, n_loads, n_waits)
