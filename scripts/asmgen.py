"""Tiny helper shared by the gfx950 inline-asm generators (gen_decode_loop.py, gen_encode_loop.py).

It records instructions and keeps the book the hardware's s_waitcnt counters imply: one wave's LDS operations complete
in issue order, and so do its vector-memory operations, so "wait until operation X is done" is
s_waitcnt <cnt>(number of operations of that kind issued after X).

Loops: the body is generated once, with the queues as the code before the loop leaves them.  `verify_loop` replays the
body from the queues a full pass leaves behind and checks that every emitted operand is still <= the exact one (a
smaller operand only waits longer) and reports what each wait additionally retires in the steady state."""
import os
import re


# Code placement (round 5).  One s_nop in front of the producer / consumer encoder's loop -- the same instructions four bytes
# further on -- cost 10 % (0.275 -> 0.303 ms, profiles/r04_placement.txt): these loops are 8 - 12 KiB of straight-line code, and
# which of their 8-byte instructions straddle the 64-byte lines of the instruction cache is decided by where the loop starts.  So
# every generated loop head ("1:") sits on a 64-byte boundary (.p2align 6: the assembler pads with s_nop, executed once per
# statement): an edit upstream of a loop no longer moves it.  GEN_ALIGN=0 switches the directive off, GEN_ALIGN_PAD=n puts n more
# s_nop behind it (the phase experiment: scripts/placement_ab.sh).
ALIGN = int(os.environ.get("GEN_ALIGN", "6"))
ALIGN_PAD = int(os.environ.get("GEN_ALIGN_PAD", "0"))


class Asm:
    def __init__(self):
        self.lines = []
        self.lds = []      # tags of LDS ops in issue order (oldest first)
        self.vm = []       # tags of vector-memory ops in issue order
        self.events = []   # ("lds"|"vm", tag) issues and ("wait_lds"|"wait_vm", tag, operand) waits, in program order

    def i(self, text, comment=None):
        if text == "1:" and ALIGN:
            self.lines.append((f".p2align {ALIGN}", "loop head on a 64-byte boundary: see asmgen.py"))
            for _ in range(ALIGN_PAD):
                self.lines.append(("s_nop 0", None))
        self.lines.append((text, comment))

    def ds(self, text, tag, comment=None):
        self.lds.append(tag)
        self.events.append(("lds", tag))
        self.i(text, comment)

    def vmem(self, text, tag, comment=None):
        self.vm.append(tag)
        self.events.append(("vm", tag))
        self.i(text, comment)

    @staticmethod
    def _younger(queue, tag):
        idx = max(k for k, t in enumerate(queue) if t == tag)
        return len(queue) - 1 - idx

    def wait_lds(self, tag, comment=None, cap=False):
        """wait until the youngest LDS op tagged `tag` (and everything older) has completed; `cap`: the counter only
        reaches 15 -- wait for more than asked instead of failing"""
        n = self._younger(self.lds, tag)
        if cap:
            n = min(n, 15)
        assert n <= 15, (tag, n)
        self.lds = self.lds[len(self.lds) - n:] if n else []
        self.events.append(("wait_lds", tag, n))
        if os.environ.get("GEN_NO_LGKM"):          # timing experiment only: results are wrong
            return
        self.i(f"s_waitcnt lgkmcnt({n})", comment)

    def wait_lds_all(self, comment=None):
        self.lds = []
        self.events.append(("wait_lds", None, 0))
        self.i("s_waitcnt lgkmcnt(0)", comment)

    def wait_vm(self, tag, comment=None):
        n = self._younger(self.vm, tag)
        assert n <= 63, (tag, n)
        self.vm = self.vm[len(self.vm) - n:] if n else []
        self.events.append(("wait_vm", tag, n))
        if os.environ.get("GEN_NO_VMWAIT_ALL"):    # timing experiment only: results are wrong
            return
        self.i(f"s_waitcnt vmcnt({n})", comment)

    def wait_vm_all(self, comment=None):
        self.vm = []
        self.events.append(("wait_vm", None, 0))
        self.i("s_waitcnt vmcnt(0)", comment)

    def verify_loop(self, first_event, lds_entry, vm_entry, passes=2):
        """Replay events[first_event:] `passes` times starting from the given queues (the state at the back edge)."""
        notes = set()
        lds, vm = list(lds_entry), list(vm_entry)
        for _ in range(passes):
            for ev in self.events[first_event:]:
                if ev[0] == "lds":
                    lds.append(ev[1])
                elif ev[0] == "vm":
                    vm.append(ev[1])
                else:
                    kind, tag, n = ev
                    q = lds if kind == "wait_lds" else vm
                    if tag is not None and tag in q:     # (not in q: an earlier, stricter wait already retired it)
                        exact = self._younger(q, tag)
                        assert n <= exact, (kind, tag, n, exact)
                        extra = q[len(q) - exact: len(q) - n]
                        if extra:
                            notes.add(f"{kind}({tag}) also retires {sorted(set(extra))} in the steady state")
                    keep = q[len(q) - n:] if n else []
                    if kind == "wait_lds":
                        lds = keep
                    else:
                        vm = keep
        return lds, vm, sorted(notes)

    def render(self, header_lines, operand_lines):
        out = list(header_lines)
        out.append("asm volatile(")
        for text, comment in self.lines:
            sep = "\\n" if text.endswith(":") else "\\n\\t"
            line = f'    "{text}{sep}"'
            if comment:
                line = f"{line:<118}// {comment}"
            out.append(line)
        out.extend(operand_lines)
        return "\n".join(out) + "\n"

    def n_instr(self):
        return sum(1 for t, _ in self.lines if not t.endswith(":") and not t.startswith(";") and not t.startswith("."))
