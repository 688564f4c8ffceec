"""Tiny helper shared by the gfx950 inline-asm generators (gen_decode_loop.py, gen_encode_loop.py).

It records instructions and keeps the book the hardware's s_waitcnt counters imply: one wave's LDS operations complete
in issue order, and so do its vector-memory operations, so "wait until operation X is done" is
s_waitcnt <cnt>(number of operations of that kind issued after X)."""


class Asm:
    def __init__(self):
        self.lines = []
        self.lds = []      # tags of LDS ops in issue order (oldest first)
        self.vm = []       # tags of vector-memory ops in issue order

    def i(self, text, comment=None):
        self.lines.append((text, comment))

    def ds(self, text, tag, comment=None):
        self.lds.append(tag)
        self.i(text, comment)

    def vmem(self, text, tag, comment=None):
        self.vm.append(tag)
        self.i(text, comment)

    def _wait(self, queue, tag, limit):
        idx = max(k for k, t in enumerate(queue) if t == tag)
        younger = len(queue) - 1 - idx
        assert younger <= limit, (tag, younger)
        return younger, queue[idx + 1:]

    def wait_lds(self, tag, comment=None):
        """wait until the youngest LDS op tagged `tag` (and everything older) has completed"""
        n, self.lds = self._wait(self.lds, tag, 15)
        self.i(f"s_waitcnt lgkmcnt({n})", comment)

    def wait_lds_all(self, comment=None):
        self.i("s_waitcnt lgkmcnt(0)", comment)
        self.lds = []

    def wait_vm(self, tag, comment=None):
        n, self.vm = self._wait(self.vm, tag, 63)
        self.i(f"s_waitcnt vmcnt({n})", comment)

    def wait_vm_all(self, comment=None):
        self.i("s_waitcnt vmcnt(0)", comment)
        self.vm = []

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
        return sum(1 for t, _ in self.lines if not t.endswith(":"))
