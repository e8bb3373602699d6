"""profiling tool: instruction sizes of a disassembled kernel (llvm-objdump -d of the code object): how many 4- and 8-byte encodings,
bytes per instruction over the whole text and per basic-block-sized window -- a lone wave fetches ~1.6 B of code per clock and issues one
instruction per 4 clocks (profiles/r06/lone_wave_issue.txt), so a stretch above 6.4 B per instruction is fetch-bound.
usage: code_bytes.py kernel.dis [window]"""
import re, sys
from collections import Counter
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-F]+):\s*((?:[0-9A-F]{8}\s*)+)$", l)
    if m:
        rows.append((m.group(1), int(m.group(2), 16), 4 * len(m.group(3).split())))
n = len(rows)
by = Counter(sz for _, _, sz in rows)
tot = sum(sz for _, _, sz in rows)
print(f"{n} instructions, {tot} bytes, {tot / n:.2f} B per instruction; by size {dict(by)}")
big = Counter(op for op, _, sz in rows if sz >= 8)
print("8-byte opcodes:", ", ".join(f"{k} {v}" for k, v in big.most_common(25)))
small = Counter(op for op, _, sz in rows if sz == 4)
print("4-byte opcodes:", ", ".join(f"{k} {v}" for k, v in small.most_common(15)))
# cycles under the two-limit model: per window of W instructions max(4 W, bytes / 1.6)
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cyc = sum(max(4 * len(rows[i:i + W]), sum(sz for _, _, sz in rows[i:i + W]) / 1.6) for i in range(0, n, W))
print(f"model over the whole text (windows of {W}): {cyc / n:.2f} clocks per instruction; issue-only 4.00; gain if every instruction were 4 bytes: {100 * (1 - 4 * n / cyc):.1f} %")
