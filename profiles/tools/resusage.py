"""profiling tool: per-kernel register / scratch / LDS usage from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or a file)."""
import re, sys
t = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0]
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    short = re.sub(r"^_ZN2pf", "", name)[:80]
    print(f"{short:80s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} SGPR {g('SGPRs'):>3} scratch {g('ScratchSize [bytes/lane]'):>5} LDS {g('LDS Size [bytes/block]'):>6} occ {g('Occupancy [waves/SIMD]')}")
