"""dev: scan a gfx950 .s file for global loads that are drained right after issue: a `s_waitcnt vmcnt(N)` with N smaller than
the number of loads issued since the previous barrier, within WINDOW instructions of the last of them.  Such a wait exposes
the memory latency where it stands (see profiles/r01/NOTES.md, deferred zero-fill).  usage: scan_waits.py file.s [filter]"""
import re, sys
WINDOW = 25
src = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_ZN4nnop\w+):.*?s_endpgm", src, re.S | re.M):
    name = m.group(1)
    if flt not in name:
        continue
    L = [l for l in m.group(0).splitlines() if re.match(r"\s+[a-z]", l)]
    hits = []
    pending = []          # indices of loads not yet waited for
    for i, l in enumerate(L):
        if re.search(r"global_load|buffer_load", l):
            pending.append(i)
        w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if w and pending:
            n = int(w.group(1))
            if n < len(pending):
                waited = pending[: len(pending) - n]
                if i - waited[-1] <= WINDOW:
                    hits.append((i, i - waited[-1], len(waited), len(L)))
                pending = pending[len(pending) - n:]
        if "s_barrier" in l:
            pass
    if hits:
        print(name[:100])
        for i, d, k, tot in hits:
            print(f"    instr {i}/{tot}: waits for {k} load(s) issued {d} instructions earlier")
