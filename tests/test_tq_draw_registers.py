"""fir_f32tq_kernel (minimal-sdr_amd/csrc/msdr_fir_f32tq.hiph) draws its tiles with a hand-written returning atomic whose result lands in
a vector register some time AFTER the instruction issued; the kernel reads that register only behind its own s_waitcnt.  The compiler
knows nothing of this, so a copy or a reuse of the register between the draw and its take would read it too early.  This test
disassembles the PRODUCT binary (lib/libmsdr.so, gfx950 code object) and checks, for every instantiation of the kernel, that between
each `global_atomic_add vD ... sc0` and the `v_readfirstlane_b32 sX, vD` that takes it no instruction touches vD, and that the take
follows an `s_waitcnt vmcnt(N)`.  It also checks the wait's ARITHMETIC: vmcnt(N) proves the draw has returned only if at least N vector-
memory instructions were issued behind it (the counter retires in issue order), so on EVERY path from the atomic to its take the number
of global / scratch / buffer / flat loads, stores and atomics must be >= N -- a compiler that merged, predicated away or dropped one of
the stores or loads the kernel counts on would otherwise let the take read a stale register.  No GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "minimal-sdr_amd", "lib", "libmsdr.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(tmp_path):
    lib = os.path.join(tmp_path, "libmsdr.so")
    shutil.copy(LIB, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert co, "no gfx950 code object inside libmsdr.so"
    # (one code object per translation unit: msdr_api.hip, msdr_chain_block.hip)
    return "\n".join(subprocess.run([OBJDUMP, "-d", os.path.join(tmp_path, f)], check=True, stdout=subprocess.PIPE, text=True).stdout for f in sorted(co))


def parse_kernels(text):
    """{kernel: [(address, text)]} for the fir_f32tq_kernel instantiations; llvm-objdump prints `<text> // <address>: <encoding>`."""
    kernels, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1) if "fir_f32tq_kernel" in m.group(1) else None
            if name:
                kernels[name] = []
            continue
        if name and "//" in line:
            ins, rest = line.split("//", 1)
            kernels[name].append((int(rest.split(":")[0].strip(), 16), ins.strip()))
    return kernels


def touches(ins, num):
    for m in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", ins):
        if m.group(1) is not None:
            if int(m.group(1)) == num:
                return True
        elif int(m.group(2)) <= num <= int(m.group(3)):
            return True
    return False


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library or llvm-objdump missing")
def test_no_instruction_touches_a_pending_draw(tmp_path):
    kernels = parse_kernels(disassemble(str(tmp_path)))
    assert len(kernels) >= 9, sorted(kernels)                   # the step counts 2..10 (x 2 where the first step is skipped)
    draws = 0
    for kname, ins in kernels.items():
        index = {a: i for i, (a, _) in enumerate(ins)}

        def successors(i):
            a, t = ins[i]
            op = t.split()[0]
            if op == "s_endpgm":
                return []
            if op in ("s_branch",) or op.startswith("s_cbranch"):
                off = int(t.split()[1])
                off = off - 65536 if off >= 32768 else off          # signed 16-bit, in dwords from the next instruction
                tgt = index[a + 4 + 4 * off]
                return [tgt] if op == "s_branch" else [tgt, i + 1]
            return [i + 1]

        for i, (a, t) in enumerate(ins):
            m = re.match(r"global_atomic_add v(\d+),", t)
            if not m:
                continue
            assert "sc0" in t, (kname, t)                           # a returning atomic
            assert ins[i - 1][1].startswith("s_and_saveexec_b64") and ins[i + 1][1].startswith("s_mov_b64 exec"), (kname, ins[i - 1], ins[i + 1])
            num = int(m.group(1))
            # every path from the draw reaches its take (s_waitcnt vmcnt + v_readfirstlane of the same register) before anything else touches it
            seen, stack, takes = set(), list(successors(i + 1)), 0
            while stack:
                j = stack.pop()
                if j in seen:
                    continue
                seen.add(j)
                tj = ins[j][1]
                if touches(tj, num):
                    assert re.match(r"v_readfirstlane_b32 s\d+, v%d$" % num, tj), "%s: %r touches the pending draw v%d of %r" % (kname, tj, num, t)
                    assert re.match(r"s_waitcnt vmcnt\(\d+\)", ins[j - 1][1]), (kname, ins[j - 1], tj)
                    takes += 1
                    continue
                nxt = successors(j)
                assert nxt or takes, "%s: a path from draw %r ends without taking it" % (kname, t)
                stack.extend(nxt)
            assert takes >= 1, (kname, t)
            # shortest path (in vector-memory instructions) from the draw to each of its takes: must cover the take's vmcnt(N)
            import heapq
            dist, heap = {i + 1: 0}, [(0, i + 1)]
            while heap:
                dj, j = heapq.heappop(heap)
                if dj > dist.get(j, 1 << 30):
                    continue
                tj = ins[j][1]
                if touches(tj, num):                                # a take (checked above): compare with its wait
                    n_wait = int(re.match(r"s_waitcnt vmcnt\((\d+)\)", ins[j - 1][1]).group(1))
                    assert dj >= n_wait, "%s: a path from %r reaches its take behind only %d vector-memory instructions, but the take waits for vmcnt(%d)" % (kname, t, dj, n_wait)
                    continue
                w = 1 if re.match(r"(global|scratch|buffer|flat)_(load|store|atomic)", tj) else 0
                for nx in successors(j):
                    if dj + w < dist.get(nx, 1 << 30):
                        dist[nx] = dj + w
                        heapq.heappush(heap, (dj + w, nx))
            draws += 1
    assert draws >= 4 * len(kernels)                             # three synchronous draws, one per slow iteration, one per fast iteration
