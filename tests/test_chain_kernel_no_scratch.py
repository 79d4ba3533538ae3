"""The wave-stream chain kernels (minimal-sdr_amd/csrc/msdr_chain_mfw.hiph) run at their 128-register budget, 4 waves per SIMD.  A
spilled register there is not a slow path but a trap: a scratch reload inside the tile loop is a vector-memory operation, `s_waitcnt vmcnt`
counts in issue order, so waiting for it also waits for the next tile's prefetch that was issued before it -- the prefetch stops
overlapping (round 3 saw exactly this whenever a change pushed the allocation over: profiles/r03/c3_trims.txt).  This test disassembles
the PRODUCT binary and checks that the two flavours the bench runs -- envelope and SSB tables with the two-section cascade on the matrix
cores -- contain no scratch instruction at all, and that every instantiation keeps the 4-waves-per-SIMD allocation.  No GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "minimal-sdr_amd", "lib", "libmsdr.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_object(tmp_path):
    lib = os.path.join(tmp_path, "libmsdr.so")
    shutil.copy(LIB, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert co, "no gfx950 code object inside libmsdr.so"
    return [os.path.join(tmp_path, f) for f in sorted(co)]      # one per translation unit (msdr_api.hip, msdr_chain_block.hip)


def kernels_of(text, needle):
    out, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1) if needle in m.group(1) else None
            if name:
                out[name] = []
            continue
        if name and "//" in line:
            out[name].append(line.split("//", 1)[0].strip())
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library or llvm-objdump missing")
def test_bench_flavours_of_the_wave_stream_kernel_have_no_scratch_traffic(tmp_path):
    text = "\n".join(subprocess.run([OBJDUMP, "-d", co], check=True, stdout=subprocess.PIPE, text=True).stdout for co in code_object(str(tmp_path)))
    kernels = kernels_of(text, "chain_mfw_kernel")
    assert len(kernels) >= 10, sorted(kernels)
    # chain_mfw_kernel<2, AM, FOLD = true, FR = false>: mangled ...ILi2ELb{0,1}ELb1ELb0E...
    for am in (0, 1):
        hit = [k for k in kernels if "ILi2ELb%dELb1ELb0E" % am in k]
        assert len(hit) == 1, (am, sorted(kernels))
        ins = kernels[hit[0]]
        assert sum(1 for i in ins if i.startswith("v_mfma_")) >= 20, hit[0]          # the right function, with its matrix products
        scratch = [i for i in ins if i.startswith("scratch_")]
        assert not scratch, (hit[0], scratch[:4])


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="library or llvm-readelf missing")
def test_every_wave_stream_instantiation_keeps_four_waves_per_simd(tmp_path):
    notes = "\n".join(subprocess.run([READELF, "--notes", co], check=True, stdout=subprocess.PIPE, text=True).stdout for co in code_object(str(tmp_path)))
    # the metadata note lists, per kernel, .name / .vgpr_count / .agpr_count
    seen = 0
    for block in notes.split(".name:")[1:]:
        name = block.split()[0]
        if "chain_mfw_kernel" not in name:
            continue
        m = re.search(r"\.vgpr_count:\s+(\d+)", block)
        a = re.search(r"\.agpr_count:\s+(\d+)", block)
        if not m:
            continue
        seen += 1
        total = int(m.group(1)) + (int(a.group(1)) if a else 0)
        assert total <= 128, (name, total)
    assert seen >= 10, seen


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library or llvm-objdump missing")
def test_block_cadence_kernels_have_no_scratch_traffic(tmp_path):
    """chain_mfb_kernel / chain_q15mb_kernel (msdr_chain_block.hip) run one or two waves per SIMD at up to 256 registers and prefetch the next
    tile's window behind the matrix products: a spill there has the same cost as in the wave-stream kernels (a scratch reload waits for
    the prefetch in front of it).  Every instantiation, no scratch instruction."""
    text = "\n".join(subprocess.run([OBJDUMP, "-d", co], check=True, stdout=subprocess.PIPE, text=True).stdout for co in code_object(str(tmp_path)))
    for needle, count in (("chain_mfb_kernel", 6), ("chain_q15mb_kernel", 7)):          # (Q15: three chain flavours, each with and without the biquad nodes as a second phase, and the FIR stage)
        kernels = kernels_of(text, needle)
        assert len(kernels) == count, (needle, sorted(kernels))
        for name, ins in kernels.items():
            assert sum(1 for i in ins if i.startswith("v_mfma_")) >= 3, name
            scratch = [i for i in ins if i.startswith("scratch_")]
            assert not scratch, (name, scratch[:4])
