"""Host-side checks of the tiled aggregation's kernels that need no GPU.

* ``tests/host/tiled_index_check.cpp`` evaluates, through the same ``__host__ __device__``
  functions the kernels call (``csrc/stm_sgm_tiled_index.h``), every global-memory access of the
  sweep, walk and tile kernels for H, W in {1,2,3,7,8,9,15,16,17,40,95,151,375,450}, D in
  {64,128,256}, batch in {1,2}, and asserts that each lies inside its buffer (cost volume, image,
  border-state blocks, mailboxes, outputs), clamped prefetches and dummy entry reads included.
* The generated code of the sweep kernels re-issues the mailbox load on every trip of a wait
  loop (the polling load must not be hoisted out of it).
* ``tests/host/skew_index_check.cpp`` replays the DATAFLOW of the skewed-tile form
  (``csrc/stm_sgm_skew_index.h``) symbolically: every state the two passes store gets the label
  (direction, pixel, producing tile); every entry state they load must find the label the
  recurrence needs at its address, produced by a tile the progress rule has waited for; every
  address lies inside the image's workspace block.
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo-matching_amd", "csrc")


def test_every_tiled_kernel_address_lies_inside_its_buffer(tmp_path):
    exe = str(tmp_path / "tiled_index_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", CSRC,
                           os.path.join(ROOT, "tests", "host", "tiled_index_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.match(r"(\d+) shapes, (\d+) checks, 0 failures", out.stdout)
    assert m and int(m.group(1)) == 14 * 14 * 3 * 2 and int(m.group(2)) > 10 ** 9, out.stdout


def test_skewed_tile_dataflow_is_consistent(tmp_path):
    """Stores and loads of the skewed-tile form meet: right state at every address, producers
    covered by progress_needed(), everything inside the block, chains tile the tiles -- for 330
    shapes (H, W in {1,2,7,8,9,15,16,17,40,95,151,375,450}, D in {128, 256}).  Mutations of the
    header (a shifted offset, a wrong side index, a shorter lead) make it fail."""
    exe = str(tmp_path / "skew_index_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", CSRC,
                           os.path.join(ROOT, "tests", "host", "skew_index_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.match(r"(\d+) shapes, (\d+) checks, 0 failures", out.stdout.strip().splitlines()[-1])
    assert m and int(m.group(1)) == 330 and int(m.group(2)) > 10 ** 7, out.stdout
    # the check has teeth: a bottom exit shifted by one column must be caught
    hdr = open(os.path.join(CSRC, "stm_sgm_skew_index.h")).read()
    bad = hdr.replace("K.fb = rel_row(D, c - 8, d0);", "K.fb = rel_row(D, c - 7, d0);")
    assert bad != hdr
    mut = tmp_path / "mut"
    mut.mkdir()
    (mut / "stm_sgm_skew_index.h").write_text(bad)
    (mut / "stm_sgm_tiled_index.h").write_text(open(os.path.join(CSRC, "stm_sgm_tiled_index.h")).read())
    exe2 = str(tmp_path / "skew_index_check_mut")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", str(mut),
                           os.path.join(ROOT, "tests", "host", "skew_index_check.cpp"), "-o", exe2])
    assert subprocess.run([exe2], capture_output=True, text=True).returncode != 0
    # ... and the kernels form none of these offsets by hand
    body = open(os.path.join(CSRC, "stm_sgm_skew_body.h")).read()
    code = "\n".join(line.split("//")[0] for line in body.splitlines())
    for pattern in ("rowb_in_img(", "side_in_img(", "G.nTC *", "* G.nTC", "kSide"):
        assert pattern not in code, "stm_sgm_skew_body.h forms a state address by hand (%r)" % pattern


def test_kernels_use_the_index_header_for_their_addresses():
    """The kernels' files form no border-state or mailbox address by hand: the expressions the
    range model covers are the only ones (a guard against the model and the kernels drifting
    apart)."""
    for name in ("stm_sgm_sweep.hip", "stm_sgm_hwalk.hip", "stm_sgm_tile.hip", "stm_sgm_tile_h2.hip",
                 "stm_sgm_tile_common.h"):
        src = open(os.path.join(CSRC, name)).read()
        code = "\n".join(line.split("//")[0] for line in src.splitlines() if "STM_STAMP" not in line)
        for pattern in ("* G.nTR +", "* G.nTC +", "nTR + ", "nTC + "):  # ((k * nTR + t) * W + x) * D and friends
            assert pattern not in code, name + ": border-state index formed by hand (%r)" % pattern
        assert "kMboxSlots *" not in code and "% kMboxSlots" not in code, name + ": mailbox index formed by hand"


import pytest


@pytest.mark.parametrize("unit,least", [("stm_sgm_sweep.hip", 12), ("stm_sgm_skew_d128.hip", 4)])
def test_mailbox_poll_is_reissued_inside_the_wait_loop(tmp_path, unit, least):
    """ISA of the sweep kernels (mailboxes) and of the skewed-tile kernels (progress counters):
    every spin loop (s_sleep) contains a load with the sc1 policy before its backward branch --
    the poll is not hoisted out of the loop."""
    hipcc = "/opt/rocm/bin/hipcc"
    asm = str(tmp_path / "unit.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-S",
                           os.path.join(CSRC, unit), "-o", asm],
                          stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    sleeps = [i for i, l in enumerate(lines) if re.match(r"\s*s_sleep\b", l)]
    assert len(sleeps) >= least, "expected the wait loops of the unit's kernels"
    for i in sleeps:
        body = []
        for l in lines[i + 1:i + 80]:
            body.append(l)
            if re.match(r"\s*s_cbranch", l):
                break
        loads = [l for l in body if re.search(r"(buffer|global)_load_dword", l)]
        assert loads and all(" sc1" in l for l in loads), "no sc1 poll in the loop at line %d:\n%s" % (
            i + 1, "\n".join(body))


def _vrange(reg):
    m = re.match(r"v\[(\d+):(\d+)\]", reg)
    if m:
        return int(m.group(1)), int(m.group(2))
    n = int(reg[1:])
    return n, n


def test_register_soffset_buffer_stores_hold_their_data(tmp_path):
    """ISA of the kernels that store through buffer descriptors with an SGPR soffset: each such
    store is followed by the s_nop of hold_store_data (stm_common.h) before anything writes its
    data registers -- hipcc
    pads the store-data hazard only for immediate soffsets, and on gfx950 the next VALU write
    to the data registers otherwise races with the store's read of them."""
    hipcc = "/opt/rocm/bin/hipcc"
    checked = ("stm_sgm_sweep.hip", "stm_sgm_hwalk.hip", "stm_dp.hip", "stm_sgm_skew_d128.hip", "stm_sgm_skew_d256.hip")
    for f in os.listdir(CSRC):  # every unit that stores through a descriptor is in the list
        if not f.endswith(".hip"):
            continue
        src = open(os.path.join(CSRC, f)).read()
        for inc in re.findall(r'#include "(stm_sgm_skew_body\.h)"', src):  # (a unit that is its body header)
            src += open(os.path.join(CSRC, inc)).read()
        if f.endswith(".hip") and re.search(r"raw_buffer_store|buf_store_row", src):
            assert f in checked, f + " stores through a buffer descriptor: add it to this test"
    for name in checked:
        asm = str(tmp_path / (name + ".s"))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-S",
                               os.path.join(CSRC, name), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
        n = 0
        for i, l in enumerate(lines):
            m = re.match(r"\s*buffer_store_dword(x[234])?\s+(v\[?\d+[:\d\]]*), (?:v\d+|off), s\[\d+:\d+\], (\S+)", l)
            # (every store with a register soffset, and every 16-byte store whatever its soffset)
            if not m or not (m.group(3).startswith("s") or m.group(1) == "x4"):
                continue
            n += 1
            lo, hi = _vrange(m.group(2))
            held = False
            for x in lines[i + 1:i + 120]:  # up to the s_nop: nothing writes the data registers
                if re.match(r"\s*s_nop\b", x):
                    held = True
                    break
                w = re.match(r"\s*(v_\w+|global_load\w*|buffer_load\w*|scratch_load\w*|ds_read\w*)\s+(v\[\d+:\d+\]|v\d+)\b", x)
                if w:
                    a, b = _vrange(w.group(2))
                    assert b < lo or a > hi, "%s line %d: %s writes the data of %s" % (name, i + 1, x.strip(), l.strip())
                assert not re.match(r"\s*(s_cbranch|s_branch|s_endpgm)", x), "%s line %d: control flow behind %s before its s_nop" % (name, i + 1, l.strip())
            assert held, "%s line %d: no s_nop behind %s" % (name, i + 1, l.strip())
        if name != "stm_dp.hip":
            assert n > 0, name + ": expected buffer stores with a register soffset"


def test_tile_loops_touch_no_scratch_memory(tmp_path):
    """ISA of the kernels that rebuild tiles (skewed-tile passes, tile kernels behind sweeps / walks):
    no scratch or flat access beyond a kernel's prologue and epilogue.  A spilled register or an
    array behind a pointer comes back by `scratch_load` + `s_waitcnt vmcnt(0)`, and vmcnt counts in
    order: the wait covers the entry states and the next tile's cost rows requested ahead on
    purpose -- a fifth of the skewed-tile forward pass until round 4."""
    hipcc = "/opt/rocm/bin/hipcc"
    units = ("stm_sgm_skew_d128.hip", "stm_sgm_skew_d256.hip", "stm_sgm_tile.hip", "stm_sgm_tile_h2.hip")
    procs = []
    for name in units:
        asm = str(tmp_path / (name + ".s"))
        procs.append((name, asm, subprocess.Popen(
            [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "--cuda-device-only", "-S", os.path.join(CSRC, name), "-o", asm], stderr=subprocess.DEVNULL)))
    for name, asm, p in procs:
        assert p.wait() == 0, name
        lines = open(asm).read().splitlines()
        # the tile loop: the skewed-tile kernels' loop over a band's tiles lies inside the loop over the
        # units (image, band) a workgroup takes; the tile kernels' loop over tiles is the outermost
        tile_depth = 2 if "skew" in name else 1
        kernel, depth, in_loop = None, 0, []
        for l in lines:
            m = re.match(r"^(_Z\w+):", l)
            if m:
                kernel, depth = m.group(1), 0
            # (hipcc annotates every basic block of a loop: "; in Loop: Header=BB0_70 Depth=2")
            if re.match(r"^\.LBB\d+_\d+:", l):
                d = re.search(r"in Loop: Header=\S+ Depth=(\d+)", l)
                depth = int(d.group(1)) if d else 0
            if re.match(r"\s*(scratch_load|scratch_store|flat_load|flat_store)", l) and kernel and depth >= tile_depth:
                in_loop.append((kernel, l.strip()))
        assert not in_loop, "%s: scratch / flat accesses inside the tile loop:\n%s" % (
            name, "\n".join("%s: %s" % kl for kl in in_loop[:10]))
