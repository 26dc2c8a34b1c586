"""The hand-scheduled K loops (conv3x3_halo3 / conv_quad_halo3) synchronise with a COUNTED wait: `s_waitcnt lgkmcnt(N)`
in front of a raw `s_barrier` retires this step's LDS writes only if exactly N fragment reads were issued after the last write — an
ordering the sources pin with sched_group_barrier but the compiler ultimately decides.  This test cross-compiles the three kernels for
gfx950 (no GPU needed) and checks the emitted ISA: in every step that ends in the counted wait, at least N ds_read (and nothing else on the LDS queue) follow the last ds_write."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


_ISA_CACHE = {}


def _isa(src, extra=()):
    key = (src, tuple(extra))
    if key not in _ISA_CACHE:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            # (the product flags of __graft_entry__._compile for the convolution family)
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                            os.path.join(ROOT, "diffusynth_amd", "csrc", src)] + list(extra), check=True, capture_output=True)
            with open(out) as f:
                _ISA_CACHE[key] = f.read()
    return _ISA_CACHE[key]


def _steps(isa):
    """Per kernel: list of (ops string, N) for every barrier-delimited segment that ends in the inline-asm counted wait."""
    res, name, ops, pending_asm = {}, None, [], False
    for line in isa.split("\n"):
        m = re.match(r"^(_ZN\S*kernel\S*):", line)
        if m:
            name, ops = m.group(1), []
            res[name] = []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            pending_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            pending_asm = False
            continue
        op = t.split()[0] if t else ""
        if pending_asm and op == "s_waitcnt":
            n = re.search(r"lgkmcnt\((\d+)\)", t)
            if n:
                ops.append("K%s" % n.group(1))
            continue
        if op.startswith("ds_write") or op.startswith("ds_store"):
            ops.append("W")
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            ops.append("R")
        elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_scratch_load") or \
                op.startswith("flat_") or op.startswith("s_sendmsg") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
            # anything else that is counted on the LGKM queue (other LDS operations, scalar memory reads — which return OUT of order —
            # flat accesses, messages): in the window between the last LDS write and the counted wait it would make "at most N
            # outstanding" mean something else than "every write has retired"
            ops.append("X")
        elif op == "s_barrier":
            if ops and ops[-1].startswith("K"):
                res[name].append(ops)
            ops = []
        elif op == "s_endpgm":
            name = None
    return res


@pytest.mark.parametrize("src,n_reads,min_steps", [("conv3x3_halo3.hip", 10, 18), ("conv_quad_halo3.hip", 10, 24)])
def test_counted_lgkm_wait_covers_every_lds_write(src, n_reads, min_steps):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    steps = _steps(_isa(src))
    assert steps, "no kernels found in the ISA"
    for kern, segs in steps.items():
        assert len(segs) >= min_steps, (kern, len(segs))
        for ops in segs:
            assert ops[-1] == "K%d" % n_reads, (kern, ops)
            body = ops[:-1]
            assert "W" in body, (kern, ops)
            after = body[len(body) - 1 - body[::-1].index("W") + 1:]
            # in-order completion: waiting until at most N operations are outstanding retires everything older than the last N —
            # provided nothing but LDS reads shares the queue from the first write of the step on (scalar loads return out of order)
            assert set(after) <= {"R"} and len(after) >= n_reads, (kern, "".join(o[0] for o in ops))
            assert "X" not in body[body.index("W"):], (kern, "".join(o[0] for o in ops))


def test_the_wait_check_catches_a_build_that_breaks_the_protocol():
    """The protocol is the compiler's to break: sched_group_barrier is best effort.  With the register budget of THREE blocks per CU
    (-DDS_MINBLK=3: 168 VGPRs, 520 - 600 bytes of scratch per lane) hipcc emits steps whose last LDS write is followed by fewer than ten
    reads — lgkmcnt(10) would then let the barrier pass with writes in flight.  The check above must flag that build (so that it can be
    trusted when it passes the product build)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    steps = _steps(_isa("conv3x3_halo3.hip", ["-DDS_MINBLK=3"]))
    broken = 0
    for kern, segs in steps.items():
        for ops in segs:
            body = ops[:-1]
            if "W" not in body:
                continue
            after = body[len(body) - 1 - body[::-1].index("W") + 1:]
            broken += not (set(after) <= {"R"} and len(after) >= 10)
    assert broken > 0


@pytest.mark.parametrize("src,scratch_max", [("conv3x3_halo3.hip", 0), ("conv3x3_smalln.hip", 0), ("conv1x1_x3.hip", 0), ("conv_quad_halo3.hip", (0, 0, 40))])
def test_hand_scheduled_kernels_do_not_spill(src, scratch_max):
    """The product build of the hand-scheduled kernels must fit its register budget (two blocks per CU = 256 VGPRs) without scratch.
    Why it matters (round 4): sched_group_barrier is best effort, and a build under register pressure re-orders a step's LDS
    writes behind some of its fragment reads — the counted `s_waitcnt lgkmcnt(10)` then no longer retires every write (the test above
    checks exactly that on the emitted ISA; the -DDS_MINBLK=3 build shows it happening).
    conv_quad_halo3: the 32- and 16-wide instantiations are at zero since round 4 (tap addresses as base + XOR mask, wave-uniform second
    weight piece, epilogue coordinates re-derived from mbcnt); the 8-wide one stages one more halo iteration (seven) and still parks up to
    ten loop-invariant registers in scratch (reloaded once per six chunks, through vmcnt — not the LGKM queue): pinned so it cannot grow."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    isa = _isa(src)
    names = re.findall(r"\.amdhsa_kernel (\S+)", isa)
    sizes = [int(x) for x in re.findall(r"; ScratchSize: (\d+)", isa)]
    vgprs = [int(x) for x in re.findall(r"; NumVgprs: (\d+)", isa)]
    assert sizes and len(sizes) == len(names)
    if isinstance(scratch_max, tuple):
        # instantiations in source order of their tile width: <5>, <4>, <3>
        order = sorted(range(len(names)), key=lambda i: -int(re.search(r"ILi(\d)", names[i]).group(1)))
        assert [sizes[i] <= scratch_max[k] for k, i in enumerate(order)] == [True] * len(order), list(zip(names, sizes))
    else:
        assert all(x <= scratch_max for x in sizes), sizes
    assert vgprs and all(x <= 256 for x in vgprs), vgprs
    assert "v_pk_fma_f32" not in isa and "v_pk_add_f32" not in isa and "v_pk_mul_f32" not in isa      # packed fp32 starves beside a busy MFMA pipe
