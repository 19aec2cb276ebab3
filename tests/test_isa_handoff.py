"""CPU: the cross-workgroup hand-off of the fused loss kernels, checked in the compiled gfx950 ISA.

csrc/mcav_common.h states the protocol ("every workgroup leaves a partial result, the last one to draw a ticket finishes"): published words
are stored at agent scope (sc1), EVERY thread waits for its stores to be acknowledged (s_waitcnt vmcnt(0)) before the workgroup's ticket --
an agent-scope atomic add -- is taken, and the finisher reads with sc1 loads.  The HIP memory model only promises this for agent-scope
release / acquire fences, which on this chip cost an L2 write-back + invalidate per workgroup; the cheap form relies on what the compiler
emits, so this test compiles csrc/warp_loss.hip for the device (hipcc cross-compiles without a GPU, ~10 s) and reads the assembly: a
compiler or ROCm upgrade that drops the sc1 bit or moves the wait is caught here, not as silently wrong losses on the GPU.
"""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "unsupervised-pseuso-lidar_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _device_functions(tmp_path_factory, source):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / (source + ".s"))
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           "-o", out, os.path.join(CSRC, source)], stderr=subprocess.DEVNULL)
    text = open(out).read()
    # split into functions: "<name>:" at column 0 up to its ".Lfunc_end"
    funcs = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        body = [l.strip() for l in m.group(2).splitlines()]
        funcs[m.group(1)] = [l for l in body if l and not l.startswith((";", ".", "//")) or l.startswith(".LBB")]
    return funcs


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    return _device_functions(tmp_path_factory, "warp_loss.hip")


@pytest.fixture(scope="module")
def nn_kernels(tmp_path_factory):
    return _device_functions(tmp_path_factory, "nn_ops.hip")


def _ticket_kernels(funcs):
    return {n: b for n, b in funcs.items() if any(i.startswith("global_atomic_add") for i in b)}


def test_loss_kernels_take_tickets(kernels):
    tk = _ticket_kernels(kernels)
    names = " ".join(tk)
    assert "warp_loss_l1_kernel" in names and "warp_loss_ssim_kernel" in names, names


def test_stores_are_acknowledged_before_every_ticket(kernels):
    """Walking back from each ticket atomic, an `s_waitcnt vmcnt(0)` comes before any global store (barriers and ALU work may sit between)."""
    for name, body in _ticket_kernels(kernels).items():
        for i, ins in enumerate(body):
            if not ins.startswith("global_atomic_add"):
                continue
            assert " sc0" in ins, (name, ins)                     # returns the pre-op value: the ticket
            for j in range(i - 1, -1, -1):
                prev = body[j]
                if prev.startswith("s_waitcnt") and "vmcnt(0)" in prev:
                    break
                assert not prev.startswith(("global_store", "buffer_store", "flat_store")), \
                    "%s: a global store is issued between the last s_waitcnt vmcnt(0) and the ticket at instruction %d: %s" % (name, i, prev)
            else:
                raise AssertionError("%s: no s_waitcnt vmcnt(0) in front of the ticket at instruction %d" % (name, i))


def test_published_words_and_finisher_reads_are_agent_scope(kernels):
    """After the first ticket of a kernel only the finisher runs: every store it issues to global memory that other workgroups / the next
    launch's finisher read (sample sums, ticket resets) is sc1, and it reads the slab and the sample sums with sc1 loads.  In front of the
    ticket the workgroup's slab entry is stored sc1."""
    for name, body in _ticket_kernels(kernels).items():
        first = next(i for i, ins in enumerate(body) if ins.startswith("global_atomic_add"))
        before, after = body[:first], body[first:]
        sc1_before = [i for i in before if i.startswith("global_store") and " sc1" in i]
        assert sc1_before, "%s: no agent-scope slab store in front of the ticket" % name
        sc1_loads = [i for i in after if i.startswith("global_load") and " sc1" in i]
        assert len(sc1_loads) >= 2, "%s: the finisher's slab / sample-sum loads are not agent-scope: %s" % (name, sc1_loads)
        sc1_after = [i for i in after if i.startswith("global_store") and " sc1" in i]
        assert len(sc1_after) >= 3, "%s: sample sums / ticket resets are not agent-scope stores: %s" % (name, sc1_after)
        assert not any(i.startswith(("buffer_wbl2", "buffer_inv")) for i in body), "%s: an L2 write-back / invalidate crept in" % name


def test_batchnorm_finalize_hand_off(nn_kernels):
    """bn_partial_finalize_kernel (csrc/nn_ops.hip: stage-1 partial sums and the finalize in one launch, round 4) uses the same protocol: its
    partial sums are stored sc1, every thread waits vmcnt(0) before the column's ticket, the finisher loads sc1 and resets the ticket sc1."""
    tk = {n: b for n, b in _ticket_kernels(nn_kernels).items() if "bn_partial_finalize" in n}
    assert tk, list(nn_kernels)[:5]
    test_stores_are_acknowledged_before_every_ticket(tk)
    for name, body in tk.items():
        first = next(i for i, ins in enumerate(body) if ins.startswith("global_atomic_add"))
        assert any(i.startswith("global_store") and " sc1" in i for i in body[:first]), name
        assert sum(1 for i in body[first:] if i.startswith("global_load") and " sc1" in i) >= 2, name
        assert any(i.startswith("global_store") and " sc1" in i for i in body[first:]), name
        assert not any(i.startswith(("buffer_wbl2", "buffer_inv")) for i in body), name


@pytest.fixture(scope="module")
def bf16_kernels(tmp_path_factory):
    return _device_functions(tmp_path_factory, "conv_bf16.hip")


def test_transposing_lds_reads_of_the_weight_gradient_kernel_run_with_all_lanes(bf16_kernels):
    """wgrad3x3_patch_kernel feeds its MFMAs through ds_read_b64_tr_b16, whose gather crosses lanes: the ISA requires EXEC to be all ones (a
    masked lane's stale address still takes part and the active lanes get wrong data, silently -- cdna_hip_programming.md T10).  The kernel's
    only divergent code is the guarded staging stores; in the compiled ISA no transposing read may sit between an `s_and_saveexec` and the
    instruction that restores EXEC, the split-form kernels carry the 96 reads of a block (4 steps x (3 + 9) fragments x 2), and nothing spills."""
    names = [n for n in bf16_kernels if "wgrad3x3_patch_kernel" in n]
    assert len(names) == 4, names                      # <3 | 1 planes> x <bias | no bias>
    for n in names:
        body = bf16_kernels[n]
        reads = [i for i, ins in enumerate(body) if ins.startswith("ds_read_b64_tr_b16")]
        assert len(reads) == (96 if "ILi3E" in n else 32), (n, len(reads))
        assert not any(ins.startswith("scratch_") for ins in body), n      # no register spills
        masked = False
        for ins in body:
            if ins.startswith("s_and_saveexec_b64") or ins.startswith("s_andn2_saveexec_b64"):
                masked = True
            elif re.match(r"s_(or|mov|xor|andn2)_b64 exec,", ins) or ins.startswith(".LBB"):
                # (a label is a join point: the compiler restores EXEC at it or before the code that follows runs unmasked; the reads of a step
                # follow the previous step's guarded stores, so the walk must see a restore -- or a join -- before them)
                if re.match(r"s_(or|mov)_b64 exec,", ins):
                    masked = False
            elif ins.startswith("ds_read_b64_tr_b16"):
                assert not masked, "%s: a transposing LDS read under a partial EXEC mask" % n
