"""ISA lint of the BUILT library (round 6): checks on the gfx950 code objects inside libfabind_hip.so that the C++ source cannot promise.

1. Asynchronous LDS reads.  `ds_read*` / `ds_bpermute*` results land later than the instruction issues; only an `s_waitcnt lgkmcnt(n)` makes
   the destination registers valid.  For compiler-generated reads the compiler inserts the wait; for INLINE-ASM reads with a hand-placed
   wait (gemm_tn_bf16_kernel's `ds_read_b64_tr_b16` fragments) the compiler believes the asm output is valid at once and may copy it
   -- e.g. a loop-carried value at the back-edge -- ahead of the wait.  That was the root cause of the weight-gradient mismatches under
   device sharing (profiles/r05_contention.txt): `v_mov_b64 v[162:163], v[204:205]` ran before the wait that guarded
   v[204:205].  The lint walks every kernel's instructions along all branch targets and reports any instruction that reads or writes a
   VGPR which is the destination of an LDS read still covered by the LGKM counter.
2. `las_step_kernel` / `las_step_bwd_kernel` must not contain packed fp32 math (`v_pk_*_f32`; ADVICE r5: the scalar-FMA work-around of round
   5 depended on the SLP vectoriser's behaviour; round 6 switches packed fp32 off for the two functions and this makes it a build-time check).

usage: python tools/isa_lint.py [path/to/libfabind_hip.so]      exit code 1 on any finding
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("FABIND_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "fabind_amd", "libfabind_hip.so")

_SYM = re.compile(r"^[0-9a-f]{16} <(.+)>:$")
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_VREG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def disassemble(lib=DEFAULT_LIB):
    """-> {kernel symbol: [(addr, mnemonic, operand string)]} over every gfx950 code object bundled in the library."""
    tmp = tempfile.mkdtemp(prefix="isa_lint_")
    try:
        loc = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, loc)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", loc], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                 stdout=subprocess.PIPE, universal_newlines=True).stdout
            cur = None
            for line in out.splitlines():
                m = _SYM.match(line)
                if m:
                    cur = kernels.setdefault(m.group(1), [])
                    continue
                if cur is None:
                    continue
                m = _INS.match(line)
                if m:
                    cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _regs(text):
    """VGPR / AGPR numbers named in an operand string -> set of ('v'|'a', n)."""
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for n in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), n))
    return out


def _is_lds_read(mn):
    return mn.startswith("ds_read") or mn.startswith("ds_load") or mn.startswith("ds_bpermute") or mn.startswith("ds_permute") or \
        mn.startswith("ds_swizzle") or mn.startswith("ds_consume") or mn.startswith("ds_append") or "_rtn" in mn


def _counts_lgkm(mn):
    return mn.startswith("ds_") or mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_scratch_load") or \
        mn.startswith("s_sendmsg") or mn.startswith("s_dcache") or mn.startswith("s_memtime") or mn.startswith("s_memrealtime")


def lint_async_lds(insns, max_states=200000):
    """-> list of (addr, text, pending destination) findings for one kernel."""
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    findings, seen = {}, set()
    # work list of (instruction index, pending): pending = tuple of frozensets, one per outstanding LGKM operation, oldest first
    work = [(0, ())]
    linear_done = set()
    n_states = 0
    while work:
        i, pend = work.pop()
        while i < len(insns):
            k0 = 0
            while k0 < len(pend) and not pend[k0]:          # the oldest operations retire first: leading ones without a VGPR destination
                k0 += 1                                      # can never matter again
            pend = pend[k0:]
            key = (i, pend)
            if key in seen:
                break
            if not pend and i in linear_done:
                break
            seen.add(key)
            if not pend:
                linear_done.add(i)
            n_states += 1
            if n_states > max_states:
                raise RuntimeError("isa_lint: state explosion")
            addr, mn, ops = insns[i]
            if mn == "s_waitcnt":
                m = _LGKM.search(ops)
                if m:
                    n = int(m.group(1))
                    pend = pend[len(pend) - n:] if n and len(pend) > n else (pend if n and len(pend) <= n else ())
                i += 1
                continue
            if mn == "s_endpgm":
                break
            touched = _regs(ops)
            if pend and touched:
                inflight = set().union(*pend)
                if _is_lds_read(mn):
                    # an LDS read may REWRITE an in-flight destination (returns are in order); its address / data operands may not be in flight
                    parts = ops.split(",")
                    srcs = _regs(",".join(parts[1:]))
                    bad = srcs & inflight
                else:
                    bad = touched & inflight
                if bad:
                    findings.setdefault(addr, (addr, "%s %s" % (mn, ops), sorted(bad)))
            if _counts_lgkm(mn):
                dest = frozenset(_regs(ops.split(",")[0])) if _is_lds_read(mn) else frozenset()
                pend = pend + (dest,)
                if len(pend) > 15:           # the hardware counter holds 15: issue stalls until the oldest has returned
                    pend = pend[len(pend) - 15:]
            if mn.startswith("s_cbranch") or mn == "s_branch":
                try:
                    off = int(ops.split()[0])
                except (ValueError, IndexError):
                    off = None
                if off is not None:
                    if off >= 0x8000:            # llvm-objdump prints the 16-bit branch offset unsigned: back-edges are >= 0x8000
                        off -= 0x10000
                    tgt = index.get(addr + 4 + 4 * off)
                    if tgt is not None:
                        work.append((tgt, pend))
                if mn == "s_branch":
                    break
            i += 1
    return [findings[a] for a in sorted(findings)]


def lint_no_packed_f32(insns):
    return [(a, "%s %s" % (mn, ops)) for a, mn, ops in insns if mn.startswith("v_pk_") and mn.endswith("_f32")]


def run(lib=DEFAULT_LIB, verbose=True):
    kernels = disassemble(lib)
    report = {"kernels": len(kernels), "async_lds": {}, "pk_fma_acc": {}, "tn_kernels": 0, "tn_tr_reads": 0}
    for name, insns in kernels.items():
        f = lint_async_lds(insns)
        if f:
            report["async_lds"][name] = f
        if "gemm_tn_bf16_kernel" in name:
            report["tn_kernels"] += 1
            report["tn_tr_reads"] += sum(1 for _, mn, _ in insns if mn == "ds_read_b64_tr_b16")
        if "las_step_kernel" in name or "las_step_bwd_kernel" in name:
            report.setdefault("las_kernels", []).append(name)
            f = lint_no_packed_f32(insns)
            if f:
                report["pk_fma_acc"][name] = f
    if verbose:
        print("isa_lint: %d kernels; %d TN kernels with %d transpose reads; LAS kernels: %d" % (
            report["kernels"], report["tn_kernels"], report["tn_tr_reads"], len(report.get("las_kernels", []))))
        for name, f in report["async_lds"].items():
            print("ASYNC-LDS HAZARD in %s:" % name)
            for addr, text, bad in f[:12]:
                print("   %08x  %-70s touches in-flight %s" % (addr, text, ["%s%d" % r for r in bad]))
        for name, f in report["pk_fma_acc"].items():
            print("packed fp32 math in %s: %s" % (name, f[:4]))
    return report


if __name__ == "__main__":
    r = run(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB)
    sys.exit(1 if (r["async_lds"] or r["pk_fma_acc"]) else 0)
