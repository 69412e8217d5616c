#!/usr/bin/env python
"""Scan the device code of libsome_amd.so for an instruction pattern that is NOT safe on gfx950 although hipcc (ROCm 7.2) emits it:

    buffer_store_dwordx4 v[158:161], v146, s[8:11], s13 offen      ; > 64 bits of store data, SGPR soffset
    v_cvt_f32_f16_e32 v158, v138                                    ; the very next instruction overwrites a data register

The GCN / CDNA ISA manuals list "VMEM store of more than 64 bits followed by a write of the VGPRs holding the write data: 1 wait
state" with the exemption "BUFFER_STORE_* operations that use an SGPR for the offset do not require any wait states", and LLVM's
hazard recogniser implements exactly that exemption (no s_nop when soffset is a register).  Measured on MI355X in round 3
(profiles/r03_sgpr_epilogue_hazard.md): with the exempted form the store writes the NEW register value some of the time - the
"run-to-run different results" of round 2's SGPR-wave-index FFN1 epilogue.  Inserting `s_nop 1` behind every such store in the
compiler's assembly (tools/build_asm_patch.py, nothing else changed) makes the kernel bit-stable.

The shipped kernels avoid the form (their wide stores have VGPR-derived offsets, so the compiler wraps them in waterfall loops or
inserts the wait state itself); this scanner is the guard: tests/test_isa_hazards.py runs it over the built library.

    python tools/isa_hazard_scan.py [path/to/libsome_amd.so]      # exit code 1 if a hazard is found
"""
import pathlib
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = pathlib.Path('/opt/rocm/lib/llvm/bin')
WIDE_STORE = re.compile(r'^(buffer_store_dwordx[34]|buffer_store_format_xyzw?|buffer_store_format_d16_xyzw?|tbuffer_store_format_xyzw?|'
                        r'global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\b')
NO_WAIT = ('s_waitcnt', 's_nop')          # s_nop / any other non-VALU instruction in between provides the wait state


def _regs(tok: str):
    tok = tok.strip()
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'^v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def _parse(line: str):
    """objdump line -> (mnemonic, [operands]) or None"""
    code = line.split('//')[0].strip()
    if not code or code.endswith(':') or code.startswith(('.', ';')):
        return None
    parts = code.split(None, 1)
    ops = []
    if len(parts) > 1:
        depth, cur = 0, ''
        for ch in parts[1]:
            if ch == '[':
                depth += 1
            elif ch == ']':
                depth -= 1
            if ch == ',' and depth == 0:
                ops.append(cur.strip()); cur = ''
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
    return parts[0], ops


def _defs(mn: str, ops):
    """VGPRs a VALU instruction writes (its first operand).  Only VALU results can land inside the store's data read: LDS / VMEM
    loads into the same registers return tens of cycles later (hipcc emits `global_store_dwordx4 ..., v[92:95]` + `ds_read_b128
    v[92:95]` back to back in the QKV epilogue, which the repeat-run bit-equality tests have covered since round 2)."""
    if not ops or not mn.startswith('v_'):
        return set()
    if mn.startswith(('v_cmp', 'v_cmpx', 'v_readfirstlane', 'v_readlane', 'v_nop')):
        return set()
    return _regs(ops[0].split()[0])


PACK_CVT = re.compile(r'^v_cvt_scalef32_(2xpk16|pk32|sr_pk32|sr_2xpk16)_')


def scan_text(text: str):
    """-> list of hits.  Rule 1: wide stores whose data registers are overwritten 1 slot later.  Rule 2 (round 3, tools/mx_probe2.hip):
    a multi-register MX pack convert (v_cvt_scalef32_2xpk16_* / pk32_*) whose destination tuple overlaps one of its sources - hipcc
    allocates such overlaps when the source dies at the instruction, and the multi-pass instruction then reads what it has already
    overwritten (csrc/split.h cvt_fp6_2x16 keeps the sources alive)."""
    hits = []
    kernel = None
    lines = text.splitlines()
    insts = []
    for ln in lines:
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', ln.strip())
        if m:
            kernel = m.group(1)
            continue
        p = _parse(ln)
        if p:
            insts.append((kernel, p[0], p[1], ln.split('//')[0].strip()))
    for i, (kern, mn, ops, txt) in enumerate(insts):
        if PACK_CVT.match(mn) and len(ops) >= 2:
            dst = _regs(ops[0])
            if any(dst & _regs(o.split()[0]) for o in ops[1:]):
                hits.append({'kernel': kern, 'store': txt, 'next': '(destination overlaps a source of the pack convert)', 'sgpr_soffset': False})
            continue
        if not WIDE_STORE.match(mn) or not ops:
            continue
        # MUBUF / MTBUF: data first; FLAT / GLOBAL / SCRATCH: address first, data second
        data = _regs(ops[0] if mn.startswith(('buffer_', 'tbuffer_')) else (ops[1] if len(ops) > 1 else ''))
        if len(data) <= 2:
            continue
        # soffset operand of MUBUF: the operand after the s[..] resource; LLVM protects the non-register form itself
        sgpr_soffset = False
        if mn.startswith(('buffer_', 'tbuffer_')):
            for k, o in enumerate(ops):
                if re.match(r'^s\[\d+:\d+\]$', o) and k + 1 < len(ops):
                    nxt = ops[k + 1].split()[0]
                    sgpr_soffset = bool(re.match(r'^(s\d+|m0|ttmp\d+)$', nxt))
                    break
        if i + 1 >= len(insts) or insts[i + 1][0] != kern:
            continue
        _, mn2, ops2, txt2 = insts[i + 1]
        if mn2.startswith(NO_WAIT):
            continue
        if _defs(mn2, ops2) & data:
            hits.append({'kernel': kern, 'store': txt, 'next': txt2, 'sgpr_soffset': sgpr_soffset})
    return hits


def disassemble(lib: pathlib.Path):
    tmp = pathlib.Path(tempfile.mkdtemp(prefix='isa_scan_'))
    try:
        local = tmp / lib.name
        shutil.copy(lib, local)
        subprocess.run([str(LLVM / 'llvm-objdump'), '--offloading', local.name], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = []
        for co in sorted(tmp.glob(lib.name + '.*gfx950*')):
            r = subprocess.run([str(LLVM / 'llvm-objdump'), '-d', str(co)], check=True, capture_output=True, text=True)
            out.append(r.stdout)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan_library(lib) -> list:
    hits = []
    for text in disassemble(pathlib.Path(lib)):
        hits += scan_text(text)
    return hits


if __name__ == '__main__':
    lib = pathlib.Path(sys.argv[1]) if len(sys.argv) > 1 else pathlib.Path(__file__).resolve().parents[1] / 'some_amd' / 'libsome_amd.so'
    found = scan_library(lib)
    for h in found:
        print(f"{h['kernel'][:90]}\n    {h['store']}\n    {h['next']}      <- overwrites store data in the next slot (sgpr soffset: {h['sgpr_soffset']})")
    print(f'{len(found)} unprotected wide-store data overwrite(s) in {lib}')
    sys.exit(1 if found else 0)
