#!/usr/bin/env python
"""Build a VARIANT of libsome_amd.so in which one source's DEVICE ASSEMBLY has been edited by a regex rule - for hazard hunting:
the only difference between two libraries is then the inserted instructions (e.g. `s_nop`s behind one opcode).

    python tools/build_asm_patch.py <name> <file.hip> [-D... flags] --rule RULE [--kernel SUBSTR]

RULE is one of
    nop_after:<opcode-regex>:<n>      insert `s_nop <n>` (n + 1 wait states, n <= 15; repeated for n > 15) behind every match
    nop_after_sgpr_carry:<n>          ... behind every VALU instruction whose carry-out / mask destination is an SGPR pair
                                      (v_mad_u64_u32 / v_add_co_u32 / v_addc_co_u32 / ... with an s[..] or vcc second operand)
    none                              assemble the compiler's own output (checks that the pipeline reproduces the stock object)
--kernel restricts the edit to kernels whose mangled name contains SUBSTR.

Pipeline (what `hipcc -###` shows, with the .s edited in the middle):
    hipcc -S --cuda-device-only  ->  edit  ->  clang -x assembler (amdgcn)  ->  lld -shared  ->  clang-offload-bundler (.hipfb)
    ->  hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary  ->  link with the stock objects of every other source.
The library lands in tools/_bin/variants/<name>/ (travels to the GPU box); select it with SOME_AMD_LIBRARY=...
"""
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from some_amd import build as B  # noqa: E402

LLVM = pathlib.Path('/opt/rocm/lib/llvm/bin')

CARRY = re.compile(r'^\s+(v_mad_[ui]64_[ui]32|v_add_co_u32\S*|v_sub_co_u32\S*|v_subrev_co_u32\S*|v_addc_co_u32\S*|v_subb_co_u32\S*|'
                   r'v_subbrev_co_u32\S*|v_div_scale\S*)\s+[^,]+,\s*(s\[\d+:\d+\]|vcc)(?=[,\s])')


def nops(n):
    out = []
    while n >= 0:
        k = min(n, 15)
        out.append(f'\ts_nop {k}')
        n -= k + 1
    return out


def edit(text, rule, kernel):
    if rule == 'none':
        return text, 0
    parts = rule.split(':')
    if parts[0] == 'nop_after':
        pat, n = re.compile(r'^\s+(' + parts[1] + r')\b'), int(parts[2])
    elif parts[0] == 'nop_after_sgpr_carry':
        pat, n = CARRY, int(parts[1])
    else:
        raise SystemExit(f'unknown rule {rule}')
    out, cur, hits = [], None, 0
    for line in text.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
        out.append(line)
        if cur is not None and (kernel is None or kernel in cur) and pat.match(line):
            out.extend(nops(n))
            hits += 1
        if '.end_amdhsa_kernel' in line:
            cur = None
    return '\n'.join(out) + '\n', hits


def main():
    args = sys.argv[1:]
    name, src = args[0], args[1]
    rule, kernel, flags = 'none', None, []
    i = 2
    while i < len(args):
        if args[i] == '--rule':
            rule = args[i + 1]; i += 2
        elif args[i] == '--kernel':
            kernel = args[i + 1]; i += 2
        else:
            flags.append(args[i]); i += 1
    B.build(verbose=False)
    out = ROOT / 'tools' / '_bin' / 'variants' / name
    out.mkdir(parents=True, exist_ok=True)
    hipcc = B._hipcc()
    common = B.FLAGS + B.EXTRA_FLAGS.get(src, []) + flags
    s0, s1 = out / (src + '.dev.s'), out / (src + '.patched.s')
    subprocess.check_call([hipcc] + common + ['-S', '--cuda-device-only', str(B.CSRC / src), '-o', str(s0)])
    text, hits = edit(s0.read_text(), rule, kernel)
    s1.write_text(text)
    dev_o, dev_out, fb = out / (src + '.dev.o'), out / (src + '.dev.out'), out / (src + '.hipfb')
    subprocess.check_call([str(LLVM / 'clang'), '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', f'-mcpu={B.ARCH}', '-c', str(s1), '-o', str(dev_o)])
    subprocess.check_call([str(LLVM / 'lld'), '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', str(dev_out), str(dev_o)])
    subprocess.check_call([str(LLVM / 'clang-offload-bundler'), '-type=o', '-bundle-align=4096',
                           f'-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{B.ARCH}', '-input=/dev/null', f'-input={dev_out}', f'-output={fb}'])
    host_o = out / (src + '.o')
    subprocess.check_call([hipcc] + common + ['--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', str(fb), '-c', str(B.CSRC / src), '-o', str(host_o)])
    objs = [str(host_o) if s == src else str(B.OBJ / (s + '.o')) for s in B.SOURCES]
    lib = out / 'libsome_amd.so'
    subprocess.check_call([hipcc, '-shared', '-fPIC', f'--offload-arch={B.ARCH}', '-o', str(lib)] + objs)
    for p in (s0, dev_o, dev_out):
        p.unlink()
    print(f'{lib}  ({hits} edits, rule {rule})')


if __name__ == '__main__':
    main()
