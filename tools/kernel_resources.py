#!/usr/bin/env python
"""Print the register / LDS / scratch figures of the kernels in libsome_amd.so (code-object metadata notes).

    python tools/kernel_resources.py [substring ...]"""
import pathlib
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = pathlib.Path('/opt/rocm/lib/llvm/bin')


def main():
    lib = pathlib.Path(__file__).resolve().parents[1] / 'some_amd' / 'libsome_amd.so'
    want = sys.argv[1:]
    tmp = pathlib.Path(tempfile.mkdtemp(prefix='kres_'))
    try:
        shutil.copy(lib, tmp / lib.name)
        subprocess.run([str(LLVM / 'llvm-objdump'), '--offloading', lib.name], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(tmp.glob(lib.name + '.*gfx950*')):
            txt = subprocess.run([str(LLVM / 'llvm-readelf'), '--notes', str(co)], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r'\n\s*- \.agpr_count', txt)[1:]:
                blk = '.agpr_count' + blk
                get = lambda k: (re.search(r'\.%s:\s*(\S+)' % k, blk) or [None, '?'])[1]          # noqa: E731
                name = get('name')
                dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                if want and not any(w in dem for w in want):
                    continue
                print(f"{dem[:110]:110s} vgpr {get('vgpr_count'):>4s} agpr {get('agpr_count'):>4s} sgpr {get('sgpr_count'):>4s} "
                      f"lds {get('group_segment_fixed_size'):>6s} scratch {get('private_segment_fixed_size'):>5s} "
                      f"spill v {get('vgpr_spill_count')} s {get('sgpr_spill_count')}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
